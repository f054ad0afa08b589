"""The boundary's threading contract: Hyrise calls the operators from many scheduler workers at once
(table_scan.cpp:129-131, abstract_scheduler.cpp:53-63), several queries of several clients concurrently.  All mutable state of
the library is thread-local; here sixteen host threads run scans, joins and aggregates at the same time -- half of them on a
HIP stream of their own (hy_set_stream), half on the default stream -- and every result is bit-identical to the oracle's."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from hyrise_amd import abi
from hyrise_amd.operators import aggregate_hash, join_hash, make_predicate, table_scan
from hyrise_amd.storage import DeviceColumn
from support import assert_scan_equal, build_column, oracle_aggregate, oracle_join, oracle_scan

pytestmark = pytest.mark.gpu
THREADS = 16
ROUNDS = 6


def test_operators_from_many_threads(device):
    lib = abi.load_library()
    rng = np.random.default_rng(31)
    n = 400_000
    days = rng.integers(0, 2500, n).astype(np.int32)
    scan_host = build_column(days, rng.random(n) < 0.02, 65535, abi.ENC_DICTIONARY)
    build_host = build_column(rng.permutation(np.arange(0, 60_000, 2, dtype=np.int32)), None, 8000, abi.ENC_UNENCODED)
    probe_host = build_column(rng.integers(0, 60_000, n).astype(np.int32), None, 65535, abi.ENC_FRAME_OF_REFERENCE)
    dup_host = build_column(rng.integers(0, 5_000, 30_000).astype(np.int32), None, 8000, abi.ENC_UNENCODED)
    key_host = build_column(rng.integers(0, 40, n).astype(np.int32) * 977, None, 65535, abi.ENC_DICTIONARY)
    value_host = build_column(rng.integers(-100, 100, n).astype(np.int32), None, 65535, abi.ENC_UNENCODED)
    scan_column, build, probe, dup, key, value = (DeviceColumn(c) for c in (scan_host, build_host, probe_host, dup_host, key_host, value_host))
    predicates = [make_predicate(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, 100 * t, 100 * t + 700, nullable=True) for t in range(THREADS)]
    want_scans = [oracle_scan(scan_host, p) for p in predicates]
    want_join = oracle_join(build_host, probe_host, abi.JOIN_INNER)
    want_dup = oracle_join(dup_host, probe_host, abi.JOIN_INNER)
    want_aggregate = oracle_aggregate([key_host], [(abi.AGG_SUM, value_host), (abi.AGG_COUNT, None)])
    errors = []
    start = threading.Barrier(THREADS)

    def worker(t):
        try:
            stream = None
            if t % 2 == 0:   # a stream of this thread's own; the others stay on the default stream
                stream = torch.cuda.Stream()
                abi.check(lib.hy_set_stream(C.c_void_p(stream.cuda_stream)))
            start.wait()
            for _ in range(ROUNDS):
                assert_scan_equal(table_scan(scan_column, predicates[t]), want_scans[t], f"thread {t}")
                for left, right, want in ((build, probe, want_join), (dup, probe, want_dup)):
                    got = join_hash(left, right, abi.JOIN_INNER)
                    m = want.n_pairs
                    assert got.n_pairs == m and got.left[:m].tobytes() == want.left[:m].tobytes() and got.right[:m].tobytes() == want.right[:m].tobytes()
                    assert np.array_equal(got.slice_offsets[:want.c.n_slices + 1], want.slice_offsets[:want.c.n_slices + 1])
                result = aggregate_hash([key], [(abi.AGG_SUM, value), (abi.AGG_COUNT, None)])
                assert result.n_groups == want_aggregate.n_groups
                assert result.column(0) == want_aggregate.column(0) and result.column(1) == want_aggregate.column(1)
            if stream is not None:
                abi.check(lib.hy_set_stream(None))
            abi.check(lib.hy_shutdown())   # this thread's pool, scratch and staging
        except Exception as error:   # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(error)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(THREADS)]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    assert not errors, errors
