"""Parity of hy_validate (MVCC visibility on the device) with the CPU restatement of Validate: same visible positions in
the same order, same entirely-visible chunks -- data tables, single-chunk / EntireChunk / multi-chunk pos lists."""
import numpy as np
import pytest

from hyrise_amd import abi, storage
from hyrise_amd.operators import validate
from hyrise_amd.storage import DeviceColumn
from support import assert_scan_equal, oracle_validate
from test_oracle_validate import ENTIRE_CHUNK_VISIBLE, TRUTH_TABLE, entire_chunk_case

pytestmark = pytest.mark.gpu


def check(host, device, our_tid, snapshot, shortcut=True, flags=0, context=""):
    got = validate(device, our_tid, snapshot, shortcut, flags)
    want = oracle_validate(host, our_tid, snapshot, shortcut, flags)
    assert_scan_equal(got, want, context)
    return got


def test_truth_table_on_device(device):
    tids, begins, ends = (np.array([row[i] for row in TRUTH_TABLE], dtype=np.uint32) for i in (1, 2, 3))
    host = storage.make_mvcc_column(tids, begins, ends, chunk_size=10, mutable_chunks=(0,))
    got = check(host, DeviceColumn(host), 2, 2, context="validate_visibility_test.cpp truth table")
    assert got.pos_list(0)[:, 1].tolist() == [5, 6]


def test_is_entire_chunk_visible_on_device(device):
    """validate_test.cpp:163-214 (the shortcut is taken by prepare_visibility_jobs per chunk)."""
    for name, begin, mutable, invalid, entirely_visible in ENTIRE_CHUNK_VISIBLE:
        host = entire_chunk_case(begin, mutable, invalid)
        got = check(host, DeviceColumn(host), 1, 1, context=name)
        assert (got.chunk_state[0] == abi.CHUNK_ALL_MATCH) == entirely_visible, name


def test_random_mvcc_data(device):
    rng = np.random.default_rng(41)
    n = 400_000
    tids = rng.integers(0, 6, n).astype(np.uint32)
    begins = np.where(rng.random(n) < 0.05, storage.MAX_COMMIT_ID, rng.integers(1, 40, n)).astype(np.uint32)
    ends = np.where(rng.random(n) < 0.25, rng.integers(1, 50, n), storage.MAX_COMMIT_ID).astype(np.uint32)
    # chunks 3-5 hold only old, never invalidated rows: entirely visible for late snapshots
    chunk = 65535
    begins[3 * chunk:6 * chunk] = rng.integers(1, 10, 3 * chunk)
    ends[3 * chunk:6 * chunk] = storage.MAX_COMMIT_ID
    for chunk_size, mutable in ((chunk, (6,)), (9_000, ()), (150_000, (2,))):
        host = storage.make_mvcc_column(tids, begins, ends, chunk_size=chunk_size, mutable_chunks=mutable)
        dev = DeviceColumn(host)
        for our_tid, snapshot in ((2, 20), (0, 5), (5, 45), (9, 0)):
            for shortcut in (True, False):
                for flags in (0, abi.SCAN_MATERIALIZE_ALL_MATCH):
                    check(host, dev, our_tid, snapshot, shortcut, flags, f"chunk {chunk_size} tid {our_tid} snapshot {snapshot} shortcut {shortcut} flags {flags}")


def test_reference_inputs(device):
    rng = np.random.default_rng(43)
    n, chunk = 300_000, 20_000
    tids = rng.integers(0, 4, n).astype(np.uint32)
    begins = rng.integers(1, 30, n).astype(np.uint32)
    ends = np.where(rng.random(n) < 0.3, rng.integers(1, 40, n), storage.MAX_COMMIT_ID).astype(np.uint32)
    begins[5 * chunk:7 * chunk] = 3
    ends[5 * chunk:7 * chunk] = storage.MAX_COMMIT_ID
    data = storage.make_mvcc_column(tids, begins, ends, chunk_size=chunk)
    data_dev = DeviceColumn(data)
    rows = rng.integers(0, n, 65_535)
    multi = np.stack([rows // chunk, rows % chunk], axis=1).astype(np.uint32)
    single = np.stack([np.full(30_000, 2), rng.integers(0, chunk, 30_000)], axis=1).astype(np.uint32)
    visible_single = np.stack([np.full(1_000, 5), rng.integers(0, chunk, 1_000)], axis=1).astype(np.uint32)
    reference = storage.make_reference_column(data, [multi, single, visible_single, 6, 1, multi[:17]], [None, 2, 5, 6, 1, None])
    ref_dev = DeviceColumn(reference, refs={id(data): data_dev})
    for our_tid, snapshot in ((2, 15), (0, 2), (3, 39)):
        for shortcut in (True, False):
            check(reference, ref_dev, our_tid, snapshot, shortcut, 0, f"reference tid {our_tid} snapshot {snapshot} shortcut {shortcut}")


def test_mvcc_columns_are_for_validate_only(device):
    host = storage.make_mvcc_column([1, 2], [1, 1], [9, 9], chunk_size=10)
    dev = DeviceColumn(host)
    from hyrise_amd.operators import make_predicate, table_scan
    with pytest.raises(abi.HyriseAmdError) as err:
        table_scan(dev, make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, 1))
    assert err.value.status == abi.ERR_INVALID
    data = storage.make_column(np.arange(2, dtype=np.int32), None, abi.ENC_UNENCODED, chunk_size=10)
    with pytest.raises(abi.HyriseAmdError) as err:
        validate(DeviceColumn(data), 1, 1)
    assert err.value.status == abi.ERR_INVALID
