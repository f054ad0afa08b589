"""Parity of the HIP JoinHash with the CPU oracle: the concatenated PosList pairs and the 131 070-element slice
boundaries must be bit-identical (1 GPU: identical order, not just identical multisets)."""
import ctypes as C

import numpy as np
import pytest

from hyrise_amd import abi, storage
from hyrise_amd.operators import join_hash, join_hash_count
from hyrise_amd.storage import DeviceColumn
from support import build_column, load_tbl, oracle_join

pytestmark = pytest.mark.gpu

MODES = [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE,
         abi.JOIN_ANTI_NULL_AS_FALSE]
SEMI = (abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE, abi.JOIN_ANTI_NULL_AS_FALSE)


def assert_join_equal(got, want, mode, context=""):
    assert got.n_pairs == want.n_pairs, f"pair count {context}"
    assert got.c.n_slices == want.c.n_slices, f"slice count {context}"
    assert got.c.left_is_build == want.c.left_is_build and got.c.radix_bits == want.c.radix_bits, context
    n, s = want.n_pairs, want.c.n_slices
    np.testing.assert_array_equal(got.slice_offsets[:s + 1], want.slice_offsets[:s + 1], err_msg=f"slices {context}")
    assert got.left[:n].tobytes() == want.left[:n].tobytes(), f"left PosList differs {context}"
    if mode not in SEMI:
        assert got.right[:n].tobytes() == want.right[:n].tobytes(), f"right PosList differs {context}"


def check(left, right, mode, radix_bits=None, context=""):
    ldev, rdev = DeviceColumn(left), DeviceColumn(right)
    got = join_hash(ldev, rdev, mode, radix_bits)
    want = oracle_join(left, right, mode, radix_bits)
    assert_join_equal(got, want, mode, context)
    return got


@pytest.mark.parametrize("mode", MODES)
def test_join_runner_tables(device, mode):
    """The reference's join_test_runner inputs (join_test_runner.cpp:656-791): sizes {0,10,15} x chunk sizes {10,3,1}
    x nullable / non-nullable int and long keys x encodings x radix settings."""
    for lsize in (0, 10, 15):
        for rsize in (0, 10, 15):
            lt = load_tbl(f"join_test_runner/input_table_left_{lsize}.tbl")
            rt = load_tbl(f"join_test_runner/input_table_right_{rsize}.tbl")
            for column_name in ("int", "int_null", "long", "long_null"):
                lvals, lnull = lt.column("l_" + column_name)
                rvals, rnull = rt.column("r_" + column_name)
                for chunk in (10, 3, 1):
                    for encoding in (abi.ENC_UNENCODED, abi.ENC_DICTIONARY):
                        left = build_column(lvals, lnull, chunk, encoding)
                        right = build_column(rvals, rnull, chunk, encoding)
                        for radix_bits in (None, 0, 2):
                            check(left, right, mode, radix_bits,
                                  f"mode {mode} {column_name} sizes {lsize},{rsize} chunk {chunk} enc {encoding} radix {radix_bits}")


@pytest.mark.parametrize("mode", MODES)
def test_join_random_duplicates_and_nulls(device, mode):
    rng = np.random.default_rng(21 + mode)
    for n_left, n_right, domain in ((3000, 20000, 500), (20000, 3000, 5000), (50000, 50000, 100000)):
        lvals = rng.integers(-domain, domain, n_left).astype(np.int32)
        rvals = rng.integers(-domain, domain, n_right).astype(np.int32)
        lnull, rnull = rng.random(n_left) < 0.05, rng.random(n_right) < 0.05
        for with_nulls in (False, True):
            for lenc, renc in ((abi.ENC_UNENCODED, abi.ENC_FRAME_OF_REFERENCE), (abi.ENC_DICTIONARY, abi.ENC_UNENCODED)):
                left = build_column(lvals, lnull if with_nulls else None, 4096, lenc)
                right = build_column(rvals, rnull if with_nulls else None, 7000, renc)
                for radix_bits in (None, 0, 3, 8):
                    check(left, right, mode, radix_bits, f"mode {mode} n {n_left},{n_right} nulls {with_nulls} radix {radix_bits}")


def test_join_int_with_long(device):
    rng = np.random.default_rng(3)
    left = build_column(rng.integers(0, 1000, 5000).astype(np.int64) * 3_000_000_000, None, 1000, abi.ENC_UNENCODED)
    right = build_column(rng.integers(0, 1000, 9000).astype(np.int64) * 3_000_000_000, None, 2000, abi.ENC_DICTIONARY)
    check(left, right, abi.JOIN_INNER, 2)
    mixed = build_column(rng.integers(0, 1000, 9000).astype(np.int32), None, 2000, abi.ENC_FRAME_OF_REFERENCE)
    small = build_column(rng.integers(0, 1000, 500).astype(np.int64), None, 100, abi.ENC_UNENCODED)
    check(small, mixed, abi.JOIN_INNER, 1)


def test_join_slices_of_131070(device):
    n = 300_000
    probe_values = (np.arange(n) % 1000).astype(np.int32)
    build_values = np.arange(0, 1000, 2, dtype=np.int32)
    build = build_column(build_values, None, 1000, abi.ENC_UNENCODED)
    probe = build_column(probe_values, None, 65535, abi.ENC_UNENCODED)
    for radix_bits in (0, 1, 4):
        for mode in (abi.JOIN_INNER, abi.JOIN_SEMI, abi.JOIN_LEFT):
            if mode == abi.JOIN_SEMI:
                got = check(probe, build, mode, radix_bits)
            else:
                got = check(build, probe, mode, radix_bits)
    assert got.c.n_slices >= 2


def test_join_reference_inputs(device):
    """Join over reference tables (outputs of earlier scans): RowIDs are positions in the input tables
    (join_hash_steps.hpp:364-371)."""
    rng = np.random.default_rng(8)
    base_l = build_column(rng.integers(0, 300, 6000).astype(np.int32), rng.random(6000) < 0.1, 1000, abi.ENC_DICTIONARY)
    base_r = build_column(rng.integers(0, 300, 9000).astype(np.int32), None, 1500, abi.ENC_UNENCODED)
    pos_l = [np.stack([np.full(400, c, dtype=np.uint32), np.sort(rng.choice(1000, 400, replace=False)).astype(np.uint32)], axis=1)
             for c in range(base_l.n_chunks)]
    mixed = np.stack([rng.integers(0, base_r.n_chunks, 2500).astype(np.uint32), rng.integers(0, 1500, 2500).astype(np.uint32)], axis=1)
    mixed[::97] = 0xFFFFFFFF   # NULL_ROW_IDs from an outer join
    ref_l = storage.make_reference_column(base_l, pos_l, list(range(base_l.n_chunks)))
    ref_r = storage.make_reference_column(base_r, [mixed, 2], [None, 2])
    bl, br = DeviceColumn(base_l), DeviceColumn(base_r)
    dl, dr = DeviceColumn(ref_l, refs={id(base_l): bl}), DeviceColumn(ref_r, refs={id(base_r): br})
    for mode in MODES:
        for radix_bits in (0, 2):
            got = join_hash(dl, dr, mode, radix_bits)
            want = oracle_join(ref_l, ref_r, mode, radix_bits)
            assert_join_equal(got, want, mode, f"reference inputs mode {mode} radix {radix_bits}")


@pytest.mark.parametrize("with_nulls", [False, True], ids=["no_nulls", "null_row_ids"])
def test_join_large_reference_inputs_are_materialised(device, with_nulls):
    """Reference inputs of 131 072 rows and more are read through their PosLists once, into a plain int32 column with the input's chunk layout
    (JoinHash materialises its inputs, join_hash_steps.hpp:274-330), and the join runs on that -- unless a key is NULL (a NULL RowID of an
    outer join, a NULL cell), which keeps the reference column.  The pairs are positions in the INPUT tables either way: oracle's bytes."""
    rng = np.random.default_rng(131 + with_nulls)
    n_keys, n_l, n_r = 40_000, 150_000, 260_000
    base_l = build_column(rng.permutation(n_keys).astype(np.int32), None, 7_000, abi.ENC_UNENCODED)                      # unique keys: a rank table
    base_r = build_column(rng.integers(0, n_keys + 500, 300_000).astype(np.int32), None, 50_000, abi.ENC_FRAME_OF_REFERENCE)
    pos_l = [np.stack([np.full(5_000, c, dtype=np.uint32), np.sort(rng.choice(7_000 if c + 1 < base_l.n_chunks else n_keys - 7_000 * (base_l.n_chunks - 1), 5_000, replace=False)).astype(np.uint32)], axis=1)
             for c in range(base_l.n_chunks)]
    # the right input: chunks of irregular sizes (what a join's output looks like), positions all over the base table
    sizes = [1, 33_333, 100_001, n_r - 133_335]
    pos_r = []
    for size in sizes:
        p = np.stack([rng.integers(0, base_r.n_chunks, size).astype(np.uint32), rng.integers(0, 50_000, size).astype(np.uint32)], axis=1)
        if with_nulls:
            p[::1013] = 0xFFFFFFFF
        pos_r.append(p)
    ref_l = storage.make_reference_column(base_l, pos_l, list(range(base_l.n_chunks)))
    ref_r = storage.make_reference_column(base_r, pos_r, [None] * len(sizes))
    assert sum(len(p) for p in pos_l) < 131_072 <= sum(sizes)
    bl, br = DeviceColumn(base_l), DeviceColumn(base_r)
    dl, dr = DeviceColumn(ref_l, refs={id(base_l): bl}), DeviceColumn(ref_r, refs={id(base_r): br})
    for mode in (abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_FALSE):
        for left, right, host_l, host_r in ((dl, dr, ref_l, ref_r), (dr, dl, ref_r, ref_l)):
            got = join_hash(left, right, mode, None)
            want = oracle_join(host_l, host_r, mode, None)
            assert_join_equal(got, want, mode, f"large reference inputs, mode {mode}, nulls {with_nulls}")


def test_join_reference_inputs_with_float_keys(device):
    """The same shape of reference tables, a float column against an int64 one (HashedType float)."""
    rng = np.random.default_rng(9)
    base_l = build_column((rng.integers(0, 300, 6000) / 2).astype(np.float32), rng.random(6000) < 0.1, 1000, abi.ENC_DICTIONARY)
    base_r = build_column(rng.integers(0, 150, 9000).astype(np.int64), None, 1500, abi.ENC_UNENCODED)
    pos_l = [np.stack([np.full(400, c, dtype=np.uint32), np.sort(rng.choice(1000, 400, replace=False)).astype(np.uint32)], axis=1)
             for c in range(base_l.n_chunks)]
    mixed = np.stack([rng.integers(0, base_r.n_chunks, 2500).astype(np.uint32), rng.integers(0, 1500, 2500).astype(np.uint32)], axis=1)
    mixed[::97] = 0xFFFFFFFF   # NULL_ROW_IDs from an outer join
    ref_l = storage.make_reference_column(base_l, pos_l, list(range(base_l.n_chunks)))
    ref_r = storage.make_reference_column(base_r, [mixed, 2], [None, 2])
    bl, br = DeviceColumn(base_l), DeviceColumn(base_r)
    dl, dr = DeviceColumn(ref_l, refs={id(base_l): bl}), DeviceColumn(ref_r, refs={id(base_r): br})
    for mode in MODES:
        got = join_hash(dl, dr, mode, 2)
        want = oracle_join(ref_l, ref_r, mode, 2)
        assert_join_equal(got, want, mode, f"float reference inputs mode {mode}")
    assert join_hash(dl, dr, abi.JOIN_INNER, 2).n_pairs > 1000


def test_join_tpch_orders_lineitem(device):
    """lineitem x orders on the order key at SF 0.1 shape: unique sparse build keys (unencoded), FoR-encoded probe keys,
    every probe row has exactly one partner (config 3 of BASELINE.json, scaled down)."""
    from hyrise_amd import tpch
    data = tpch.TpchData(scale_factor=0.1, seed=7)
    orders = storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)
    lineitem = storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
    assert lineitem.segments[0].width == 2
    got = check(orders, lineitem, abi.JOIN_INNER, None, "orders x lineitem")
    assert got.n_pairs == data.n_lineitems
    assert got.c.left_is_build == 1


def _device_buffer(lib, shape, dtype, fill):
    """A device buffer through the C ABI, pre-filled (to see what a join leaves untouched)."""
    host = np.full(shape, fill, dtype=dtype)
    pointer = C.c_void_p()
    abi.check(lib.hy_device_malloc(C.byref(pointer), max(host.nbytes, 256)))
    abi.check(lib.hy_memcpy_h2d(pointer.value, host.ctypes.data, host.nbytes))
    return pointer.value, host


def _read_back(lib, pointer, like):
    out = np.empty_like(like)
    abi.check(lib.hy_memcpy_d2h(out.ctypes.data, pointer, out.nbytes))
    return out


@pytest.mark.parametrize("mode", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_SEMI])
def test_join_device_memory_result_equals_host_memory_result(device, mode):
    """Device-memory results take the path without a host round trip between the probe passes (the plan of the output is
    made on the device): same pairs, same PosList cuts as the host-memory result."""
    rng = np.random.default_rng(31)
    left_host = build_column(rng.integers(0, 40_000, 300_000).astype(np.int32), rng.random(300_000) < 0.02, 65_535, abi.ENC_DICTIONARY)
    right_host = build_column(rng.integers(0, 40_000, 50_000).astype(np.int32), None, 20_000, abi.ENC_UNENCODED)
    left, right = DeviceColumn(left_host), DeviceColumn(right_host)
    want = join_hash(left, right, mode)
    n, slices = want.n_pairs, int(want.c.n_slices)
    p_left, like_pairs = _device_buffer(device, (n + 8, 2), np.uint32, 0xABABABAB)
    p_right, _ = _device_buffer(device, (n + 8, 2), np.uint32, 0xABABABAB)
    p_offsets, like_offsets = _device_buffer(device, (slices + 8,), np.uint64, 0xCDCDCDCDCDCDCDCD)
    r = abi.JoinResult()
    r.mem, r.radix_bits = abi.MEM_DEVICE, int(want.c.radix_bits)
    r.left_pos, r.right_pos, r.capacity = p_left, p_right, n
    r.slice_offsets, r.slice_capacity = p_offsets, slices
    abi.check(device.hy_join_hash(left.handle, right.handle, mode, C.byref(r)))
    assert int(r.n_pairs) == n and int(r.n_slices) == slices and int(r.left_is_build) == int(want.c.left_is_build)
    got_left, got_right, got_offsets = _read_back(device, p_left, like_pairs), _read_back(device, p_right, like_pairs), _read_back(device, p_offsets, like_offsets)
    build_is_left = bool(want.c.left_is_build)
    semi = mode == abi.JOIN_SEMI
    if not (semi and build_is_left):
        assert got_left[:n].tobytes() == want.left[:n].tobytes()
    if not (semi and not build_is_left):
        assert got_right[:n].tobytes() == want.right[:n].tobytes()
    assert got_offsets[:slices + 1].tobytes() == want.slice_offsets[:slices + 1].tobytes()
    assert (got_left[n:] == 0xABABABAB).all() and (got_right[n:] == 0xABABABAB).all()   # nothing behind the pairs
    for p in (p_left, p_right, p_offsets):
        device.hy_device_free(p)


def test_join_async_leaves_its_status_on_the_device(device, options):
    """HY_JOIN_ASYNC: the call returns with its kernels queued, the pair count / PosList count / fit flag / hint verdict go to
    hy_join_status in device memory and hy_join_hash_finish reports what the synchronous call reports.  Three joins are queued back
    to back before the first finish (the temporaries of one are the next one's, in stream order); a shape that needs a host decision
    (duplicate build keys) runs synchronously under the flag and fills the status all the same; a result that does not fit writes
    nothing; a build column that contradicts its hint writes nothing and finish runs the join again."""
    rng = np.random.default_rng(77)
    keys = (np.arange(120_000, dtype=np.int32) * 3 + 7)
    build_host = build_column(keys, None, 65535, abi.ENC_UNENCODED)
    probe_host = build_column(np.sort(rng.integers(0, 360_100, 700_000).astype(np.int32)), None, 65535, abi.ENC_FRAME_OF_REFERENCE)
    duplicate_host = build_column(np.repeat(np.arange(50_000, dtype=np.int32), 2), None, 65535, abi.ENC_UNENCODED)
    build, probe, duplicate = DeviceColumn(build_host), DeviceColumn(probe_host), DeviceColumn(duplicate_host)
    want = join_hash(build, probe, abi.JOIN_INNER)          # (the first join over `build`: leaves the key hint behind)
    n, slices = want.n_pairs, int(want.c.n_slices)

    class Run:
        def __init__(self, left, right, capacity, slice_capacity):
            self.left, self.right = left, right
            self.p_left, self.like = _device_buffer(device, (capacity + 8, 2), np.uint32, 0xABABABAB)
            self.p_right, _ = _device_buffer(device, (capacity + 8, 2), np.uint32, 0xABABABAB)
            self.p_offsets, self.like_offsets = _device_buffer(device, (slice_capacity + 8,), np.uint64, 0xCDCDCDCDCDCDCDCD)
            self.p_status, self.like_status = _device_buffer(device, (4,), np.uint64, 0xEEEEEEEEEEEEEEEE)
            r = abi.JoinResult()
            r.mem, r.radix_bits, r.flags, r.status = abi.MEM_DEVICE, 0xFFFFFFFF, abi.JOIN_ASYNC, self.p_status
            r.left_pos, r.right_pos, r.capacity = self.p_left, self.p_right, capacity
            r.slice_offsets, r.slice_capacity = self.p_offsets, slice_capacity
            self.r = r

        def start(self):
            return device.hy_join_hash(self.left.handle, self.right.handle, abi.JOIN_INNER, C.byref(self.r))

        def finish(self):
            return device.hy_join_hash_finish(self.left.handle, self.right.handle, abi.JOIN_INNER, C.byref(self.r))

        def pairs(self):
            return _read_back(device, self.p_left, self.like), _read_back(device, self.p_right, self.like), _read_back(device, self.p_offsets, self.like_offsets)

        def status(self):
            words = _read_back(device, self.p_status, self.like_status)
            return int(words[0]), int(words[1] & 0xFFFFFFFF), int(words[1] >> 32), int(words[2] & 0xFFFFFFFF), int(words[2] >> 32)   # n_pairs, n_slices, fits, build_confirmed, error

        def free(self):
            for p in (self.p_left, self.p_right, self.p_offsets, self.p_status):
                device.hy_device_free(p)

    runs = [Run(build, probe, n, slices) for _ in range(3)]
    for run in runs:
        abi.check(run.start())
        assert device.hy_debug_join_build_was_hinted() == 1 and device.hy_debug_join_used_pkfk() == 1
    for run in runs:
        abi.check(run.finish())
        assert int(run.r.n_pairs) == n and int(run.r.n_slices) == slices
        assert run.status() == (n, slices, 1, 1, 0)
        got_left, got_right, got_offsets = run.pairs()
        assert got_left[:n].tobytes() == want.left[:n].tobytes() and got_right[:n].tobytes() == want.right[:n].tobytes()
        assert got_offsets[:slices + 1].tobytes() == want.slice_offsets[:slices + 1].tobytes()
        assert (got_left[n:] == 0xABABABAB).all() and (got_right[n:] == 0xABABABAB).all()
        run.free()
    # does not fit: nothing written, finish reports the needed sizes
    small = Run(build, probe, n - 1, slices)
    abi.check(small.start())
    assert small.finish() == abi.ERR_CAPACITY and int(small.r.n_pairs) == n
    assert small.status()[:3] == (n, slices, 0)
    got_left, got_right, got_offsets = small.pairs()
    assert (got_left == 0xABABABAB).all() and (got_right == 0xABABABAB).all() and (got_offsets == 0xCDCDCDCDCDCDCDCD).all()
    small.free()
    # duplicate build keys: the sorted directory needs host decisions -- synchronous under the flag, status filled all the same
    want_duplicate = join_hash(duplicate, probe, abi.JOIN_INNER)
    m, m_slices = want_duplicate.n_pairs, int(want_duplicate.c.n_slices)
    sync = Run(duplicate, probe, m, m_slices)
    abi.check(sync.start())
    assert int(sync.r.n_pairs) == m          # (already known: the call was synchronous)
    abi.check(sync.finish())
    assert sync.status() == (m, m_slices, 1, 1, 0)
    got_left, got_right, _ = sync.pairs()
    assert got_left[:m].tobytes() == want_duplicate.left[:m].tobytes() and got_right[:m].tobytes() == want_duplicate.right[:m].tobytes()
    sync.free()
    # a hint that does not hold: the device notices (nothing written, build_confirmed = 0), finish drops the hint and runs the join again
    options.set(abi.OPT_JOIN_BREAK_HINT, 1)
    broken = Run(build, probe, n, slices)
    abi.check(broken.start())
    device.hy_synchronize()
    assert broken.status()[2:4] == (0, 0)
    got_left, _, _ = broken.pairs()
    assert (got_left == 0xABABABAB).all()
    abi.check(broken.finish())
    assert device.hy_debug_join_build_was_hinted() == 2 and int(broken.r.n_pairs) == n
    assert broken.status() == (n, slices, 1, 1, 0)
    got_left, got_right, got_offsets = broken.pairs()
    assert got_left[:n].tobytes() == want.left[:n].tobytes() and got_right[:n].tobytes() == want.right[:n].tobytes()
    assert got_offsets[:slices + 1].tobytes() == want.slice_offsets[:slices + 1].tobytes()
    broken.free()


@pytest.mark.parametrize("mem", [abi.MEM_HOST, abi.MEM_DEVICE])
def test_join_capacity_errors_leave_the_buffers_alone(device, mem):
    """A result that does not fit is HY_ERR_CAPACITY with the needed sizes reported -- decided on the device between the
    probe passes: pass 2 must not write a single pair."""
    rng = np.random.default_rng(32)
    left_host = build_column(rng.integers(0, 1_000, 100_000).astype(np.int32), None, 65_535, abi.ENC_UNENCODED)
    right_host = build_column(np.arange(1_000, dtype=np.int32), None, 65_535, abi.ENC_UNENCODED)
    left, right = DeviceColumn(left_host), DeviceColumn(right_host)
    n = join_hash_count(left, right, abi.JOIN_INNER)
    assert n == 100_000
    for capacity, slice_capacity in ((n - 1, 64), (n, 0)):
        r = abi.JoinResult()
        r.mem, r.radix_bits, r.capacity, r.slice_capacity = mem, 0xFFFFFFFF, capacity, slice_capacity
        if mem == abi.MEM_HOST:
            left_pos, right_pos = np.full((n, 2), 0xABABABAB, dtype=np.uint32), np.full((n, 2), 0xABABABAB, dtype=np.uint32)
            offsets = np.full(80, 0xCDCDCDCDCDCDCDCD, dtype=np.uint64)
            r.left_pos, r.right_pos, r.slice_offsets = left_pos.ctypes.data, right_pos.ctypes.data, offsets.ctypes.data
        else:
            p_left, like_pairs = _device_buffer(device, (n, 2), np.uint32, 0xABABABAB)
            p_right, _ = _device_buffer(device, (n, 2), np.uint32, 0xABABABAB)
            p_offsets, like_offsets = _device_buffer(device, (80,), np.uint64, 0xCDCDCDCDCDCDCDCD)
            r.left_pos, r.right_pos, r.slice_offsets = p_left, p_right, p_offsets
        status = device.hy_join_hash(left.handle, right.handle, abi.JOIN_INNER, C.byref(r))
        assert status == abi.ERR_CAPACITY
        assert int(r.n_pairs) == n and int(r.n_slices) >= 1   # what the caller needs to retry
        if mem == abi.MEM_DEVICE:
            left_pos, right_pos, offsets = _read_back(device, p_left, like_pairs), _read_back(device, p_right, like_pairs), _read_back(device, p_offsets, like_offsets)
            for p in (p_left, p_right, p_offsets):
                device.hy_device_free(p)
        assert (left_pos == 0xABABABAB).all() and (right_pos == 0xABABABAB).all()
        assert (offsets == 0xCDCDCDCDCDCDCDCD).all()


SECONDARY_MODES = [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_FALSE]


@pytest.mark.parametrize("mode", SECONDARY_MODES)
def test_join_with_secondary_predicates_runner_tables(device, mode):
    """Multi-predicate joins on the reference's join_test_runner inputs (join_test_runner.cpp:207-211, 464-480): the
    secondary predicates filter the partners the key finds, inside the probe -- same pairs in the same order as the
    oracle (which agrees with the nested loops of JoinVerification: tests/test_oracle_join.py)."""
    from test_oracle_join import SECONDARY_SETS, runner_tables, secondary_columns
    for lsize, rsize in ((10, 15), (15, 10), (15, 15), (0, 10), (10, 0)):
        lt, rt = runner_tables(lsize, 0)[0], runner_tables(rsize, 0)[1]
        for key in ("int", "long_null"):
            lvals, lnull = lt.column("l_" + key)
            rvals, rnull = rt.column("r_" + key)
            for chunk, encoding in ((10, abi.ENC_UNENCODED), (3, abi.ENC_DICTIONARY)):
                left, right = build_column(lvals, lnull, chunk, encoding), build_column(rvals, rnull, chunk, encoding)
                ldev, rdev = DeviceColumn(left), DeviceColumn(right)
                for predicate_set in SECONDARY_SETS:
                    secondary = secondary_columns(lt, rt, predicate_set, chunk, encoding)
                    on_device = [(DeviceColumn(l), c, DeviceColumn(r)) for l, c, r in secondary]
                    for radix_bits in (None, 2):
                        got = join_hash(ldev, rdev, mode, radix_bits, secondary=on_device)
                        want = oracle_join(left, right, mode, radix_bits, secondary=secondary)
                        assert_join_equal(got, want, mode, f"key {key} sizes {lsize},{rsize} chunk {chunk} predicates {predicate_set} radix {radix_bits}")


@pytest.mark.parametrize("mode", SECONDARY_MODES)
def test_join_with_secondary_predicates_random(device, mode):
    """Many partners per key, two secondary predicates on columns of mixed types and encodings, NULLs on all sides, several
    chunks and radix partitions: the device against the oracle, pair by pair."""
    rng = np.random.default_rng(41 + mode)
    n_left, n_right, chunk = 60_000, 9_000, 20_000
    lkey = build_column(rng.integers(0, 500, n_left).astype(np.int32), rng.random(n_left) < 0.02, chunk, abi.ENC_DICTIONARY)
    rkey = build_column(rng.integers(0, 500, n_right).astype(np.int32), rng.random(n_right) < 0.02, chunk, abi.ENC_UNENCODED)
    la = build_column(rng.integers(-50, 50, n_left).astype(np.int64), rng.random(n_left) < 0.05, chunk, abi.ENC_UNENCODED)
    ra = build_column((rng.random(n_right) * 100 - 50).astype(np.float32), None, chunk, abi.ENC_DICTIONARY)
    lb = build_column(rng.integers(0, 1000, n_left).astype(np.int32), None, chunk, abi.ENC_FRAME_OF_REFERENCE)
    rb = build_column(rng.random(n_right) * 1000, rng.random(n_right) < 0.05, chunk, abi.ENC_UNENCODED)
    secondary = [(la, abi.PRED_LESS_THAN, ra), (lb, abi.PRED_NOT_EQUALS, rb)]
    ldev, rdev = DeviceColumn(lkey), DeviceColumn(rkey)
    on_device = [(DeviceColumn(l), c, DeviceColumn(r)) for l, c, r in secondary]
    for radix_bits in (None, 3):
        got = join_hash(ldev, rdev, mode, radix_bits, secondary=on_device)
        want = oracle_join(lkey, rkey, mode, radix_bits, secondary=secondary)
        assert_join_equal(got, want, mode, f"random, radix {radix_bits}")
        assert 0 < got.n_pairs
    plain = join_hash(ldev, rdev, mode, None)
    if mode == abi.JOIN_INNER:
        assert got.n_pairs < plain.n_pairs


def test_join_secondary_predicate_rejections(device):
    """AntiNullAsTrue with secondary predicates, string or mis-shaped columns, non-comparisons: rejected like
    JoinHash::supports (join_hash.cpp:39-44) / the evaluator's type check."""
    a = DeviceColumn(build_column(np.arange(10, dtype=np.int32), None, 5, abi.ENC_UNENCODED))
    b = DeviceColumn(build_column(np.arange(10, dtype=np.int32), None, 5, abi.ENC_UNENCODED))
    other_layout = DeviceColumn(build_column(np.arange(10, dtype=np.int32), None, 4, abi.ENC_UNENCODED))
    with pytest.raises(Exception):
        join_hash(a, b, abi.JOIN_ANTI_NULL_AS_TRUE, secondary=[(a, abi.PRED_LESS_THAN, b)])
    with pytest.raises(Exception):
        join_hash(a, b, abi.JOIN_INNER, secondary=[(other_layout, abi.PRED_LESS_THAN, b)])
    with pytest.raises(Exception):
        join_hash(a, b, abi.JOIN_INNER, secondary=[(a, abi.PRED_IS_NULL, b)])
    with pytest.raises(Exception):
        join_hash(a, b, abi.JOIN_INNER, secondary=[(a, abi.PRED_EQUALS, b)] * 5)


@pytest.mark.parametrize("mode", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_RIGHT])
def test_join_many_partners_per_probe_row(device, mode):
    """Eight partners per probe row and more: a 4096-row probe tile produces several staging buffers of pairs (the generic
    pass 2 lays them out window by window), with and without a secondary predicate, radix partitioning on and off."""
    rng = np.random.default_rng(51)
    build_keys = np.repeat(np.arange(2_000, dtype=np.int32), 8)
    rng.shuffle(build_keys)
    build_keys[::97] = 7                       # one key with a few hundred partners
    probe_keys = rng.integers(0, 2_100, 30_000).astype(np.int32)
    small = build_column(build_keys, rng.random(len(build_keys)) < 0.01, 5_000, abi.ENC_UNENCODED)
    large = build_column(probe_keys, rng.random(len(probe_keys)) < 0.01, 20_000, abi.ENC_DICTIONARY)
    small_other = build_column(rng.integers(0, 100, len(build_keys)).astype(np.int32), None, 5_000, abi.ENC_UNENCODED)
    large_other = build_column(rng.integers(0, 100, len(probe_keys)).astype(np.int64), None, 20_000, abi.ENC_UNENCODED)
    left, right = (large, small) if mode != abi.JOIN_RIGHT else (small, large)   # the probe side is the large one
    left_other, right_other = (large_other, small_other) if mode != abi.JOIN_RIGHT else (small_other, large_other)
    ldev, rdev = DeviceColumn(left), DeviceColumn(right)
    for radix_bits in (0, 3):
        got = join_hash(ldev, rdev, mode, radix_bits)
        assert_join_equal(got, oracle_join(left, right, mode, radix_bits), mode, f"many partners, radix {radix_bits}")
        assert got.n_pairs > 8 * 25_000
        secondary = [(left_other, abi.PRED_LESS_THAN, right_other)]
        on_device = [(DeviceColumn(left_other), abi.PRED_LESS_THAN, DeviceColumn(right_other))]
        got = join_hash(ldev, rdev, mode, radix_bits, secondary=on_device)
        assert_join_equal(got, oracle_join(left, right, mode, radix_bits, secondary=secondary), mode, f"many partners + predicate, radix {radix_bits}")


def test_join_secondary_predicates_on_reference_inputs(device):
    """Secondary predicates over reference tables: the predicate columns are reference segments with the key columns' pos
    lists (all columns of a reference table's chunk share one pos list), NULL_ROW_IDs included."""
    rng = np.random.default_rng(18)
    n_l, n_r = 6000, 9000
    base_lkey = build_column(rng.integers(0, 200, n_l).astype(np.int32), rng.random(n_l) < 0.05, 1000, abi.ENC_DICTIONARY)
    base_rkey = build_column(rng.integers(0, 200, n_r).astype(np.int32), None, 1500, abi.ENC_UNENCODED)
    base_lval = build_column(rng.integers(0, 50, n_l).astype(np.int64), rng.random(n_l) < 0.05, 1000, abi.ENC_UNENCODED)
    base_rval = build_column((rng.random(n_r) * 50).astype(np.float32), None, 1500, abi.ENC_DICTIONARY)
    pos_l = [np.stack([np.full(400, c, dtype=np.uint32), np.sort(rng.choice(1000, 400, replace=False)).astype(np.uint32)], axis=1)
             for c in range(base_lkey.n_chunks)]
    mixed = np.stack([rng.integers(0, base_rkey.n_chunks, 2500).astype(np.uint32), rng.integers(0, 1500, 2500).astype(np.uint32)], axis=1)
    mixed[::97] = 0xFFFFFFFF
    single = list(range(base_lkey.n_chunks))
    ref_lkey = storage.make_reference_column(base_lkey, pos_l, single)
    ref_lval = storage.make_reference_column(base_lval, pos_l, single)
    ref_rkey = storage.make_reference_column(base_rkey, [mixed, 2], [None, 2])
    ref_rval = storage.make_reference_column(base_rval, [mixed, 2], [None, 2])
    bases = {id(c): DeviceColumn(c) for c in (base_lkey, base_rkey, base_lval, base_rval)}
    dev = {name: DeviceColumn(col, refs={id(base): bases[id(base)]})
           for name, col, base in (("lkey", ref_lkey, base_lkey), ("lval", ref_lval, base_lval), ("rkey", ref_rkey, base_rkey), ("rval", ref_rval, base_rval))}
    for mode in SECONDARY_MODES:
        for radix_bits in (0, 2):
            got = join_hash(dev["lkey"], dev["rkey"], mode, radix_bits, secondary=[(dev["lval"], abi.PRED_GREATER_THAN_EQUALS, dev["rval"])])
            want = oracle_join(ref_lkey, ref_rkey, mode, radix_bits, secondary=[(ref_lval, abi.PRED_GREATER_THAN_EQUALS, ref_rval)])
            assert_join_equal(got, want, mode, f"reference inputs + secondary predicate, mode {mode} radix {radix_bits}")


# ---- float / double keys, also mixed with integers (JoinHashTraits: both sides are cast to one HashedType) -----------------
NUMERIC_COLUMNS = ("int", "int_null", "long", "long_null", "float", "float_null", "double", "double_null")


@pytest.mark.parametrize("mode", MODES)
def test_join_all_numeric_type_pairs_of_the_runner_tables(device, mode):
    """join_test_runner.cpp:183-193: every data type against every other one; pair order and slices as the oracle's
    (partition = std::hash<HashedType> of the cast key, libstdc++'s, restated in oracle/join.c and csrc/join.hip)."""
    lt = load_tbl("join_test_runner/input_table_left_15.tbl")
    rt = load_tbl("join_test_runner/input_table_right_10.tbl")
    for left_name in NUMERIC_COLUMNS:
        for right_name in NUMERIC_COLUMNS:
            if "float" not in left_name + right_name and "double" not in left_name + right_name:
                continue
            lvals, lnull = lt.column("l_" + left_name)
            rvals, rnull = rt.column("r_" + right_name)
            for chunk, encoding, radix_bits in ((10, abi.ENC_UNENCODED, None), (3, abi.ENC_DICTIONARY, 2), (4, abi.ENC_DICTIONARY, 0)):
                check(build_column(lvals, lnull, chunk, encoding), build_column(rvals, rnull, chunk, encoding), mode, radix_bits,
                      f"mode {mode} {left_name} x {right_name} chunk {chunk} radix {radix_bits}")


def _float_key_column(rng, data_type, n, chunk):
    """Keys that collide after the cast to the HashedType (2^24 + 1 is not a float, 2^53 + 1 not a double), +-0.0 (one key),
    NaN (no key), infinities; NULLs; a dictionary cannot hold NaN (not ordered), so those columns stay unencoded."""
    np_type = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}[data_type]
    if data_type == abi.TYPE_INT:
        domain = np.array([0, 1, -1, 16777216, 16777217, 16777218, 3, 1338, 2147483647, -2147483648], dtype=np_type)
    elif data_type == abi.TYPE_LONG:
        domain = np.array([0, 1, -1, 16777216, 16777217, 3, 1338, 2**40, 2**40 + 1, 2**53, 2**53 + 1, -2**53 - 1, 2**63 - 1], dtype=np_type)
    else:
        domain = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 16777216.0, 16777217.0, 16777218.0, 2.0**40, 2.0**53, 1e30, -1e30, np.inf, -np.inf, np.nan, 3.0, 1338.0,
                           2147483648.0, -2147483648.0, 9.223372036854775807e18], dtype=np_type)
    values = domain[rng.integers(0, len(domain), n)]
    nulls = rng.random(n) < 0.15
    has_nan = np_type in (np.float32, np.float64) and bool(np.isnan(values).any())
    encoding = abi.ENC_UNENCODED if has_nan or rng.random() < 0.5 else abi.ENC_DICTIONARY
    return build_column(values, nulls, chunk, encoding)


@pytest.mark.parametrize("mode", MODES)
def test_join_float_keys_random(device, mode):
    rng = np.random.default_rng(23)
    types = (abi.TYPE_INT, abi.TYPE_LONG, abi.TYPE_FLOAT, abi.TYPE_DOUBLE)
    for left_type in types:
        for right_type in types:
            if left_type in (abi.TYPE_INT, abi.TYPE_LONG) and right_type in (abi.TYPE_INT, abi.TYPE_LONG):
                continue
            for n_left, n_right, chunk, radix_bits in ((400, 250, 70, 3), (120, 6000, 1000, None), (9000, 5000, 4096, 0), (5000, 9000, 65535, 5)):
                check(_float_key_column(rng, left_type, n_left, chunk), _float_key_column(rng, right_type, n_right, chunk), mode, radix_bits,
                      f"mode {mode} types {left_type} x {right_type} rows {n_left},{n_right} radix {radix_bits}")


def test_join_double_keys_many_tiles(device):
    """300 000 probe rows (74 tiles) of doubles with a few thousand distinct values against 40 000 build rows with about a
    dozen rows per key (millions of pairs), then an int64 probe column against the same doubles."""
    rng = np.random.default_rng(29)
    build_values = np.round(rng.normal(size=40000) * 50, 1)
    probe_values = np.round(rng.normal(size=300000) * 50, 1)
    build = build_column(build_values, rng.random(40000) < 0.02, 65535, abi.ENC_DICTIONARY)
    probe = build_column(probe_values.astype(np.float32).astype(np.float64), None, 65535, abi.ENC_UNENCODED)
    for mode in (abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_SEMI):
        got = check(probe, build, mode, None, f"doubles mode {mode}")
    assert got.n_pairs > 1000
    longs = build_column(rng.integers(-300, 300, 300000).astype(np.int64), rng.random(300000) < 0.05, 65535, abi.ENC_UNENCODED)
    got = check(longs, build, abi.JOIN_INNER, None, "long x double")
    assert got.n_pairs > 100000


@pytest.mark.parametrize("mode", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_FALSE])
def test_join_float_keys_with_secondary_predicates(device, mode):
    """Float / double keys and secondary predicates share the general probe kernels: both at once."""
    rng = np.random.default_rng(31)
    for left_type, right_type in ((abi.TYPE_DOUBLE, abi.TYPE_INT), (abi.TYPE_FLOAT, abi.TYPE_DOUBLE), (abi.TYPE_LONG, abi.TYPE_FLOAT)):
        n_left, n_right, chunk = 3000, 5000, 700
        left, right = _float_key_column(rng, left_type, n_left, chunk), _float_key_column(rng, right_type, n_right, chunk)
        left_extra = build_column(rng.integers(0, 10, n_left).astype(np.int32), rng.random(n_left) < 0.1, chunk, abi.ENC_DICTIONARY)
        right_extra = build_column(rng.integers(0, 10, n_right).astype(np.float32), None, chunk, abi.ENC_UNENCODED)
        ldev, rdev = DeviceColumn(left), DeviceColumn(right)
        lx, rx = DeviceColumn(left_extra), DeviceColumn(right_extra)
        for condition in (abi.PRED_LESS_THAN, abi.PRED_NOT_EQUALS):
            got = join_hash(ldev, rdev, mode, 3, secondary=[(lx, condition, rx)])
            want = oracle_join(left, right, mode, 3, secondary=[(left_extra, condition, right_extra)])
            assert_join_equal(got, want, mode, f"mode {mode} types {left_type} x {right_type} condition {condition}")


@pytest.mark.parametrize("mode", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE])
def test_join_string_keys_as_join_ids(device, mode):
    """String keys join as the adapter's ids (hyrise_amd/join_keys.py: unique << 20 | std::hash & 0xFFFFF): runner tables,
    and 20 000 x 50 000 random words; the pairs are checked against the strings themselves."""
    from test_oracle_join import string_key_columns
    lt = load_tbl("join_test_runner/input_table_left_15.tbl")
    rt = load_tbl("join_test_runner/input_table_right_10.tbl")
    for name in ("string", "string_null"):
        lvals, lnull = lt.column("l_" + name)
        rvals, rnull = rt.column("r_" + name)
        for chunk, radix_bits in ((10, None), (3, 2), (1, 8)):
            left, right, _, _ = string_key_columns(lvals, lnull, rvals, rnull, chunk)
            check(left, right, mode, radix_bits, f"strings {name} mode {mode} chunk {chunk} radix {radix_bits}")
    rng = np.random.default_rng(37)
    words = np.array(["w%d" % i + "x" * int(i % 9) for i in range(3000)], dtype=object)
    lvals, rvals = words[rng.integers(0, 3000, 20000)], words[rng.integers(0, 2500, 50000)]
    lnull, rnull = rng.random(20000) < 0.05, rng.random(50000) < 0.05
    left, right, _, _ = string_key_columns(lvals, lnull, rvals, rnull, 4096)
    got = check(left, right, mode, None, f"random words mode {mode}")
    if mode == abi.JOIN_INNER:
        assert got.n_pairs > 100000
        for k in rng.integers(0, got.n_pairs, 200):
            l, r = got.left[k], got.right[k]
            assert lvals[int(l[0]) * 4096 + int(l[1])] == rvals[int(r[0]) * 4096 + int(r[1])]


def used_rank_table():
    lib = abi.load_library()
    lib.hy_debug_join_used_rank_table.restype = C.c_int
    return int(lib.hy_debug_join_used_rank_table())


@pytest.mark.parametrize("mode", MODES)
def test_join_unique_build_keys_take_the_rank_table(device, mode):
    """Unique integer build keys (primary keys) are looked up in the rank table (csrc/join.hip: RankTable): sorted and
    shuffled build sides, sparse keys (TPC-H's 8 of every 32), negative keys, probe NULLs and probe keys outside the build
    range, every radix setting; pairs and 131 070-element cuts bit-identical to the oracle."""
    rng = np.random.default_rng(100 + mode)
    n_build, n_probe = 20000, 150000
    i = np.arange(1, n_build + 1, dtype=np.int64)
    sparse = (((i >> 3) << 5) | (i & 7)).astype(np.int32)
    dense_negative = (np.arange(n_build, dtype=np.int32) - 7000)
    for name, keys in (("sparse", sparse), ("dense_negative", dense_negative)):
        probe_values = rng.choice(keys, n_probe).astype(np.int32)
        outside = rng.random(n_probe) < 0.03
        probe_values[outside] = rng.integers(int(keys.min()) - 500, int(keys.max()) + 500, int(outside.sum())).astype(np.int32)
        probe_nulls = rng.random(n_probe) < 0.02
        sorted_probe = np.sort(probe_values)
        for shuffled in (False, True):
            build_values = rng.permutation(keys) if shuffled else keys
            for build_encoding, probe_encoding, chunk in ((abi.ENC_UNENCODED, abi.ENC_FRAME_OF_REFERENCE, 4096), (abi.ENC_DICTIONARY, abi.ENC_UNENCODED, 3000)):
                build = build_column(build_values, None, chunk, build_encoding)
                for pname, pvalues, pnulls in (("random", probe_values, probe_nulls), ("sorted", sorted_probe, None)):
                    probe = build_column(pvalues, pnulls, 65535, probe_encoding)
                    for radix_bits in (None, 0, 3, 8):
                        context = f"mode {mode} {name} shuffled {shuffled} enc {build_encoding} probe {pname} radix {radix_bits}"
                        if mode in SEMI or mode == abi.JOIN_LEFT:
                            check(probe, build, mode, radix_bits, context)
                        else:
                            check(build, probe, mode, radix_bits, context)
                        assert used_rank_table() in (1, 2), context
                        if build_encoding == abi.ENC_UNENCODED and not shuffled and mode == abi.JOIN_INNER:
                            assert used_rank_table() == 2, context   # dense, sorted, equally sized chunks: ranks are row numbers


@pytest.mark.parametrize("mode", [abi.JOIN_INNER, abi.JOIN_LEFT, abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE])
def test_join_large_shuffled_build_keys_are_sorted_into_the_rank_table(device, mode):
    """65 536 and more shuffled 32-bit build keys: the (key - smallest key, RowID) pairs are radix-sorted and the rank table filled from the
    sorted keys (the atomicOr marking stays for small and 64-bit builds).  Negative keys, sparse keys, a probe side with NULLs and keys
    outside the range; and a build side that turns out to have duplicates after the sort: the sorted directory takes over with the pairs
    in build-row order.  Pairs and PosList cuts bit-identical to the oracle."""
    rng = np.random.default_rng(5 + mode)
    n_build, n_probe = 150_000, 400_000
    keys = (np.arange(n_build, dtype=np.int64) * 5 - 300_000).astype(np.int32)       # sparse (one in five), negative and positive
    positive = keys + 300_007
    for name, keys, build_values in (("unique", keys, rng.permutation(keys)), ("duplicates", keys, rng.permutation(np.concatenate([keys, keys[:5000]]))),
                                     ("duplicates, no negative key", positive, rng.permutation(np.concatenate([positive, positive[-5000:]])))):
        probe_values = rng.choice(keys, n_probe).astype(np.int32)
        outside = rng.random(n_probe) < 0.03
        probe_values[outside] = rng.integers(int(keys.min()) - 500, int(keys.max()) + 500, int(outside.sum())).astype(np.int32)
        probe_nulls = rng.random(n_probe) < 0.02
        build = build_column(build_values, None, 65535, abi.ENC_UNENCODED)
        probe = build_column(probe_values, probe_nulls, 65535, abi.ENC_FRAME_OF_REFERENCE)
        for radix_bits in (None, 0, 4):
            context = f"mode {mode} {name} radix {radix_bits}"
            if mode in SEMI or mode == abi.JOIN_LEFT:
                check(probe, build, mode, radix_bits, context)
            else:
                check(build, probe, mode, radix_bits, context)
            if name == "unique":
                assert used_rank_table() == 1, context
            elif mode == abi.JOIN_INNER:   # (Semi / Anti joins keep presence bits only: duplicates still serve them)
                assert used_rank_table() == 0, context


@pytest.mark.parametrize("radix_bits", [None, 0, 5])
def test_join_scattered_probe_keys_hand_their_ranks_over(device, options, radix_bits):
    """Inner PK-FK joins whose probe keys have no locality (unencoded int32 / 4-byte FrameOfReference offsets): pass 1 leaves every probe
    row's partner rank behind and pass 2 reads it back (pk_count_wave RANKS; forced here for a small probe side).  Keys outside the
    build range, probe rows without a partner inside it (sparse build keys), a partial last tile, NULL-free probe columns of both
    layouts; pairs and PosList cuts bit-identical to the oracle, with and without the hand-over."""
    rng = np.random.default_rng(31)
    n_build, n_probe = 40_000, 300_011
    build_keys = (np.arange(n_build, dtype=np.int64) * 3 - 20_000).astype(np.int32)
    probe_values = rng.integers(int(build_keys.min()) - 300, int(build_keys.max()) + 300, n_probe).astype(np.int32)
    build = build_column(build_keys, None, 65535, abi.ENC_UNENCODED)
    for encoding in (abi.ENC_UNENCODED, abi.ENC_FRAME_OF_REFERENCE):
        probe = build_column(probe_values, None, 65535, encoding)
        for threshold in (1, 0):
            options.set(abi.OPT_JOIN_HAND_OVER_RANKS, threshold)
            options.set(abi.OPT_JOIN_LDS_BUILD, 0)    # (the build side's bits staged in LDS is another path: not this test's)
            check(build, probe, abi.JOIN_INNER, radix_bits, f"encoding {encoding} threshold {threshold} radix {radix_bits}")
            assert used_rank_table() in (1, 2)


def test_join_rank_table_int64_and_reference_build(device):
    rng = np.random.default_rng(77)
    keys = (np.arange(5000, dtype=np.int64) * 3 + 10_000_000_000)
    build = build_column(rng.permutation(keys), None, 1000, abi.ENC_UNENCODED)
    probe = build_column(rng.choice(keys, 40000) + rng.integers(0, 2, 40000), rng.random(40000) < 0.05, 7000, abi.ENC_DICTIONARY)
    for mode in MODES:
        if mode in SEMI or mode == abi.JOIN_LEFT:
            check(probe, build, mode, 2, f"int64 mode {mode}")
        else:
            check(build, probe, mode, 2, f"int64 mode {mode}")
        assert used_rank_table() == 1
    # a filtered dimension table as the build side: reference segments over unique keys, ascending
    base = build_column(np.arange(0, 9000, 3, dtype=np.int32), None, 1000, abi.ENC_UNENCODED)
    pos = [np.stack([np.full(300, c, dtype=np.uint32), np.sort(rng.choice(1000, 300, replace=False)).astype(np.uint32)], axis=1) for c in range(base.n_chunks)]
    ref = storage.make_reference_column(base, pos, list(range(base.n_chunks)))
    fact = build_column(rng.integers(0, 9000, 50000).astype(np.int32), None, 65535, abi.ENC_UNENCODED)
    db, df = DeviceColumn(base), DeviceColumn(fact)
    dr = DeviceColumn(ref, refs={id(base): db})
    for mode in (abi.JOIN_INNER, abi.JOIN_RIGHT):
        got = join_hash(dr, df, mode, None)
        want = oracle_join(ref, fact, mode, None)
        assert_join_equal(got, want, mode, f"reference build mode {mode}")
        assert used_rank_table() == 1


def test_join_rank_table_falls_back(device, options):
    """Duplicate build keys (found while the table is marked) and sparse key ranges use the sorted directory; so does
    HY_OPT_JOIN_RANK_TABLE = 0, with identical results."""
    rng = np.random.default_rng(5)
    keys = rng.permutation(np.arange(3000, dtype=np.int32))
    keys[17] = keys[2900]   # one duplicate, far apart
    build = build_column(keys, None, 500, abi.ENC_UNENCODED)
    probe = build_column(rng.integers(0, 3000, 20000).astype(np.int32), None, 4000, abi.ENC_UNENCODED)
    check(build, probe, abi.JOIN_INNER, 2, "duplicate")
    assert used_rank_table() == 0
    sparse = build_column((np.arange(2000, dtype=np.int64) * 1_000_003).astype(np.int32), None, 500, abi.ENC_UNENCODED)
    check(sparse, probe, abi.JOIN_INNER, 2, "sparse")
    assert used_rank_table() == 0
    unique = build_column(np.arange(3000, dtype=np.int32), None, 500, abi.ENC_UNENCODED)
    a = check(unique, probe, abi.JOIN_INNER, 2, "rank table")
    assert used_rank_table() == 2
    options.set(abi.OPT_JOIN_RANK_TABLE, 0)
    b = check(unique, probe, abi.JOIN_INNER, 2, "directory")
    assert used_rank_table() == 0
    assert a.left[:a.n_pairs].tobytes() == b.left[:b.n_pairs].tobytes() and a.right[:a.n_pairs].tobytes() == b.right[:b.n_pairs].tobytes()


def test_rank_table_join_with_and_without_lane_ordered_atomics(device):
    """rt_probe_emit ranks the pairs of a partition inside a wave either with one returning LDS atomic per pair (where the device
    serves the lanes of one instruction in lane order -- probed once per process) or with match-any groups; both produce the
    oracle's bytes.  The probe's verdict is reported."""
    rng = np.random.default_rng(77)
    build = np.arange(0, 400_000, dtype=np.int32) * 3
    probe = np.sort(rng.integers(0, 1_200_000, 900_000).astype(np.int32))
    lcol, rcol = build_column(build, None, 65_535, abi.ENC_UNENCODED), build_column(probe, None, 65_535, abi.ENC_FRAME_OF_REFERENCE)
    left, right = DeviceColumn(lcol), DeviceColumn(rcol)
    want = oracle_join(lcol, rcol, abi.JOIN_INNER)
    got = join_hash(left, right, abi.JOIN_INNER)
    assert used_rank_table() in (1, 2)
    n = want.n_pairs
    assert got.n_pairs == n and got.left[:n].tobytes() == want.left[:n].tobytes() and got.right[:n].tobytes() == want.right[:n].tobytes()
    lib = abi.load_library()
    lib.hy_debug_join_lane_ordered_atomics.restype = int
    verdict = lib.hy_debug_join_lane_ordered_atomics()
    assert verdict in (1, 2)
    print("lane-ordered LDS atomics:", "yes" if verdict == 1 else "no (match-any ranking)")


def test_shutdown_releases_the_thread_state_and_the_library_keeps_working(device):
    """hy_shutdown frees what the calling thread holds (scratch, pools, pinned staging, the join mailbox, profiling events);
    everything is allocated again on demand."""
    rng = np.random.default_rng(5)
    lcol = build_column(rng.integers(0, 5_000, 60_000).astype(np.int32), None, 20_000, abi.ENC_DICTIONARY)
    rcol = build_column(rng.integers(0, 5_000, 90_000).astype(np.int32), None, 20_000, abi.ENC_UNENCODED)
    left, right = DeviceColumn(lcol), DeviceColumn(rcol)
    first = join_hash(left, right, abi.JOIN_INNER)
    for _ in range(3):
        abi.check(device.hy_shutdown())
        again = join_hash(left, right, abi.JOIN_INNER)
        n = first.n_pairs
        assert again.n_pairs == n > 0 and again.left[:n].tobytes() == first.left[:n].tobytes() and again.right[:n].tobytes() == first.right[:n].tobytes()


@pytest.mark.parametrize("np_type", [np.int32, np.int64, np.float32, np.float64], ids=lambda t: t.__name__)
def test_hash_map_inputs_of_the_reference(device, np_type):
    """join_hash_types_test.cpp:57-76 as joins of a column with itself: 500 x the value 17 (250 000 pairs: every probe row has 500
    partners) and i^3 for i < 500 (500 pairs) -- bytes equal to the oracle's."""
    same = build_column(np.full(500, 17).astype(np_type), None, 500, abi.ENC_UNENCODED)
    got = check(same, same, abi.JOIN_INNER, context="500 x 17")
    assert got.n_pairs == 250_000
    cubes = build_column((np.arange(500, dtype=np.float64) ** 3).astype(np_type), None, 500, abi.ENC_UNENCODED)
    got = check(cubes, cubes, abi.JOIN_INNER, context="cubes")
    assert got.n_pairs == 500 and (got.left[:500, 1] == got.right[:500, 1]).all()


def used_pkfk():
    lib = abi.load_library()
    lib.hy_debug_join_used_pkfk.restype = C.c_int
    return int(lib.hy_debug_join_used_pkfk())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("build_in_lds", [False, True], ids=["table_in_l2", "bits_in_lds"])
def test_primary_key_foreign_key_probe(device, mode, build_in_lds, options):
    """The kernels of csrc/join_pkfk.hpp (pk_count / pk_scan / pk_emit / pk_cuts: 8192-row tiles, one launch for scan and plan):
    a primary-key build side and a probe column of NULL-free int32 value / FrameOfReference segments (1-, 2- and 4-byte offsets),
    sorted and random, with keys outside the build range, ragged chunks and every radix setting -- pairs and 131 070-element cuts
    equal to the oracle's bytes, and equal to what the general kernels produce (HY_OPT_JOIN_PKFK = 0).
    bits_in_lds: the same joins with the build side's presence bits staged in LDS (pk_count_lds: persistent workgroups, the Bloom filter
    answered from the same bits, found / materialised masks handed to pk_emit<., true>) -- a path large probes against small tables
    take on their own (star joins), forced here for small ones."""
    if build_in_lds:
        options.set(abi.OPT_JOIN_LDS_BUILD_TILES, 1)
    rng = np.random.default_rng(300 + mode)
    n_build = 30000
    i = np.arange(1, n_build + 1, dtype=np.int64)
    sparse = (((i >> 3) << 5) | (i & 7)).astype(np.int32)
    dense_negative = np.arange(n_build, dtype=np.int32) - 11000
    for name, keys, n_probe in (("sparse", sparse, 300_000), ("dense_negative", dense_negative, 70_001)):
        base = rng.choice(keys, n_probe).astype(np.int32)
        outside = rng.random(n_probe) < 0.04
        base[outside] = rng.integers(int(keys.min()) - 300, int(keys.max()) + 300, int(outside.sum())).astype(np.int32)
        alias = rng.random(n_probe) < 0.02     # no build keys, but a build key's Bloom filter bit (key & 0xFFFFF): materialised, they move the PosList cuts
        base[alias] = (rng.choice(keys, int(alias.sum())).astype(np.int64) + (1 << 20) * rng.integers(1, 3, int(alias.sum()))).astype(np.int32)
        narrow = (np.sort(rng.choice(keys[:200], n_probe))).astype(np.int32)          # FrameOfReference offsets of one byte
        for pname, pvalues, probe_encoding, chunk in (("sorted", np.sort(base), abi.ENC_FRAME_OF_REFERENCE, 65535), ("random", base, abi.ENC_FRAME_OF_REFERENCE, 20000),
                                                      ("values", base, abi.ENC_UNENCODED, 65535), ("narrow", narrow, abi.ENC_FRAME_OF_REFERENCE, 8192 * 3 + 5)):
            build = build_column(keys, None, 4096, abi.ENC_UNENCODED)
            probe = build_column(pvalues, None, chunk, probe_encoding)
            for radix_bits in (None, 0, 1, 5, 8):
                context = f"pk mode {mode} {name} probe {pname} radix {radix_bits}"
                args = (probe, build) if mode in SEMI or mode == abi.JOIN_LEFT else (build, probe)
                got = check(*args, mode, radix_bits, context)
                assert used_pkfk() == (2 if build_in_lds and radix_bits != 8 else 1), context   # (256 partitions: the workgroup's cells would not fit beside the bits)
                if radix_bits in (None, 5) and not build_in_lds:
                    options.set(abi.OPT_JOIN_PKFK, 0)
                    general = check(*args, mode, radix_bits, context + " general kernels")
                    options.reset(abi.OPT_JOIN_PKFK)
                    assert used_pkfk() == 0
                    n = got.n_pairs
                    assert general.n_pairs == n and general.left[:n].tobytes() == got.left[:n].tobytes()


def test_primary_key_hint(device, options):
    """The first join over a resident primary-key column leaves its key range behind (hy_column::join_hint); later joins fill the rank
    table in ONE pass sized by the hint, check every key against it and confirm it when the join has finished.  Same bytes either way;
    a hint that does not hold (HY_OPT_JOIN_BREAK_HINT shrinks it) is dropped and the join runs again with the two-pass build."""
    rng = np.random.default_rng(8)
    keys = (np.arange(100_000, dtype=np.int32) * 2 + 40)
    build_host = build_column(keys, None, 65535, abi.ENC_UNENCODED)
    probe_host = build_column(np.sort(rng.integers(0, 200_200, 500_000).astype(np.int32)), None, 65535, abi.ENC_FRAME_OF_REFERENCE)
    want = oracle_join(build_host, probe_host, abi.JOIN_INNER)
    lib = abi.load_library()
    lib.hy_debug_join_build_was_hinted.restype = C.c_int
    for broken in (False, True):
        build, probe = DeviceColumn(build_host), DeviceColumn(probe_host)
        first = join_hash(build, probe, abi.JOIN_INNER)
        assert lib.hy_debug_join_build_was_hinted() == 0
        assert_join_equal(first, want, abi.JOIN_INNER, "first join")
        if broken:
            options.set(abi.OPT_JOIN_BREAK_HINT, 1)
        second = join_hash(build, probe, abi.JOIN_INNER)
        assert lib.hy_debug_join_build_was_hinted() == (2 if broken else 1)     # 2: the hinted attempt was discarded
        assert_join_equal(second, want, abi.JOIN_INNER, f"second join, broken hint {broken}")
        if broken:
            options.reset(abi.OPT_JOIN_BREAK_HINT)
        third = join_hash(build, probe, abi.JOIN_SEMI if False else abi.JOIN_INNER)
        assert lib.hy_debug_join_build_was_hinted() == (0 if broken else 1)     # a dropped hint stays dropped
        assert_join_equal(third, want, abi.JOIN_INNER, "third join")
        assert join_hash_count(build, probe, abi.JOIN_INNER) == want.n_pairs
    # The build side's Bloom filter decides which partner-less probe rows count as materialised (they move the 131 070-element cuts,
    # join_hash_steps.hpp:354-358).  The two-pass build keeps it as one byte per bit, the hinted one-pass build folds the rank table's
    # presence words into 2^20 bits: probe keys that are no build keys but share their low 20 bits with one, and keys that do not.
    keys = np.arange(450_000, dtype=np.int32) * 3 + 40                 # 40 .. 1 350 037: more than 2^20 apart
    alias = keys[:90_000] + (1 << 20)                                  # (2^20 mod 3 == 1: none of them is a build key, each has a build key's filter bit)
    probe_values = np.sort(np.concatenate([rng.choice(keys, 400_000), rng.choice(alias, 200_000), rng.integers(41, 1_350_000, 100_000).astype(np.int32)]).astype(np.int32))
    build_host = build_column(keys, None, 65535, abi.ENC_UNENCODED)
    probe_host = build_column(probe_values, None, 65535, abi.ENC_FRAME_OF_REFERENCE)
    build, probe = DeviceColumn(build_host), DeviceColumn(probe_host)
    for radix_bits in (1, None):
        want = oracle_join(build_host, probe_host, abi.JOIN_INNER, radix_bits)
        assert want.c.n_slices > (1 << int(want.c.radix_bits))        # (partitions with cuts inside)
        for attempt in range(2):
            got = join_hash(build, probe, abi.JOIN_INNER, radix_bits)
            assert lib.hy_debug_join_build_was_hinted() == (0 if radix_bits == 1 and attempt == 0 else 1)
            assert_join_equal(got, want, abi.JOIN_INNER, f"filter aliases, radix {radix_bits}, attempt {attempt}")


@pytest.mark.parametrize("workgroups_per_cu", [0, 1, 4])
def test_hinted_fill_kernels(device, options, workgroups_per_cu):
    """The one-pass checked fill of a hinted build side, as short-lived workgroups (one per slice, HY_OPT_JOIN_FILL_WGS_PER_CU = 0) and as
    wave by wave (rank_table_fill_waves<1 | 2 | 4>: every wave a run of 512-row steps, read a step ahead, built in its own LDS window):
    int32 values and FrameOfReference offsets of every width, ragged chunks (steps of 3 and 511 rows), sparse stretches (steps whose
    keys span more table words than the window), one step and several steps per wave -- and a column that is NOT sorted although its
    hint says so: the verdict must catch it."""
    options.set(abi.OPT_JOIN_FILL_WGS_PER_CU, workgroups_per_cu)
    lib = abi.load_library()
    rng = np.random.default_rng(400 + workgroups_per_cu)
    n = 3_300_000 if workgroups_per_cu else 700_000   # (6 464 steps: seven per wave with one workgroup per CU, two with four)
    dense = np.arange(n, dtype=np.int32) * 2 - 50_000                                   # 4-byte values; as FoR: 2-byte offsets
    stepped = (np.arange(n, dtype=np.int64) // 8 * 32 + np.arange(n) % 8).astype(np.int32)   # dbgen's sparse keys: 8 of every 32
    sparse_tail = np.concatenate([np.arange(n - 20_000, dtype=np.int64), (n - 20_000) + np.arange(20_000, dtype=np.int64) * 37]).astype(np.int32)   # the last slices: 37 key values per key
    small = np.arange(200_000, dtype=np.int32) % 120 + np.arange(200_000, dtype=np.int32) // 120 * 128    # 1-byte FoR offsets (blocks of 2048 rows span < 256)... keys ascending
    cases = [("dense int32", dense, abi.ENC_UNENCODED, 65535), ("dense FoR", dense, abi.ENC_FRAME_OF_REFERENCE, 65535), ("stepped int32", stepped, abi.ENC_UNENCODED, 65535),
             ("ragged chunks", dense[:500_000], abi.ENC_UNENCODED, 8195), ("sparse tail", sparse_tail, abi.ENC_UNENCODED, 65535),
             ("wide FoR", sparse_tail, abi.ENC_FRAME_OF_REFERENCE, 65535)]
    for name, keys, encoding, chunk in cases:
        build_host = build_column(keys, None, chunk, encoding)
        probe_keys = np.sort(np.concatenate([rng.choice(keys, len(keys) + 50_000), rng.integers(int(keys[0]) - 100, int(keys[-1]) + 100, 50_000).astype(np.int32)]).astype(np.int32))   # (more probe rows than build rows: JoinHash builds over the smaller side)
        probe_host = build_column(probe_keys, None, 65535, abi.ENC_FRAME_OF_REFERENCE)
        build, probe = DeviceColumn(build_host), DeviceColumn(probe_host)
        want = oracle_join(build_host, probe_host, abi.JOIN_INNER)
        for attempt in range(3):
            got = join_hash(build, probe, abi.JOIN_INNER)
            assert lib.hy_debug_join_build_was_hinted() == (0 if attempt == 0 else 1), name
            assert_join_equal(got, want, abi.JOIN_INNER, f"{name}, attempt {attempt}, fill workgroups per CU {workgroups_per_cu}")
    # a column that stops being sorted behind its hint: the build column lives in CALLER-owned device memory (HY_MEM_DEVICE), the first
    # join leaves the hint, then two keys swap places in that memory (what "encoded segments are immutable" rules out -- the check
    # must hold anyway): the verdict says unsorted, nothing is written, the join runs again on the two-pass build
    keys = np.arange(600_000, dtype=np.int32) * 2
    chunk = 65520   # (chunk buffers 16-byte aligned inside the one allocation: the checked fill reads 16 bytes per load)
    probe_host = build_column(np.sort(rng.choice(keys, 700_000)).astype(np.int32), None, 65535, abi.ENC_FRAME_OF_REFERENCE)   # (more probe rows than build rows)
    probe = DeviceColumn(probe_host)
    pointer, _ = _device_buffer(lib, keys.shape, np.int32, 0)
    abi.check(lib.hy_memcpy_h2d(pointer, keys.ctypes.data, keys.nbytes))
    n_chunks = (len(keys) + chunk - 1) // chunk
    segments = (abi.Segment * n_chunks)()
    for c in range(n_chunks):
        rows = min(chunk, len(keys) - c * chunk)
        segments[c].encoding, segments[c].data_type, segments[c].size, segments[c].width = abi.ENC_UNENCODED, abi.TYPE_INT, rows, 4
        segments[c].data = pointer + 4 * c * chunk
    handle = C.c_void_p()
    abi.check(lib.hy_column_create(segments, n_chunks, abi.MEM_DEVICE, C.byref(handle)))

    class Handle:   # (what join_hash wants of a DeviceColumn)
        pass
    build = Handle()
    build.handle, build.rows, build.n_chunks = handle, len(keys), n_chunks
    sorted_host = build_column(keys, None, chunk, abi.ENC_UNENCODED)
    first = join_hash(build, probe, abi.JOIN_INNER)
    assert_join_equal(first, oracle_join(sorted_host, probe_host, abi.JOIN_INNER), abi.JOIN_INNER, "caller-owned build column")
    swapped = keys.copy()
    swapped[[123_456, 400_000]] = swapped[[400_000, 123_456]]
    abi.check(lib.hy_memcpy_h2d(pointer, swapped.ctypes.data, swapped.nbytes))
    got = join_hash(build, probe, abi.JOIN_INNER)
    assert lib.hy_debug_join_build_was_hinted() == 2          # the hinted attempt was discarded
    assert_join_equal(got, oracle_join(build_column(swapped, None, chunk, abi.ENC_UNENCODED), probe_host, abi.JOIN_INNER), abi.JOIN_INNER, "unsorted column behind a sorted column's hint")
    abi.check(lib.hy_column_destroy(handle))
    lib.hy_device_free(pointer)


@pytest.mark.parametrize("mode", SEMI)
def test_semi_join_over_a_sorted_build_side_with_duplicates(device, mode):
    """Semi / Anti joins without secondary predicates only ask whether a key exists (the reference's ExistenceOnly hash table,
    join_hash_steps.hpp:97-236): a sorted, dense build column WITH duplicate keys -- lineitem's order keys, the build side of the
    reference's BM_HashSemiProbeRelationSmaller -- still gets a rank table, of which only the presence bits are read."""
    rng = np.random.default_rng(61)
    orders = (np.arange(1, 40_001, dtype=np.int32) * 3)
    lineitem = np.repeat(orders[rng.random(40_000) < 0.8], 3)        # four orders in five have lineitems, three each
    probe = build_column(orders, None, 65535, abi.ENC_UNENCODED)
    build = build_column(lineitem, None, 65535, abi.ENC_FRAME_OF_REFERENCE)
    for radix_bits in (None, 0, 4):
        got = check(probe, build, mode, radix_bits, f"existence only, mode {mode} radix {radix_bits}")
        assert used_rank_table() == 2 and used_pkfk() == 1
        assert got.n_pairs == (int((np.isin(orders, lineitem)).sum()) if mode == abi.JOIN_SEMI else int((~np.isin(orders, lineitem)).sum()))
    again = DeviceColumn(probe), DeviceColumn(build)
    for _ in range(2):   # the second join over the resident build column is filled from the key hint (duplicates allowed for Semi / Anti only)
        got = join_hash(again[0], again[1], mode)
        assert_join_equal(got, oracle_join(probe, build, mode), mode, "resident columns")
    inner = join_hash(again[1], again[0], abi.JOIN_INNER)             # ... and an Inner join of the same columns must not use that table
    assert_join_equal(inner, oracle_join(build, probe, abi.JOIN_INNER), abi.JOIN_INNER, "inner after semi")
