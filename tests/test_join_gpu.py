"""Parity of the HIP JoinHash with the CPU oracle: the concatenated PosList pairs and the 131 070-element slice
boundaries must be bit-identical (1 GPU: identical order, not just identical multisets)."""
import numpy as np
import pytest

from hyrise_amd import abi, storage
from hyrise_amd.operators import join_hash
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
