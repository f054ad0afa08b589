"""Parity of the HIP AggregateHash with the CPU oracle: same groups in the same order, same representative rows,
integer results identical; SUM/AVG over float/double within 1e-9 relative (parallel f64 atomics vs. the reference's
sequential double additions -- its own tests only ask for 1e-4, check_table_equal.cpp:34,109-115)."""
import json
import os

import numpy as np
import pytest

from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import aggregate_hash
from hyrise_amd.storage import DeviceColumn
from support import GOLDEN, AggregateCase, build_column, oracle_aggregate

pytestmark = pytest.mark.gpu
FLOAT_TOLERANCE = 1e-9
CASES = json.load(open(os.path.join(os.path.dirname(GOLDEN), "aggregate_cases.json")))["cases"]   # all 79 (string columns as key names / ranks: hyrise_amd/string_keys.py)


def assert_aggregate_equal(got, want, n_aggregates, context=""):
    assert got.n_groups == want.n_groups, f"group count {context}"
    n = want.n_groups
    np.testing.assert_array_equal(got.row_ids[:n], want.row_ids[:n], err_msg=f"group order / representative rows {context}")
    for a in range(n_aggregates):
        g, w = got.column(a), want.column(a)
        for x, y in zip(g, w):
            if x is None or y is None:
                assert x is None and y is None, f"NULL mismatch aggregate {a} {context}"
            elif isinstance(y, float):
                assert abs(x - y) <= FLOAT_TOLERANCE * max(1.0, abs(y)), f"aggregate {a}: {x} vs {y} {context}"
            else:
                assert x == y, f"aggregate {a}: {x} vs {y} {context}"


def run_both(groupby_hosts, aggregate_hosts, context=""):
    cache = {}

    def dev(col):
        if id(col) not in cache:
            cache[id(col)] = DeviceColumn(col)
        return cache[id(col)]

    got = aggregate_hash([dev(c) for c in groupby_hosts], [(f, dev(c) if c is not None else None) for f, c in aggregate_hosts])
    want = oracle_aggregate(groupby_hosts, aggregate_hosts)
    assert_aggregate_equal(got, want, len(aggregate_hosts), context)
    return got


@pytest.mark.parametrize("case", CASES, ids=[f"L{c['line']}" for c in CASES])
def test_reference_aggregate_fixture_on_device(device, case):
    columns = AggregateCase(case)
    if not columns.runnable:
        pytest.skip("COUNT(*) without GROUP BY and without a column to take the table's shape from")
    run_both(columns.groupby, columns.aggregates, f"aggregate_test.cpp:{case['line']}")


def test_group_order_and_immediate_key(device):
    keys = np.array([9_000_000, -5, 70_000, -5, 9_000_000, 123, 70_000, 0], dtype=np.int32)
    values = np.arange(8, dtype=np.int32)
    run_both([build_column(keys, None, 3, abi.ENC_UNENCODED)], [(abi.AGG_SUM, build_column(values, None, 3, abi.ENC_UNENCODED))])
    dense = np.array([5, 3, 0, 4, 3, 0, 5, 3], dtype=np.int32)
    nulls = np.array([0, 0, 1, 0, 0, 0, 0, 0], dtype=bool)
    got = run_both([build_column(dense, nulls, 3, abi.ENC_UNENCODED)], [(abi.AGG_COUNT, None)])
    assert got.column(0) == [1, 1, 3, 1, 2]


@pytest.mark.parametrize("n_groups", [3, 700, 40_000])
def test_random_groups(device, n_groups):
    """Few groups (all in the LDS tables), more groups than LDS slots, and many groups (global table, retry path)."""
    rng = np.random.default_rng(n_groups)
    n, chunk = 300_000, 65535
    k1 = rng.integers(0, n_groups, n).astype(np.int32) * 7919           # sparse keys: no immediate-key shortcut
    k2 = rng.integers(0, 3, n).astype(np.int64)
    k1_null = rng.random(n) < 0.01
    ints = rng.integers(-1000, 1000, n).astype(np.int32)
    longs = rng.integers(-10**12, 10**12, n).astype(np.int64)
    floats = (rng.random(n) * 1000).astype(np.float32)
    doubles = rng.random(n) * 1e6
    vnull = rng.random(n) < 0.05
    g1 = build_column(k1, k1_null, chunk, abi.ENC_DICTIONARY)
    g2 = build_column(k2, None, chunk, abi.ENC_UNENCODED)
    ci = build_column(ints, vnull, chunk, abi.ENC_FRAME_OF_REFERENCE)
    cl = build_column(longs, None, chunk, abi.ENC_UNENCODED)
    cf = build_column(floats, vnull, chunk, abi.ENC_DICTIONARY)
    cd = build_column(doubles, None, chunk, abi.ENC_UNENCODED)
    aggregates = [(abi.AGG_SUM, ci), (abi.AGG_MIN, ci), (abi.AGG_MAX, cl), (abi.AGG_SUM, cf), (abi.AGG_AVG, cd), (abi.AGG_COUNT, ci),
                  (abi.AGG_COUNT, None), (abi.AGG_MIN, cf)]
    run_both([g1], aggregates, f"{n_groups} groups, 1 key")
    run_both([g1, g2], aggregates[:5] + [(abi.AGG_ANY, g2)], f"{n_groups} groups, 2 keys")
    run_both([], aggregates, "no GROUP BY")
    run_both([g2, g1], [], "DISTINCT")


def test_tpch_q1_core(device):
    """Q1 core (config 4 of BASELINE.json, SF 0.05): GROUP BY l_returnflag, l_linestatus (1-character strings, passed as
    their AggregateKey names) with SUM / AVG / COUNT over dictionary-encoded float columns."""
    data = tpch.TpchData(scale_factor=0.05, seed=3)
    flag = storage.make_column(data.l_returnflag, None, abi.ENC_DICTIONARY)
    status = storage.make_column(data.l_linestatus, None, abi.ENC_DICTIONARY)
    qty = storage.make_column(data.l_quantity, None, abi.ENC_DICTIONARY)
    price = storage.make_column(data.l_extendedprice, None, abi.ENC_DICTIONARY)
    disc = storage.make_column(data.l_discount, None, abi.ENC_DICTIONARY)
    got = run_both([flag, status], [(abi.AGG_SUM, qty), (abi.AGG_SUM, price), (abi.AGG_AVG, qty), (abi.AGG_AVG, price), (abi.AGG_AVG, disc),
                                    (abi.AGG_COUNT, None)], "Q1 core")
    assert got.n_groups == 4
    assert sum(got.column(5)) == data.n_lineitems


def test_count_distinct_and_stddev(device):
    """COUNT(DISTINCT) = groups of (GROUP BY columns, column) counted per outer group; STDDEV_SAMP from n, sum, sum of
    squares (the reference runs Welford's recurrence, abstract_aggregate_operator.hpp:83-113): 1e-9 relative."""
    rng = np.random.default_rng(77)
    n = 150_000
    key_a = rng.integers(0, 7, n).astype(np.int32)
    key_b = rng.integers(-3, 3, n).astype(np.int32)
    ints = rng.integers(0, 400, n).astype(np.int32)
    floats = (rng.integers(0, 50, n) / 4.0).astype(np.float32)
    doubles = rng.normal(1000.0, 25.0, n)
    int_nulls = rng.random(n) < 0.1
    key_nulls = rng.random(n) < 0.02
    for chunk, encoding in ((65535, abi.ENC_DICTIONARY), (9000, abi.ENC_UNENCODED)):
        ka = build_column(key_a, key_nulls, chunk, encoding)
        kb = build_column(key_b, None, chunk, encoding)
        ci = build_column(ints, int_nulls, chunk, encoding)
        cf = build_column(floats, None, chunk, encoding)
        cd = build_column(doubles, None, chunk, abi.ENC_UNENCODED)
        aggregates = [(abi.AGG_COUNT_DISTINCT, ci), (abi.AGG_STDDEV_SAMP, ci), (abi.AGG_COUNT_DISTINCT, cf), (abi.AGG_STDDEV_SAMP, cd),
                      (abi.AGG_SUM, ci), (abi.AGG_COUNT, None)]
        run_both([ka, kb], aggregates, f"distinct/stddev chunk {chunk} enc {encoding}")
        run_both([ka], aggregates[:4], f"distinct/stddev one key chunk {chunk} enc {encoding}")
        run_both([], aggregates[:4], f"distinct/stddev no GROUP BY chunk {chunk} enc {encoding}")


@pytest.mark.parametrize("n", [300_001, 65_535 * 2 + 3])
def test_direct_mapped_groups_and_plain_value_columns(device, n):
    """The fast paths of aggregate_rows: every GROUP BY column a dictionary segment with a small combined domain (the
    combined value-id code is the slot; codes above 31 take the LDS bitmap, more than four groups per slice take pass 3),
    NULL keys, chunk ends that are no multiple of four rows, and unencoded 4-byte aggregate columns with and without a null
    bitmap (wide loads, the four rows' NULL bits from one byte of the bitmap) next to 8-byte ones (generic decoder)."""
    rng = np.random.default_rng(n)
    chunk = 65_535
    k1 = rng.integers(0, 40, n).astype(np.int32) * 3
    k2 = rng.integers(0, 5, n).astype(np.int64)
    k1_null, k2_null = rng.random(n) < 0.01, rng.random(n) < 0.02
    g1 = build_column(k1, k1_null, chunk, abi.ENC_DICTIONARY)     # 41 x 6 codes = 246 <= 256: direct-mapped
    g2 = build_column(k2, k2_null, chunk, abi.ENC_DICTIONARY)
    few = build_column(rng.integers(0, 2, n).astype(np.int32), None, chunk, abi.ENC_DICTIONARY)   # 3 x ... : at most four groups with `few` alone
    ints = build_column(rng.integers(-50_000, 50_000, n).astype(np.int32), None, chunk, abi.ENC_UNENCODED)
    floats = build_column((rng.random(n) * 100).astype(np.float32), None, chunk, abi.ENC_UNENCODED)
    nullable = build_column((rng.random(n) * 100).astype(np.float32), rng.random(n) < 0.1, chunk, abi.ENC_UNENCODED)
    nullable_ints = build_column(rng.integers(-9_000, 9_000, n).astype(np.int32), rng.random(n) < 0.3, chunk, abi.ENC_UNENCODED)
    doubles = build_column(rng.normal(0.0, 1e3, n), None, chunk, abi.ENC_UNENCODED)
    longs = build_column(rng.integers(-10**14, 10**14, n).astype(np.int64), None, chunk, abi.ENC_UNENCODED)
    aggregates = [(abi.AGG_SUM, ints), (abi.AGG_AVG, ints), (abi.AGG_MIN, ints), (abi.AGG_MAX, floats), (abi.AGG_SUM, floats), (abi.AGG_AVG, floats),
                  (abi.AGG_SUM, nullable), (abi.AGG_COUNT, nullable)]
    more = [(abi.AGG_MIN, floats), (abi.AGG_MAX, ints), (abi.AGG_SUM, doubles), (abi.AGG_MAX, longs), (abi.AGG_STDDEV_SAMP, floats), (abi.AGG_COUNT, None),
            (abi.AGG_MIN, nullable_ints)]
    got = run_both([g1, g2], aggregates, "246 codes")
    assert got.n_groups == 41 * 6
    run_both([g2, g1], more, "246 codes, other aggregates")
    got = run_both([few], aggregates, "two groups")
    assert got.n_groups == 2
    run_both([few, g2], more, "12 codes")
    run_both([g2], aggregates[:6] + more[:2], "six groups: two of them behind the dense four")


def test_stddev_of_large_values_with_a_small_spread(device):
    """STDDEV_SAMP where |mean| >> spread (values around 1e9, spread 1): the plain sums of x and x^2 cancel catastrophically
    (x^2 ~ 1e18: one ulp is 128).  The device accumulates values shifted by a value of the column and lands within 1e-9 of the
    exact standard deviation; the reference's Welford recurrence (abstract_aggregate_operator.hpp:83-113, the oracle) rounds
    its running mean at these magnitudes (1e-7 .. 2e-4 relative here), so both are checked against the exact value."""
    rng = np.random.default_rng(12)
    n = 200_000
    keys = rng.integers(0, 5, n).astype(np.int32) * 1000
    key_column = build_column(keys, None, 65535, abi.ENC_UNENCODED)
    for values in ((1_000_000_000 + rng.integers(0, 3, n)).astype(np.int32), (1e9 + rng.random(n)).astype(np.float64),
                   (5_000_000_000_000 + rng.integers(-2, 3, n)).astype(np.int64)):
        nulls = rng.random(n) < 0.01
        column = build_column(values, nulls, 65535, abi.ENC_UNENCODED)
        got = aggregate_hash([DeviceColumn(key_column)], [(abi.AGG_STDDEV_SAMP, DeviceColumn(column))])
        want = oracle_aggregate([key_column], [(abi.AGG_STDDEV_SAMP, column)])
        assert got.n_groups == want.n_groups == 5
        np.testing.assert_array_equal(got.row_ids[:5], want.row_ids[:5])
        group_keys = [int(keys[int(r[0]) * 65535 + int(r[1])]) for r in want.row_ids[:5]]
        for g, key in enumerate(group_keys):
            member = (keys == key) & ~nulls
            shifted = (values[member] - values[member][0]).astype(np.longdouble)   # exact: the differences are small
            exact = float(np.sqrt(((shifted - shifted.mean()) ** 2).sum() / (member.sum() - 1)))
            assert abs(got.column(0)[g] - exact) <= 1e-9 * exact, f"device {got.column(0)[g]} vs exact {exact}"
            assert abs(want.column(0)[g] - exact) <= 1e-3 * exact, f"oracle {want.column(0)[g]} vs exact {exact}"   # (its running mean has an ulp of 1e-3 at 5e12)


def aggregate_path():
    lib = abi.load_library()
    lib.hy_debug_aggregate_path.restype = int
    return lib.hy_debug_aggregate_path()


def test_more_aggregates_than_one_pass_has_accumulators(device):
    """Eight device accumulators per pass (STDDEV_SAMP takes two): a plan with more -- AggregateHash takes any number,
    aggregate_hash.cpp:1016-1176 -- runs in several passes over the same GROUP BY columns; the groups, their order and every column equal the
    oracle's.  Few groups (LDS tables) and many (the partitioned path), with NULLs."""
    rng = np.random.default_rng(12)
    n = 150_000
    for distinct in (7, 20_000):
        keys = build_column(rng.integers(0, distinct, n).astype(np.int32), rng.random(n) < 0.01, 65535, abi.ENC_DICTIONARY)
        a = build_column(rng.integers(-1000, 1000, n).astype(np.int32), rng.random(n) < 0.05, 65535, abi.ENC_UNENCODED)
        b = build_column(rng.random(n).astype(np.float64) * 100, None, 65535, abi.ENC_UNENCODED)
        c = build_column(rng.integers(0, 50, n).astype(np.int64), None, 65535, abi.ENC_DICTIONARY)
        aggregates = [(abi.AGG_SUM, a), (abi.AGG_AVG, a), (abi.AGG_MIN, a), (abi.AGG_MAX, a), (abi.AGG_COUNT, a), (abi.AGG_STDDEV_SAMP, a), (abi.AGG_SUM, b), (abi.AGG_MIN, b), (abi.AGG_MAX, b),
                      (abi.AGG_STDDEV_SAMP, b), (abi.AGG_SUM, c), (abi.AGG_AVG, c), (abi.AGG_MAX, c), (abi.AGG_COUNT_DISTINCT, c), (abi.AGG_COUNT, None), (abi.AGG_ANY, a), (abi.AGG_MIN, c)]
        got = run_both([keys], aggregates, f"{len(aggregates)} aggregates, {distinct} distinct keys")
        assert distinct * 0.99 < got.n_groups <= distinct + 1   # (run_both compared the groups themselves with the oracle's)


@pytest.fixture
def forced_partitions(options):
    """HY_OPT_AGG_PARTITION_BITS: every aggregate of the test runs the partitioned path (partition -> LDS tables -> one merge)."""
    options.set(abi.OPT_AGG_PARTITION_BITS, 3)
    yield


@pytest.mark.parametrize("case", CASES, ids=[f"L{c['line']}" for c in CASES])
def test_reference_aggregate_fixture_on_the_partitioned_path(device, forced_partitions, case):
    columns = AggregateCase(case)
    if not columns.runnable:
        pytest.skip("COUNT(*) without GROUP BY and without a column to take the table's shape from")
    run_both(columns.groupby, columns.aggregates, f"aggregate_test.cpp:{case['line']} (partitioned)")
    assert aggregate_path() == (3 if columns.groupby else 0)


@pytest.mark.parametrize("n_groups,expected_path", [(100, 0), (5_000, 6), (150_000, 14)])
def test_many_groups_take_the_partitioned_path(device, n_groups, expected_path):
    """More groups than a slice's LDS table holds: aggregate_rows gives up after a few slices and the table is partitioned by
    the hash of its keys (two GROUP BY columns, NULL keys, dictionary / unencoded / FrameOfReference inputs, every mergeable
    function)."""
    rng = np.random.default_rng(n_groups)
    n, chunk = 700_000, 65535
    k1 = (rng.integers(0, n_groups, n).astype(np.int64) * 1_000_003 - 17)            # sparse int64 keys
    k2 = rng.integers(0, 2, n).astype(np.int32)
    k1_null = rng.random(n) < 0.003
    ints = rng.integers(-1000, 1000, n).astype(np.int32)
    floats = (rng.random(n) * 1000).astype(np.float32)
    doubles = rng.random(n) * 1e6
    vnull = rng.random(n) < 0.05
    groupby = [build_column(k1, k1_null, chunk, abi.ENC_DICTIONARY), build_column(k2, None, chunk, abi.ENC_UNENCODED)]
    aggregates = [(abi.AGG_SUM, build_column(ints, vnull, chunk, abi.ENC_FRAME_OF_REFERENCE)), (abi.AGG_AVG, build_column(floats, None, chunk, abi.ENC_DICTIONARY)),
                  (abi.AGG_MIN, build_column(doubles, vnull, chunk, abi.ENC_UNENCODED)), (abi.AGG_MAX, build_column(ints, None, chunk, abi.ENC_UNENCODED)),
                  (abi.AGG_COUNT, None), (abi.AGG_SUM, build_column(doubles, None, chunk, abi.ENC_UNENCODED))]
    got = run_both(groupby, aggregates, f"{n_groups} groups")
    assert got.n_groups > n_groups * 0.9
    assert aggregate_path() == expected_path
    assert finished_on_device() == (1 if got.n_groups > 4096 else 0)


def finished_on_device():
    lib = abi.load_library()
    lib.hy_debug_aggregate_finished_on_device.restype = int
    return lib.hy_debug_aggregate_finished_on_device()


def ragged_column(values, nulls, sizes, encoding):
    """Chunks of the given sizes (a table behind deletes / a partial last chunk in the middle after a merge): RowIDs by search."""
    segments, begin = [], 0
    for size in sizes:
        segments.append(storage.encode_segment(values[begin:begin + size], None if nulls is None else nulls[begin:begin + size], encoding))
        begin += size
    assert begin == len(values)
    return storage.HostColumn(segments, storage.TYPE_OF_NP[values.dtype])


@pytest.mark.parametrize("order", ["first_row", "immediate_key"])
@pytest.mark.parametrize("chunks", ["uniform", "ragged"])
def test_large_results_are_finished_on_the_device(device, order, chunks):
    """More than 4096 groups of COUNT / SUM / AVG / MIN / MAX: ordered (by first row, or by key with the NULL group first:
    aggregate_hash.cpp:388-401, 770-804), turned into RowIDs and typed columns by kernels -- byte for byte the oracle's result; the second
    call over the same columns starts on the path the first one ended on and returns the same."""
    rng = np.random.default_rng(77)
    n, distinct = 400_000, 30_000
    sizes = [65535] * (n // 65535) + [n % 65535] if chunks == "uniform" else [50_000, 1, 65_535, 120_000, 999, n - 50_000 - 1 - 65_535 - 120_000 - 999]
    keys = rng.integers(0, distinct, n).astype(np.int32)
    keys = keys - 15_000 if order == "immediate_key" else keys * 50_021 - (1 << 30)   # dense: the key is the order; sparse: the first row is
    key_nulls = rng.random(n) < 0.002
    ints = rng.integers(-1000, 1000, n).astype(np.int32)
    longs = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    floats = (rng.random(n) * 100 - 50).astype(np.float32)
    doubles = rng.random(n) * 1e6 - 5e5
    vnull = rng.random(n) < 0.3
    groupby = [ragged_column(keys, key_nulls, sizes, abi.ENC_UNENCODED)]
    aggregates = [(abi.AGG_SUM, ragged_column(ints, vnull, sizes, abi.ENC_UNENCODED)), (abi.AGG_AVG, ragged_column(ints, vnull, sizes, abi.ENC_UNENCODED)),
                  (abi.AGG_MIN, ragged_column(ints, vnull, sizes, abi.ENC_UNENCODED)), (abi.AGG_MAX, ragged_column(floats, None, sizes, abi.ENC_DICTIONARY)),
                  (abi.AGG_MIN, ragged_column(doubles, vnull, sizes, abi.ENC_UNENCODED)), (abi.AGG_SUM, ragged_column(longs, None, sizes, abi.ENC_UNENCODED)),
                  (abi.AGG_COUNT, ragged_column(doubles, vnull, sizes, abi.ENC_UNENCODED)), (abi.AGG_COUNT, None)]
    cache = {}

    def dev(col):
        return cache.setdefault(id(col), DeviceColumn(col))

    want = oracle_aggregate(groupby, aggregates)
    for attempt in range(2):
        got = aggregate_hash([dev(c) for c in groupby], [(f, dev(c) if c is not None else None) for f, c in aggregates])
        assert_aggregate_equal(got, want, len(aggregates), f"{order}, {chunks} chunks, call {attempt}")
        assert finished_on_device() == 1 and aggregate_path() > 0
    assert got.n_groups > 29_000


def test_four_byte_columns_travel_in_narrow_records(device):
    """Every GROUP BY column and every aggregate input an int32 / float32 column: the partitioned path's records are 32-bit words
    (partition_rows NARROW).  Negative keys and values, -0.0 and 0.0 float keys in one group, NULL keys and inputs, every function."""
    rng = np.random.default_rng(5)
    n, chunk = 500_000, 65535
    k1 = rng.integers(-20_000, 20_000, n).astype(np.int32)
    k2 = rng.choice(np.array([-0.0, 0.0, 1.5, -2.25], dtype=np.float32), n)
    k1_null = rng.random(n) < 0.004
    k2_null = rng.random(n) < 0.01
    ints = rng.integers(-(1 << 31), (1 << 31) - 1, n).astype(np.int32)
    floats = ((rng.random(n) - 0.5) * 1e6).astype(np.float32)
    vnull = rng.random(n) < 0.2
    int_column = build_column(ints, vnull, chunk, abi.ENC_FRAME_OF_REFERENCE)
    float_column = build_column(floats, vnull, chunk, abi.ENC_DICTIONARY)
    groupby = [build_column(k1, k1_null, chunk, abi.ENC_DICTIONARY), build_column(k2, k2_null, chunk, abi.ENC_UNENCODED)]
    aggregates = [(abi.AGG_SUM, int_column), (abi.AGG_MIN, int_column), (abi.AGG_MAX, float_column), (abi.AGG_AVG, float_column), (abi.AGG_MIN, float_column), (abi.AGG_COUNT, int_column),
                  (abi.AGG_COUNT, None)]
    got = run_both(groupby, aggregates, "narrow records")
    assert got.n_groups > 100_000 and aggregate_path() > 0 and finished_on_device() == 1
    got = run_both(groupby[:1], [(abi.AGG_STDDEV_SAMP, float_column), (abi.AGG_AVG, int_column)], "narrow records, host finish")
    assert aggregate_path() > 0 and finished_on_device() == 0


def test_partitions_with_more_groups_than_their_tables(device, options):
    """Every row its own group at 2^3 forced partitions: the partitions' LDS tables overflow, the rows go to the global table
    directly, the path gives up and partitions more finely -- the result is the same."""
    options.set(abi.OPT_AGG_PARTITION_BITS, 3)
    try:
        n = 200_000
        keys = np.random.default_rng(4).permutation(n).astype(np.int32) * 13
        values = np.arange(n, dtype=np.int64)
        got = run_both([build_column(keys, None, 65535, abi.ENC_UNENCODED)], [(abi.AGG_SUM, build_column(values, None, 65535, abi.ENC_UNENCODED)), (abi.AGG_COUNT, None)])
        assert got.n_groups == n and aggregate_path() == 14
    finally:
        options.reset()


@pytest.mark.parametrize("case", CASES, ids=[f"L{c['line']}" for c in CASES])
def test_reference_aggregate_fixture_on_a_reference_table(device, case):
    """aggregate_test.cpp runs every case a second time on a reference table produced by a pass-through TableScan (:40-79): all columns
    as ReferenceSegments -- explicit PosLists for the even chunks, EntireChunkPosLists for the odd ones -- over the data columns."""
    columns = AggregateCase(case)
    if not columns.runnable:
        pytest.skip("COUNT(*) without GROUP BY and without a column to take the table's shape from")
    devices, references = {}, {}

    def reference_of(column):
        if id(column) not in references:
            lists = [np.stack([np.full(s.size, c, dtype=np.uint32), np.arange(s.size, dtype=np.uint32)], axis=1) if c % 2 == 0 else c for c, s in enumerate(column.segments)]
            host = storage.make_reference_column(column, lists, list(range(column.n_chunks)))
            devices[id(column)] = DeviceColumn(column)
            references[id(column)] = (host, DeviceColumn(host, refs={id(column): devices[id(column)]}))
        return references[id(column)]

    groupby = [reference_of(c) for c in columns.groupby]
    aggregates = [(f, reference_of(c) if c is not None else None) for f, c in columns.aggregates]
    got = aggregate_hash([d for _, d in groupby], [(f, r[1] if r is not None else None) for f, r in aggregates])
    want = oracle_aggregate(columns.groupby, columns.aggregates)                 # the data table's result: same groups, same order, same rows
    on_references = oracle_aggregate([h for h, _ in groupby], [(f, r[0] if r is not None else None) for f, r in aggregates])
    assert_aggregate_equal(on_references, want, len(aggregates), f"oracle, reference table, aggregate_test.cpp:{case['line']}")
    assert_aggregate_equal(got, want, len(aggregates), f"device, reference table, aggregate_test.cpp:{case['line']}")


def used_small_domain():
    lib = abi.load_library()
    lib.hy_debug_aggregate_small_domain.restype = int
    return lib.hy_debug_aggregate_small_domain()


def test_small_domain_kernel(device, options):
    """sd_groups / sd_wide (csrc/aggregate_small.hpp): a handful of groups over dictionary GROUP BY columns, SUM / AVG / COUNT / MIN /
    MAX over dictionary-encoded int / long / float / double columns with 1- and 2-byte value ids -- the TPC-H Q1 shape and its
    neighbours.  Against the oracle: 1 - 3 GROUP BY columns, NULLs in keys and inputs, more than four groups per chunk (the shared-cell
    path), ragged chunks, one-row chunks, negative integers, and the same answers as the generic kernel (HY_OPT_AGG_SMALL_DOMAIN = 0)."""
    rng = np.random.default_rng(91)
    for n, chunk in ((200_000, 65535), (70_001, 8192), (5, 2), (40_000, 40_000), (150_000, 100_000)):   # (the last: chunks of more than 65536 rows -- not the small-domain kernels' shape, the generic kernel answers)
        flags = rng.integers(0, 3, n).astype(np.int32)                       # 3 distinct
        status = rng.integers(0, 2, n).astype(np.int64)                      # 2 distinct
        third = (rng.integers(0, 2, n) * 7).astype(np.int32)
        flag_nulls = rng.random(n) < 0.01
        quantity = rng.integers(1, 51, n).astype(np.float32)                 # 1-byte value ids
        discount = (rng.integers(0, 11, n) / 100.0).astype(np.float64)       # 1-byte value ids, double
        price = (rng.integers(0, min(60_000, max(300, n)), n) * 1.37 + 900.0).astype(np.float32)   # 2-byte value ids where the chunk is large enough
        wide_double = rng.integers(0, max(300, n // 3), n).astype(np.float64) * 0.25
        price_nulls = rng.random(n) < 0.02
        g1 = build_column(flags, flag_nulls, chunk, abi.ENC_DICTIONARY)
        g2 = build_column(status, None, chunk, abi.ENC_DICTIONARY)
        g3 = build_column(third, None, chunk, abi.ENC_DICTIONARY)
        q = build_column(quantity, None, chunk, abi.ENC_DICTIONARY)
        d = build_column(discount, rng.random(n) < 0.03, chunk, abi.ENC_DICTIONARY)
        p = build_column(price, price_nulls, chunk, abi.ENC_DICTIONARY)
        w = build_column(wide_double, None, chunk, abi.ENC_DICTIONARY)
        qi = build_column(rng.integers(-40, 41, n).astype(np.int32), rng.random(n) < 0.05, chunk, abi.ENC_DICTIONARY)                 # int, 1-byte value ids
        ql = build_column(rng.integers(-3, 4, n).astype(np.int64) * (1 << 40), None, chunk, abi.ENC_DICTIONARY)                       # long, 1-byte value ids, sums beyond 2^53
        pi = build_column(rng.integers(-30_000, 30_000, n).astype(np.int32) * 1000, rng.random(n) < 0.02, chunk, abi.ENC_DICTIONARY)  # int, 2-byte value ids where the chunk is large enough
        pl = build_column(rng.integers(-20_000, 20_000, n).astype(np.int64) * (1 << 33) - 5, None, chunk, abi.ENC_DICTIONARY)         # long, 2-byte value ids
        q1 = [(abi.AGG_SUM, q), (abi.AGG_SUM, p), (abi.AGG_AVG, q), (abi.AGG_AVG, p), (abi.AGG_AVG, d), (abi.AGG_COUNT, None)]
        for name, groupby, aggregates in (("q1 shape", [g1, g2], q1), ("one key", [g2], q1[:3] + [(abi.AGG_COUNT, p)]),
                                          ("three keys, 36 codes: the generic kernel", [g1, g2, g3], [(abi.AGG_SUM, w), (abi.AGG_AVG, p), (abi.AGG_COUNT, d), (abi.AGG_COUNT, None)]),
                                          ("two keys, up to nine groups", [g2, g3, g2], [(abi.AGG_SUM, w), (abi.AGG_AVG, p), (abi.AGG_COUNT, d), (abi.AGG_COUNT, None)]),
                                          ("twelve groups with NULL keys", [g1, g3], [(abi.AGG_SUM, w), (abi.AGG_AVG, p), (abi.AGG_COUNT, d), (abi.AGG_COUNT, None)]),
                                          ("no group by", [], [(abi.AGG_SUM, q), (abi.AGG_AVG, w), (abi.AGG_COUNT, None)]),
                                          ("integers and extremes", [g1, g2], [(abi.AGG_SUM, qi), (abi.AGG_AVG, qi), (abi.AGG_MIN, qi), (abi.AGG_MAX, pi), (abi.AGG_SUM, pi), (abi.AGG_MIN, d),
                                                                               (abi.AGG_MAX, d), (abi.AGG_COUNT, None)]),
                                          ("longs", [g2], [(abi.AGG_SUM, ql), (abi.AGG_AVG, ql), (abi.AGG_MAX, ql), (abi.AGG_SUM, pl), (abi.AGG_MIN, pl), (abi.AGG_MAX, pl), (abi.AGG_AVG, pl), (abi.AGG_COUNT, pl)]),
                                          ("extremes of floats", [g1], [(abi.AGG_MIN, p), (abi.AGG_MAX, p), (abi.AGG_MIN, q), (abi.AGG_MAX, q), (abi.AGG_AVG, p)]),
                                          ("extremes of a wide double, no group by", [], [(abi.AGG_MIN, w), (abi.AGG_MAX, w), (abi.AGG_SUM, w)]),
                                          ("twelve groups, integers and extremes", [g1, g3], [(abi.AGG_MIN, p), (abi.AGG_MAX, p), (abi.AGG_SUM, qi), (abi.AGG_AVG, qi), (abi.AGG_MAX, qi), (abi.AGG_MIN, d),
                                                                                              (abi.AGG_COUNT, p)]),
                                          ("twelve groups, a wide long", [g1, g3], [(abi.AGG_SUM, pl), (abi.AGG_MIN, pl), (abi.AGG_MAX, pl), (abi.AGG_SUM, ql), (abi.AGG_MIN, ql)])):
            context = f"small domain, {n} rows in chunks of {chunk}, {name}"
            got = run_both(groupby, aggregates, context)
            widths = {c.segments[0].width for _, c in aggregates if c is not None} | {c.segments[0].width for c in groupby}
            codes = max((int(np.prod([g.segments[k].aux_size + 1 for g in groupby])) for k in range(q.n_chunks)), default=1)
            inputs = {id(c): c.segments[0].width for _, c in aggregates if c is not None}
            narrow, wide = sum(1 for v in inputs.values() if v == 1), sum(1 for v in inputs.values() if v == 2)
            assert used_small_domain() == (1 if widths <= {1, 2} and codes <= 16 and narrow <= 2 and wide <= 1 and len(groupby) <= 2 and chunk <= 65536 else 0), context
            options.set(abi.OPT_AGG_SMALL_DOMAIN, 0)
            generic = run_both(groupby, aggregates, context + " (generic kernel)")
            options.reset(abi.OPT_AGG_SMALL_DOMAIN)
            assert used_small_domain() == 0
            np.testing.assert_array_equal(got.row_ids[:got.n_groups], generic.row_ids[:generic.n_groups])
    # a 2-byte column's value that sixteen rows of one group share inside one chunk: its 4-bit counter overflows, the kernel notices
    # (the counters' sum falls short of the rows counted) and the generic kernel answers
    n = 60_000
    price = (np.arange(n) % 50_000 * 1.37 + 900.0).astype(np.float32)
    price[rng.choice(n, 40, replace=False)] = 123.5
    key = build_column(np.zeros(n, np.int32), None, 65535, abi.ENC_DICTIONARY)
    repeated = build_column(price, None, 65535, abi.ENC_DICTIONARY)
    assert repeated.segments[0].width == 2
    run_both([key], [(abi.AGG_SUM, repeated), (abi.AGG_COUNT, None)], "a value forty times in one group of one chunk")
    assert used_small_domain() == 0
    price[price == 123.5] = 77.25 + np.arange(40, dtype=np.float32)   # (the same column without the repeated value: the kernel keeps it)
    run_both([key], [(abi.AGG_SUM, build_column(price, None, 65535, abi.ENC_DICTIONARY)), (abi.AGG_COUNT, None)], "no value sixteen times")
    assert used_small_domain() == 1
    # shapes the kernel does not take: a GROUP BY column with too many distinct values, STDDEV_SAMP, an unencoded input -- and two it takes now
    ints = build_column(rng.integers(0, 9, 1000).astype(np.int32), None, 500, abi.ENC_DICTIONARY)
    many = build_column(rng.integers(0, 40, 1000).astype(np.int32), None, 500, abi.ENC_DICTIONARY)
    few = build_column(rng.integers(0, 2, 1000).astype(np.int32), None, 500, abi.ENC_DICTIONARY)
    floats = build_column(rng.integers(0, 9, 1000).astype(np.float32), None, 500, abi.ENC_DICTIONARY)
    run_both([few], [(abi.AGG_SUM, ints)], "integer input")
    assert used_small_domain() == 1
    run_both([many], [(abi.AGG_SUM, floats)], "forty groups")
    assert used_small_domain() == 0
    run_both([few], [(abi.AGG_SUM, floats), (abi.AGG_MIN, floats)], "MIN")
    assert used_small_domain() == 1
    run_both([few], [(abi.AGG_STDDEV_SAMP, floats)], "STDDEV_SAMP")
    assert used_small_domain() == 0
    run_both([few], [(abi.AGG_SUM, build_column(rng.integers(0, 9, 1000).astype(np.int32), None, 500, abi.ENC_UNENCODED))], "unencoded input")
    assert used_small_domain() == 0


def test_aggregate_result_in_device_memory(device):
    """hy_aggregate_result.mem = HY_MEM_DEVICE: the group rows, values and NULL flags arrive in the caller's device buffers (an operator
    chain that ends on the device) and are the host-memory result's bytes; a result that does not fit reports the groups it needs."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(17)
    n = 50_000
    keys = build_column(rng.integers(0, 300, n).astype(np.int32), rng.random(n) < 0.01, 7000, abi.ENC_DICTIONARY)
    ints = build_column(rng.integers(-1000, 1000, n).astype(np.int32), rng.random(n) < 0.1, 7000, abi.ENC_FRAME_OF_REFERENCE)
    floats = build_column(rng.random(n).astype(np.float32), None, 7000, abi.ENC_UNENCODED)
    devices = {id(c): DeviceColumn(c) for c in (keys, ints, floats)}
    spec = [(abi.AGG_SUM, ints), (abi.AGG_AVG, floats), (abi.AGG_MIN, ints), (abi.AGG_MAX, floats), (abi.AGG_COUNT, None)]
    want = aggregate_hash([devices[id(keys)]], [(f, devices[id(c)] if c is not None else None) for f, c in spec])
    lib = abi.load_library()
    capacity = 512
    dev = torch.device("cuda", 0)
    rows = torch.zeros((capacity, 2), dtype=torch.int32, device=dev)
    values = [torch.zeros(capacity, dtype=torch.int64, device=dev) for _ in spec]
    nulls = [torch.zeros(capacity, dtype=torch.uint8, device=dev) for _ in spec]
    columns = (abi.AggregateColumn * len(spec))()
    for a in range(len(spec)):
        columns[a].values, columns[a].is_null = values[a].data_ptr(), nulls[a].data_ptr()
    result = abi.AggregateResult()
    result.mem, result.group_capacity, result.group_row_ids, result.columns = abi.MEM_DEVICE, capacity, rows.data_ptr(), columns
    garr = (C.c_void_p * 1)(devices[id(keys)].handle)
    specs = (abi.AggregateSpec * len(spec))()
    for i, (function, column) in enumerate(spec):
        specs[i].function = function
        specs[i].column = devices[id(column)].handle if column is not None else None
    abi.check(lib.hy_aggregate_hash(garr, 1, specs, len(spec), C.byref(result)))
    torch.cuda.synchronize()
    groups = int(result.n_groups)
    assert groups == want.n_groups == 301
    assert rows[:groups].cpu().numpy().view(np.uint32).tobytes() == want.row_ids[:groups].tobytes()
    for a in range(len(spec)):
        assert columns[a].data_type == want.columns[a].data_type
        width = 4 if columns[a].data_type in (abi.TYPE_INT, abi.TYPE_FLOAT) else 8
        assert values[a].cpu().numpy().tobytes()[:width * groups] == want.raw[a].tobytes()[:width * groups], f"aggregate {a}"
        assert nulls[a][:groups].cpu().numpy().tobytes() == want.nulls[a][:groups].tobytes()
    result.group_capacity = 100   # too small: nothing is promised about the buffers, the group count is reported
    assert lib.hy_aggregate_hash(garr, 1, specs, len(spec), C.byref(result)) == abi.ERR_CAPACITY and int(result.n_groups) == 301


@pytest.mark.parametrize("selectivity", [0.99, 0.5, 0.02])
def test_aggregate_behind_a_scan_reads_the_shared_pos_list_once(device, selectivity):
    """AggregateHash over the reference table a TableScan leaves (every column behind the SAME per-chunk PosList, as tpch.run_q1 builds it):
    aggregate_rows stages a slice's offsets in LDS -- sixteen bits and three steps of at most 31 per four rows -- when the list is ascending and
    dense enough (0.99, 0.5), and reads the list from memory per column when a step does not fit (0.02: steps of 50 on average).  Both against
    numpy; a ragged last chunk; a second group of columns behind ANOTHER list of the same shape (compared by pointer: not the cached one)."""
    import torch
    from hyrise_amd.distributed import HipExecutor
    from hyrise_amd.operators import make_predicate
    rng = np.random.default_rng(int(selectivity * 100))
    n, chunk = 150_001, 40_000
    flag = rng.integers(0, 3, n).astype(np.int32)
    other = rng.integers(0, 5, n).astype(np.int32)
    value = rng.integers(-1000, 1000, n).astype(np.int32)
    price = rng.integers(0, 50_000, n).astype(np.int32)
    pick = (rng.random(n) < selectivity).astype(np.int32)
    host = {"flag": storage.make_column(flag, None, abi.ENC_DICTIONARY, chunk), "other": storage.make_column(other, None, abi.ENC_DICTIONARY, chunk),
            "value": storage.make_column(value, None, abi.ENC_UNENCODED, chunk), "price": storage.make_column(price, None, abi.ENC_DICTIONARY, chunk),
            "pick": storage.make_column(pick, None, abi.ENC_UNENCODED, chunk)}
    columns = {name: DeviceColumn(column) for name, column in host.items()}
    ex = HipExecutor(torch.device("cuda:0"))
    predicate = make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, 1)
    lists = ex.scan_chunked(columns["pick"], predicate)
    again = ex.scan_chunked(columns["pick"], predicate)                       # the same rows in another buffer
    ref = {name: ex.reference_column_chunked(column, lists) for name, column in columns.items()}
    ref_again = {name: ex.reference_column_chunked(column, again) for name, column in columns.items()}
    keep = pick == 1

    def expected(keys):
        want = {}
        for row in np.flatnonzero(keep):
            k = tuple(int(c[row]) for c in keys)
            count, total, top, cheapest = want.get(k, (0, 0, -(1 << 40), 1 << 40))
            want[k] = (count + 1, total + int(value[row]), max(top, int(price[row])), min(cheapest, int(value[row])))
        return want

    for groupby, keys, source in (([ref["flag"], ref["other"]], [flag, other], ref), ([ref["flag"]], [flag], ref_again), ([ref_again["other"]], [other], ref)):
        aggregates = [(abi.AGG_COUNT, None), (abi.AGG_SUM, source["value"]), (abi.AGG_MAX, source["price"]), (abi.AGG_MIN, ref_again["value"])]
        result = aggregate_hash(groupby, aggregates + [(abi.AGG_MIN, g) for g in groupby], group_capacity=64)
        got = {tuple(result.column(4 + g)[i] for g in range(len(groupby))): tuple(result.column(a)[i] for a in range(4)) for i in range(result.n_groups)}
        assert got == expected(keys)
