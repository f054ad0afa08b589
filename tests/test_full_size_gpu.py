"""The BASELINE.json configurations at their FULL size (TPC-H SF10: 59 986 052 lineitem rows, 15 000 000 orders), compared
BYTE FOR BYTE with the oracle run on the same columns (PosLists and counts of configuration 2; both pair arrays and the
PosList cuts of configuration 3; the group rows and every aggregate value of configuration 4 as SURVEY.md 8(d) specifies it),
and additionally through properties that do not depend on the oracle: counts against numpy, order invariants of the
reference's output, key equality of EVERY join pair, aggregate totals within the stated float tolerance."""
import os

import ctypes as C

import numpy as np
import pytest

from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import aggregate_hash, make_predicate
from hyrise_amd.storage import DeviceColumn

from support import DeviceArray, oracle_aggregate, oracle_join, oracle_scan

ORACLE_THREADS = max(1, min(32, os.cpu_count() or 1))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sf10():
    return tpch.TpchData(scale_factor=10.0, seed=42)


def test_scan_sf10_shipdate(device, sf10):
    rows = sf10.n_lineitems
    host = storage.make_column(sf10.l_shipdate, None, abi.ENC_DICTIONARY)
    column = DeviceColumn(host)
    matches = DeviceArray(device, (rows, 2), np.uint32)
    offsets = DeviceArray(device, (host.n_chunks + 1,), np.int64)
    counts = DeviceArray(device, (host.n_chunks,), np.int32)
    result = abi.ScanResult()
    result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS
    result.matches, result.capacity = matches.pointer, rows
    result.offsets, result.counts = offsets.pointer, counts.pointer
    for condition, literal, literal2, expected in (
            (abi.PRED_LESS_THAN, tpch.DAY_1995_01_01, 0, sf10.l_shipdate < tpch.DAY_1995_01_01),
            (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, tpch.DAY_1994_01_01, tpch.DAY_1995_01_01, (sf10.l_shipdate >= tpch.DAY_1994_01_01) & (sf10.l_shipdate < tpch.DAY_1995_01_01))):
        predicate = make_predicate(condition, abi.TYPE_INT, literal, literal2)
        abi.check(device.hy_table_scan(column.handle, C.byref(predicate), None, 0, C.byref(result)))
        abi.check(device.hy_synchronize())
        got_counts = counts.numpy().astype(np.int64)
        chunk = abi.CHUNK_DEFAULT_SIZE
        per_chunk = np.add.reduceat(expected.astype(np.int64), np.arange(0, rows, chunk))
        np.testing.assert_array_equal(got_counts, per_chunk)                       # every chunk's count
        region = offsets.numpy()[:-1]
        np.testing.assert_array_equal(region, np.arange(host.n_chunks, dtype=np.int64) * chunk)
        # every RowID: concatenating the chunk regions gives exactly the matching row numbers, ascending
        host_matches = matches.numpy()
        keep = (np.arange(rows) % chunk) < np.repeat(got_counts, chunk)[:rows]      # the filled prefix of every region
        got_rows = host_matches[keep]
        np.testing.assert_array_equal(got_rows[:, 0].astype(np.int64) * chunk + got_rows[:, 1], np.flatnonzero(expected))
        # ... and the oracle's bytes: per-chunk counts, chunk states and every PosList, chunk by chunk
        # (the oracle packs the PosLists back to back; the device wrote each into its chunk's region, `keep` is the same bytes)
        want = oracle_scan(host, predicate, threads=ORACLE_THREADS)
        assert want.n_chunks == host.n_chunks and int(got_counts.sum()) == int(want.c.total_matches) == want.total   # (device-memory results leave total_matches to the caller)
        np.testing.assert_array_equal(got_counts, want.counts[:host.n_chunks].astype(np.int64))
        assert want.matches[:want.total].tobytes() == np.ascontiguousarray(got_rows).tobytes()


def test_scan_sf10_shipdate_string_twin(device, sf10):
    """l_shipdate as Hyrise holds it -- DictionarySegment<pmr_string> of ISO dates (tpch_table_generator.cpp:46) -- against its int
    twin: the host resolves the string literal per chunk (column_vs_value_table_scan_impl.cpp:211-226), the device compares value ids;
    PosLists, counts and chunk states are the oracle's bytes and the int twin's."""
    from hyrise_amd.operators import string_predicate, table_scan
    ints = storage.make_column(sf10.l_shipdate, None, abi.ENC_DICTIONARY)
    strings, dictionaries = tpch.string_date_column(ints)
    int_column, string_column = DeviceColumn(ints), DeviceColumn(strings)
    for condition, a, b, ia, ib in ((abi.PRED_LESS_THAN, "1995-01-01", None, tpch.DAY_1995_01_01, None),
                                    (abi.PRED_LESS_THAN_EQUALS, "1998-09-02", None, tpch.DAY_1998_09_02, None),
                                    (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, "1994-01-01", "1995-01-01", tpch.DAY_1994_01_01, tpch.DAY_1995_01_01),
                                    (abi.PRED_EQUALS, "1995-06-17", None, tpch.CURRENT_DATE, None), (abi.PRED_GREATER_THAN, "1999-01-01", None, 2600, None)):
        predicate = string_predicate(condition, dictionaries, a, b)
        got = table_scan(string_column, predicate)
        twin = table_scan(int_column, make_predicate(condition, abi.TYPE_INT, ia, ib))
        want = oracle_scan(strings, predicate, threads=ORACLE_THREADS)
        for other in (twin, want):
            assert got.total == other.total
            assert got.matches[:got.total].tobytes() == other.matches[:other.total].tobytes()
            np.testing.assert_array_equal(got.counts, other.counts)
            np.testing.assert_array_equal(got.chunk_state, other.chunk_state)


def test_join_sf10_orders_lineitem(device, sf10):
    orders = DeviceColumn(storage.make_column(sf10.o_orderkey, None, abi.ENC_UNENCODED))
    lineitem = DeviceColumn(storage.make_column(sf10.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE))
    n = sf10.n_lineitems
    left = DeviceArray(device, (n, 2), np.uint32)
    right = DeviceArray(device, (n, 2), np.uint32)
    slice_offsets = DeviceArray(device, (4096,), np.int64)
    r = abi.JoinResult()
    r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
    r.left_pos, r.right_pos, r.capacity = left.pointer, right.pointer, n
    r.slice_offsets, r.slice_capacity = slice_offsets.pointer, 4000
    abi.check(device.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(r)))
    abi.check(device.hy_synchronize())
    assert r.n_pairs == n and r.radix_bits == 7 and r.left_is_build == 1            # every lineitem has exactly one order
    chunk = abi.CHUNK_DEFAULT_SIZE
    build = left.numpy().astype(np.int64)
    probe = right.numpy().astype(np.int64)
    build_rows, probe_rows = build[:, 0] * chunk + build[:, 1], probe[:, 0] * chunk + probe[:, 1]
    np.testing.assert_array_equal(sf10.o_orderkey[build_rows], sf10.l_orderkey[probe_rows])        # EVERY pair joins equal keys
    partition = sf10.l_orderkey[probe_rows].astype(np.int64) & 127
    assert np.all(np.diff(partition) >= 0)                                          # pairs come radix partition by partition
    inside = np.diff(partition) == 0
    assert np.all(np.diff(probe_rows)[inside] > 0)                                  # ... ascending in the probe row inside one
    assert np.array_equal(np.sort(probe_rows), np.arange(n))                        # every probe row exactly once
    cuts = slice_offsets.numpy()[:r.n_slices + 1]
    assert cuts[0] == 0 and cuts[-1] == n and np.all(np.diff(cuts) > 0) and np.all(np.diff(cuts) <= 131070)
    per_partition = np.bincount(partition, minlength=128)
    assert r.n_slices == int(np.sum((per_partition + 131069) // 131070))           # a new PosList every 131 070 probe elements
    # ... and the oracle's bytes: both pair arrays and every PosList cut
    want = oracle_join(orders.host, lineitem.host, abi.JOIN_INNER, threads=ORACLE_THREADS, capacity=n)
    assert want.n_pairs == n and int(want.c.radix_bits) == 7 and int(want.c.left_is_build) == 1 and int(want.c.n_slices) == r.n_slices
    want_left, want_right = want.pairs()
    assert want_left.tobytes() == left.numpy().tobytes() and want_right.tobytes() == right.numpy().tobytes()
    np.testing.assert_array_equal(want.slice_offsets[:r.n_slices + 1].astype(np.int64), cuts)


def test_aggregate_sf10_q1_core(device, sf10):
    flag = DeviceColumn(storage.make_column(sf10.l_returnflag, None, abi.ENC_DICTIONARY))
    status = DeviceColumn(storage.make_column(sf10.l_linestatus, None, abi.ENC_DICTIONARY))
    quantity = DeviceColumn(storage.make_column(sf10.l_quantity, None, abi.ENC_UNENCODED))
    price = DeviceColumn(storage.make_column(sf10.l_extendedprice, None, abi.ENC_UNENCODED))
    got = aggregate_hash([flag, status], [(abi.AGG_SUM, quantity), (abi.AGG_AVG, price), (abi.AGG_COUNT, None), (abi.AGG_ANY, flag), (abi.AGG_ANY, status)],
                         group_capacity=64)
    keys = sf10.l_returnflag.astype(np.int64) * 256 + sf10.l_linestatus
    groups, first, counts = np.unique(keys, return_index=True, return_counts=True)
    order = np.argsort(first)                                                        # groups in order of first occurrence
    assert got.n_groups == len(groups)
    assert got.column(2) == counts[order].tolist()
    assert [f * 256 + s for f, s in zip(got.column(3), got.column(4))] == groups[order].tolist()
    for i, g in enumerate(groups[order]):
        members = keys == g
        total = float(sf10.l_quantity[members].astype(np.float64).sum())
        mean = float(sf10.l_extendedprice[members].astype(np.float64).mean())
        assert abs(got.column(0)[i] - total) <= 1e-9 * total and abs(got.column(1)[i] - mean) <= 1e-9 * mean


def test_aggregate_sf10_q1_as_specified(device, sf10):
    """Configuration 4 as SURVEY.md 8(d) specifies it: GROUP BY the two string columns (dictionary segments whose entries
    are AggregateKeyEntry names, aggregate_hash.cpp:852-914) over DictionarySegment<float> measures with u8 / u16 / u8
    attribute vectors -- against the oracle's bytes.  Byte-exact: the group rows (order of first occurrence), COUNT, ANY and
    the sums whose double accumulation is exact in any order (l_quantity: whole numbers; l_discount: hundredths as float32,
    49 significant bits at most).  SUM / AVG(l_extendedprice) need up to 60 bits, so they carry the stated 1e-9 relative
    tolerance (the device adds in another order than the reference's single thread)."""
    groupby, measures, _ = tpch.q1_core_columns(sf10)
    assert [s.width for s in (groupby[0].segments[0], groupby[1].segments[0])] == [1, 1]
    assert [measures[m].segments[0].width for m in ("l_quantity", "l_extendedprice", "l_discount")] == [1, 2, 1]
    device_groupby = [DeviceColumn(c) for c in groupby]
    device_measures = {name: DeviceColumn(c) for name, c in measures.items()}
    functions = [(abi.AGG_SUM, "l_quantity"), (abi.AGG_SUM, "l_extendedprice"), (abi.AGG_SUM, "l_discount"), (abi.AGG_AVG, "l_quantity"),
                 (abi.AGG_AVG, "l_extendedprice"), (abi.AGG_AVG, "l_discount"), (abi.AGG_COUNT, None)]
    got = aggregate_hash(device_groupby, [(f, device_measures[m] if m else None) for f, m in functions], group_capacity=64)
    want = oracle_aggregate(groupby, [(f, measures[m] if m else None) for f, m in functions], group_capacity=64)
    groups = want.n_groups
    assert got.n_groups == groups == 4
    assert got.row_ids[:groups].tobytes() == want.row_ids[:groups].tobytes()         # the first row of every group, in group order
    for a, (function, measure) in enumerate(functions):
        assert got.columns[a].data_type == want.columns[a].data_type
        assert got.nulls[a][:groups].tobytes() == want.nulls[a][:groups].tobytes()
        if measure == "l_extendedprice":
            np.testing.assert_allclose(got.raw[a][:groups].view(np.float64), want.raw[a][:groups].view(np.float64), rtol=1e-9, atol=0)
        else:
            assert got.raw[a][:groups].tobytes() == want.raw[a][:groups].tobytes(), (function, measure)
    # independent of the oracle: the four (returnflag, linestatus) groups and their sizes from numpy
    keys = sf10.l_returnflag.astype(np.int64) * 256 + sf10.l_linestatus
    values, first, counts = np.unique(keys, return_index=True, return_counts=True)
    order = np.argsort(first)
    assert got.column(6) == counts[order].tolist()
    chunk = abi.CHUNK_DEFAULT_SIZE
    assert (got.row_ids[:groups, 0].astype(np.int64) * chunk + got.row_ids[:groups, 1]).tolist() == first[order].tolist()


def test_join_large_unsorted_and_duplicate_builds(device):
    """The other build-side structures at a size where every kernel runs thousands of workgroups, against the oracle's bytes: unique build
    keys in random order (rank table filled with atomics + scattered rows), a shuffled probe side (random lookups, 4-byte FrameOfReference
    offsets), and every build key four times (sorted directory, multi-partner tiles)."""
    from hyrise_amd.operators import join_hash
    rng = np.random.default_rng(123)
    n_build, n_probe = 3_000_000, 12_000_000
    keys = tpch.sparse_orderkeys(n_build)
    probe_sorted = np.sort(keys[rng.integers(0, n_build, n_probe)])
    cases = {"shuffled build": (keys[rng.permutation(n_build)], probe_sorted),
             "shuffled probe": (keys, probe_sorted[rng.permutation(n_probe)]),
             "duplicate build x4": (np.repeat(keys[:n_build // 4], 4)[rng.permutation(n_build // 4 * 4)], probe_sorted[:n_probe // 3])}
    for name, (build, probe) in cases.items():
        build_host = storage.make_column(build, None, abi.ENC_UNENCODED)
        probe_host = storage.make_column(probe, None, abi.ENC_FRAME_OF_REFERENCE)
        got = join_hash(DeviceColumn(build_host), DeviceColumn(probe_host), abi.JOIN_INNER)
        want = oracle_join(build_host, probe_host, abi.JOIN_INNER, threads=ORACLE_THREADS, capacity=got.n_pairs + 16)
        n = want.n_pairs
        assert got.n_pairs == n > 0 and int(got.c.n_slices) == int(want.c.n_slices) and int(got.c.radix_bits) == int(want.c.radix_bits), name
        assert got.left[:n].tobytes() == want.left[:n].tobytes() and got.right[:n].tobytes() == want.right[:n].tobytes(), name
        np.testing.assert_array_equal(got.slice_offsets[:int(got.c.n_slices) + 1], want.slice_offsets[:int(want.c.n_slices) + 1], err_msg=name)


def test_aggregate_many_groups_at_size(device):
    """The hash-partitioned path at 20 M rows / 100 000 groups against the oracle: group rows and COUNT / integer SUM / MIN / MAX bytes,
    float sums within the stated tolerance."""
    rng = np.random.default_rng(321)
    n = 20_000_000
    keys = storage.make_column(rng.integers(0, 100_000, n).astype(np.int32) * 7 - 3, None, abi.ENC_UNENCODED)
    ints = storage.make_column(rng.integers(-1000, 1000, n).astype(np.int32), None, abi.ENC_FRAME_OF_REFERENCE)
    floats = storage.make_column((rng.random(n) * 100).astype(np.float32), None, abi.ENC_DICTIONARY)
    aggregates = [(abi.AGG_SUM, ints), (abi.AGG_MIN, ints), (abi.AGG_MAX, floats), (abi.AGG_SUM, floats), (abi.AGG_COUNT, None)]
    device_columns = {id(c): DeviceColumn(c) for c in (keys, ints, floats)}
    got = aggregate_hash([device_columns[id(keys)]], [(f, device_columns[id(c)] if c is not None else None) for f, c in aggregates], group_capacity=100_016)
    want = oracle_aggregate([keys], aggregates, group_capacity=100_016)
    groups = want.n_groups
    assert got.n_groups == groups == 100_000
    assert got.row_ids[:groups].tobytes() == want.row_ids[:groups].tobytes()
    for a in (0, 1, 2, 4):
        assert got.raw[a][:groups].tobytes() == want.raw[a][:groups].tobytes(), a
    np.testing.assert_allclose(got.raw[3][:groups].view(np.float64), want.raw[3][:groups].view(np.float64), rtol=1e-9, atol=0)
    lib = abi.load_library()
    lib.hy_debug_aggregate_path.restype = int
    assert lib.hy_debug_aggregate_path() > 0


def test_fused_q6_and_q1_sf10(device, sf10):
    """hy_scan_project_aggregate at full size: TPC-H Q6 against numpy (exact row count, revenue within the float tolerance) and Q1
    against an independent numpy evaluation per group (float32 expressions node by node, sums in double) -- and both against the
    operator chain run on the device."""
    import torch
    from hyrise_amd.distributed import HipExecutor
    ex = HipExecutor(torch.device("cuda", 0))
    q6 = {name: DeviceColumn(column) for name, column in tpch.q6_columns(sf10).items()}
    revenue, qualifying = tpch.q6_fused(q6)
    keep = (sf10.l_shipdate >= tpch.DAY_1994_01_01) & (sf10.l_shipdate < tpch.DAY_1995_01_01) & (sf10.l_discount >= np.float32(0.05)) & \
           (sf10.l_discount <= np.float32(0.07)) & (sf10.l_quantity < 24)
    exact = float((sf10.l_extendedprice[keep] * sf10.l_discount[keep]).astype(np.float64).sum())
    assert qualifying == int(keep.sum())
    assert abs(revenue - exact) <= 1e-9 * exact
    chain_revenue, chain_rows = tpch.run_q6(ex, q6)
    assert chain_rows == qualifying and abs(chain_revenue - revenue) <= 1e-9 * exact
    del q6

    q1 = {name: DeviceColumn(column) for name, column in tpch.q1_columns(sf10).items()}
    fused = tpch.q1_fused(q1)
    chain = tpch.run_q1(ex, q1)
    assert fused.n_groups == chain.n_groups == 4
    keep = sf10.l_shipdate <= tpch.DAY_1998_09_02
    one = np.float32(1)
    disc_price = (sf10.l_extendedprice * (one - sf10.l_discount)).astype(np.float32)
    charge = (disc_price * (one + sf10.l_tax)).astype(np.float32)
    chunk = abi.CHUNK_DEFAULT_SIZE
    for g in range(4):
        first = int(fused.row_ids[g][0]) * chunk + int(fused.row_ids[g][1])          # a row of the DATA table: the group's first row
        members = keep & (sf10.l_returnflag == sf10.l_returnflag[first]) & (sf10.l_linestatus == sf10.l_linestatus[first])
        assert members[first] and not members[:first].any()
        n = int(members.sum())
        want = [sf10.l_quantity[members].astype(np.float64).sum(), sf10.l_extendedprice[members].astype(np.float64).sum(), disc_price[members].astype(np.float64).sum(),
                charge[members].astype(np.float64).sum(), None, None, None, n]
        want[4], want[5], want[6] = want[0] / n, want[1] / n, sf10.l_discount[members].astype(np.float64).sum() / n
        for a, w in enumerate(want):
            got, through_chain = fused.column(a)[g], chain.column(a)[g]
            assert abs(got - w) <= 1e-9 * max(1.0, abs(w)), f"group {g} {tpch.Q1_AGGREGATES[a]}: {got} vs numpy {w}"
            assert abs(got - through_chain) <= 1e-9 * max(1.0, abs(w)), f"group {g} {tpch.Q1_AGGREGATES[a]}: {got} vs the chain's {through_chain}"
