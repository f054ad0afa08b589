"""Real dbgen rows ("real dbgen strings for parity with Hyrise"): the reference's vendored generator,
third_party/tpch-dbgen, compiled from the reference tree by `make -C oracle ref` and driven like TPCHTableGenerator::generate
(oracle/dbgen/tpch_rows.c).  The committed fixture tests/golden/dbgen/tpch_sf0.02.npz (tools/make_dbgen_fixture.py) travels to the GPU
box; where the generator binary itself is present (this container; it also travels, oracle/_ref is not gpurun-ignored) the fixture is
regenerated and compared, and scale factor 1 is checked against the published first rows of lineitem.tbl.

On these rows: l_shipdate as DictionarySegment<pmr_string> scans like its int twin (config 2's two forms), TPC-H Q6 and Q1 in the
reference's plans agree with SQLite / numpy (the tolerances of tests/test_tpch_q6.py), orders x lineitem joins like the oracle says."""
import os
import tempfile

import numpy as np
import pytest

from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import make_predicate, string_predicate
from oracle_executor import OracleExecutor
from support import GOLDEN, assert_scan_equal, oracle_scan
from test_tpch_q6 import check as check_q6

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(os.path.dirname(GOLDEN), "dbgen", "tpch_sf0.02.npz")
GENERATOR = os.path.join(ROOT, "oracle", "_ref", "tpch_rows")
needs_generator = pytest.mark.skipif(not os.path.exists(GENERATOR), reason="oracle/_ref/tpch_rows is built where /root/reference is (make -C oracle ref)")


@pytest.fixture(scope="module")
def data():
    return tpch.DbgenData.from_fixture(FIXTURE)


def test_fixture_shape(data):
    assert (data.n_orders, data.n_lineitems) == (30_000, 120_515)
    assert data.o_orderkey[:9].tolist() == [1, 2, 3, 4, 5, 6, 7, 32, 33]                      # mk_sparse: 8 of every 32 keys
    assert np.array_equal(data.o_orderkey, tpch.sparse_orderkeys(data.n_orders))
    assert np.all(np.diff(data.l_orderkey) >= 0) and set(np.unique(data.l_orderkey)) <= set(data.o_orderkey.tolist())
    assert data.l_quantity.min() == 1 and data.l_quantity.max() == 50
    assert sorted(set(np.round(data.l_discount * 100).astype(int))) == list(range(11)) and sorted(set(np.round(data.l_tax * 100).astype(int))) == list(range(9))
    assert set(map(chr, np.unique(data.l_returnflag))) == {"A", "N", "R"} and set(map(chr, np.unique(data.l_linestatus))) == {"F", "O"}
    # returnflag / linestatus follow the dates (TPC-H 4.2.3): shipped after 1995-06-17 = 'O'; received by then = 'R' or 'A'
    assert np.array_equal(data.l_linestatus == ord("O"), data.l_shipdate > tpch.CURRENT_DATE)
    assert np.array_equal(data.l_returnflag == ord("N"), data.l_receiptdate > tpch.CURRENT_DATE)


@needs_generator
def test_fixture_is_what_the_generator_writes(data):
    with tempfile.TemporaryDirectory() as tmp:
        fresh = tpch.DbgenData.generate(0.02, GENERATOR, tmp)
    for name in ("o_orderkey", "l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate", "l_commitdate", "l_receiptdate"):
        assert np.array_equal(getattr(fresh, name), getattr(data, name)), name


@needs_generator
def test_scale_factor_one_is_the_published_table():
    """The first rows of lineitem.tbl at scale factor 1 as every TPC-H kit prints them, and its row count."""
    with tempfile.TemporaryDirectory() as tmp:
        sf1 = tpch.DbgenData.generate(1, GENERATOR, tmp)
    assert (sf1.n_orders, sf1.n_lineitems) == (1_500_000, 6_001_215)
    rows = [(1, 17, 21168.23, 0.04, 0.02, "N", "O", "1996-03-13", "1996-02-12", "1996-03-22"),
            (1, 36, 45983.16, 0.09, 0.06, "N", "O", "1996-04-12", "1996-02-28", "1996-04-20"),
            (1, 8, 13309.60, 0.10, 0.02, "N", "O", "1996-01-29", "1996-03-05", "1996-01-31")]
    for i, (key, quantity, price, discount, tax, flag, status, ship, commit, receipt) in enumerate(rows):
        assert (sf1.l_orderkey[i], sf1.l_quantity[i], chr(sf1.l_returnflag[i]), chr(sf1.l_linestatus[i])) == (key, quantity, flag, status)
        assert sf1.l_extendedprice[i] == np.float32(price) and sf1.l_discount[i] == np.float32(discount) and sf1.l_tax[i] == np.float32(tax)
        assert (tpch.iso_date(sf1.l_shipdate[i]), tpch.iso_date(sf1.l_commitdate[i]), tpch.iso_date(sf1.l_receiptdate[i])) == (ship.encode(), commit.encode(), receipt.encode())


def shipdate_twins(data, chunk_size=abi.CHUNK_DEFAULT_SIZE):
    ints = storage.make_column(data.l_shipdate, None, abi.ENC_DICTIONARY, chunk_size)
    strings, dictionaries = tpch.string_date_column(ints)
    return ints, strings, dictionaries


SCANS = [(abi.PRED_LESS_THAN, "1995-01-01", None), (abi.PRED_LESS_THAN_EQUALS, "1998-09-02", None), (abi.PRED_BETWEEN_UPPER_EXCLUSIVE, "1994-01-01", "1995-01-01"),
         (abi.PRED_EQUALS, "1996-03-13", None), (abi.PRED_NOT_EQUALS, "1996-03-13", None), (abi.PRED_GREATER_THAN, "1998-12-01", None), (abi.PRED_GREATER_THAN_EQUALS, "1992-01-02", None)]


def day_of(text):
    return int((np.datetime64(text) - np.datetime64("1992-01-01")).astype(int))


def test_string_dates_scan_like_their_int_twin_on_the_oracle(data):
    ints, strings, dictionaries = shipdate_twins(data, chunk_size=20_000)
    for condition, low, high in SCANS:
        as_strings = oracle_scan(strings, string_predicate(condition, dictionaries, low, high))
        as_ints = oracle_scan(ints, make_predicate(condition, abi.TYPE_INT, day_of(low), day_of(high) if high else None))
        assert_scan_equal(as_strings, as_ints, f"dbgen l_shipdate, condition {condition} {low} {high}")
        days = data.l_shipdate
        expected = {abi.PRED_LESS_THAN: days < day_of(low), abi.PRED_LESS_THAN_EQUALS: days <= day_of(low), abi.PRED_EQUALS: days == day_of(low),
                    abi.PRED_NOT_EQUALS: days != day_of(low), abi.PRED_GREATER_THAN: days > day_of(low), abi.PRED_GREATER_THAN_EQUALS: days >= day_of(low),
                    abi.PRED_BETWEEN_UPPER_EXCLUSIVE: (days >= day_of(low)) & (days < day_of(high or low))}[condition]
        assert oracle_scan(ints, make_predicate(condition, abi.TYPE_INT, day_of(low), day_of(high) if high else None), flags=abi.SCAN_MATERIALIZE_ALL_MATCH).total == int(expected.sum())


def test_q6_on_dbgen_rows_matches_sqlite(data):
    revenue, qualifying = tpch.run_q6(OracleExecutor(), tpch.q6_columns(data, chunk_size=20_000))
    check_q6(data, revenue, qualifying)


def numpy_q1(data):
    keep = data.l_shipdate <= tpch.DAY_1998_09_02
    out = {}
    for flag in np.unique(data.l_returnflag):
        for status in np.unique(data.l_linestatus):
            rows = keep & (data.l_returnflag == flag) & (data.l_linestatus == status)
            if not rows.any():
                continue
            quantity, price, discount, tax = (getattr(data, name)[rows] for name in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"))
            disc_price = price * (np.float32(1) - discount)            # float arithmetic, like the reference's expression evaluator
            charge = disc_price * (np.float32(1) + tax)
            n = int(rows.sum())
            out[(int(flag), int(status))] = [float(quantity.astype(np.float64).sum()), float(price.astype(np.float64).sum()), float(disc_price.astype(np.float64).sum()),
                                             float(charge.astype(np.float64).sum()), float(quantity.astype(np.float64).sum()) / n, float(price.astype(np.float64).sum()) / n,
                                             float(discount.astype(np.float64).sum()) / n, n]
    return out


def check_q1(data, result):
    want = numpy_q1(data)
    assert result.n_groups == len(want) == 4
    columns = [result.column(a) for a in range(len(tpch.Q1_AGGREGATES))]
    counts = sorted(int(c) for c in columns[7])
    assert counts == sorted(v[7] for v in want.values())
    by_count = {v[7]: v for v in want.values()}
    for g in range(result.n_groups):
        expected = by_count[int(columns[7][g])]
        for a in range(7):
            assert abs(columns[a][g] - expected[a]) <= 1e-9 * abs(expected[a]), (tpch.Q1_AGGREGATES[a], columns[a][g], expected[a])


def test_q1_on_dbgen_rows_matches_numpy(data):
    result = tpch.run_q1(OracleExecutor(), tpch.q1_columns(data, chunk_size=20_000))
    check_q1(data, result)


def check_on_device(data, with_sqlite=True):
    """Config 2's two forms, config 3 and config 4 on real dbgen rows, through the C ABI: PosLists / pairs / groups are the oracle's."""
    import torch
    from hyrise_amd.distributed import HipExecutor
    from hyrise_amd.operators import aggregate_hash, join_hash, table_scan
    from hyrise_amd.storage import DeviceColumn
    from support import oracle_aggregate, oracle_join
    ints, strings, dictionaries = shipdate_twins(data)
    ints_dev, strings_dev = DeviceColumn(ints), DeviceColumn(strings)
    for condition, low, high in SCANS:
        string_pred = string_predicate(condition, dictionaries, low, high)
        int_pred = make_predicate(condition, abi.TYPE_INT, day_of(low), day_of(high) if high else None)
        want = oracle_scan(ints, int_pred)
        assert_scan_equal(table_scan(ints_dev, int_pred), want, f"device, int dates, condition {condition}")
        assert_scan_equal(table_scan(strings_dev, string_pred), want, f"device, string dates, condition {condition}")
    orders = storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)
    lineitem = storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
    for mode in (abi.JOIN_INNER, abi.JOIN_SEMI):
        got, want = join_hash(DeviceColumn(orders), DeviceColumn(lineitem), mode), oracle_join(orders, lineitem, mode)
        assert got.n_pairs == want.n_pairs and got.left[:got.n_pairs].tobytes() == want.left[:want.n_pairs].tobytes()
        if mode == abi.JOIN_INNER:
            assert got.n_pairs == data.n_lineitems and got.right[:got.n_pairs].tobytes() == want.right[:want.n_pairs].tobytes()
    groupby, measures, _ = tpch.q1_core_columns(data)
    spec = [(abi.AGG_SUM, "l_quantity"), (abi.AGG_SUM, "l_extendedprice"), (abi.AGG_AVG, "l_quantity"), (abi.AGG_AVG, "l_extendedprice"), (abi.AGG_AVG, "l_discount"), (abi.AGG_COUNT, None)]
    device_measures = {name: DeviceColumn(column) for name, column in measures.items()}
    got = aggregate_hash([DeviceColumn(c) for c in groupby], [(f, device_measures[c] if c else None) for f, c in spec])
    want = oracle_aggregate(groupby, [(f, measures[c] if c else None) for f, c in spec])
    assert got.n_groups == want.n_groups == 4 and np.array_equal(got.row_ids[:4], want.row_ids[:4])
    for a in range(len(spec)):
        for x, y in zip(got.column(a), want.column(a)):
            assert abs(x - y) <= 1e-9 * abs(y)
    ex = HipExecutor(torch.device("cuda", 0))
    revenue, qualifying = tpch.run_q6(ex, {name: DeviceColumn(column) for name, column in tpch.q6_columns(data).items()})
    if with_sqlite:
        check_q6(data, revenue, qualifying)
    else:   # (at scale: numpy and the oracle's plan; loading six million rows into SQLite takes longer than everything else here)
        from test_tpch_q6 import numpy_q6
        exact_revenue, exact_rows = numpy_q6(data)
        oracle_revenue, oracle_rows = tpch.run_q6(OracleExecutor(threads=os.cpu_count() or 1), tpch.q6_columns(data))
        assert qualifying == exact_rows == oracle_rows > 0 and abs(revenue - exact_revenue) <= 1e-9 * exact_revenue and abs(revenue - oracle_revenue) <= 1e-9 * exact_revenue
    check_q1(data, tpch.run_q1(ex, {name: DeviceColumn(column) for name, column in tpch.q1_columns(data).items()}))
    check_q1(data, tpch.q1_fused({name: DeviceColumn(column) for name, column in tpch.q1_columns(data).items()}))


@pytest.mark.gpu
def test_dbgen_rows_on_device(device, data):
    check_on_device(data)


@pytest.mark.gpu
@needs_generator
@pytest.mark.timeout(600)
def test_dbgen_scale_factor_one_on_device(device):
    """The same checks on the reference generator's scale factor 1 (1 500 000 orders, 6 001 215 lineitems, 92 chunks): string-date PosLists,
    orders x lineitem pair bytes (Inner and Semi), the Q1 core groups, Q6 and Q1 -- generated on the spot by oracle/_ref/tpch_rows (the
    reference's dbgen, compiled from the reference tree; skipped where the binary did not travel)."""
    with tempfile.TemporaryDirectory() as tmp:
        sf1 = tpch.DbgenData.generate(1, GENERATOR, tmp)
    assert (sf1.n_orders, sf1.n_lineitems) == (1_500_000, 6_001_215)
    check_on_device(sf1, with_sqlite=False)
