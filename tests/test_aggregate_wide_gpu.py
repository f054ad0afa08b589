"""AggregateHash over more than four GROUP BY columns (the reference takes any number: aggregate_hash.cpp:1184-1198 dispatches one to four
columns to fixed-size keys and everything else to AggregateKeySmallVector; key construction :661-948).  TPC-H Q10 groups by seven
columns, Q18 by five.  The device runs these plans on the nine-word build of its aggregate kernels (csrc/aggregate_wide.hip), nine to
sixteen columns on the seventeen-word build (csrc/aggregate_widest.hip); groups, their order and representative rows are compared with the
CPU oracle byte for byte, SUM / AVG of floats within 1e-9."""
import json
import os

import numpy as np
import pytest

from hyrise_amd import abi, storage
from hyrise_amd.operators import aggregate_hash, make_predicate, scan_project_aggregate
from hyrise_amd.storage import DeviceColumn
from support import GOLDEN, AggregateCase, build_column, oracle_aggregate, oracle_chain
from test_aggregate_gpu import CASES, aggregate_path, assert_aggregate_equal, finished_on_device, run_both
from test_fused_gpu import assert_matches_chain

pytestmark = pytest.mark.gpu


def customers(rng, n_customers):
    """One row per customer: the columns TPC-H Q10 groups by (strings as their key names: hyrise_amd/string_keys.py hands an int64 per
    distinct string)."""
    return {"c_custkey": np.arange(1, n_customers + 1, dtype=np.int32) * 3,
            "c_name": rng.permutation(n_customers).astype(np.int64) + (1 << 33),          # key names are arbitrary 64-bit words
            "c_acctbal": (rng.integers(-99999, 999999, n_customers) / 100.0).astype(np.float32),
            "c_phone": rng.integers(0, 1 << 40, n_customers).astype(np.int64),
            "n_name": rng.integers(0, 25, n_customers).astype(np.int32),
            "c_address": rng.integers(0, 1 << 20, n_customers).astype(np.int32),
            "c_comment": rng.integers(0, 50, n_customers).astype(np.int32)}


@pytest.mark.parametrize("n_customers,chunk", [(40, 20_000), (3_000, 65_535), (120_000, 65_535)], ids=["lds", "global", "partitioned"])
def test_q10_shaped_seven_columns(device, n_customers, chunk):
    """GROUP BY c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment over the join result's rows (a customer's rows carry the
    customer's values -- except that c_comment and c_address differ for a few rows, so groups that agree in the first five columns exist),
    SUM(l_extendedprice * (1 - l_discount)) as a float column, NULLs in two key columns."""
    rng = np.random.default_rng(n_customers)
    n = 400_000
    table = customers(rng, n_customers)
    of_row = rng.integers(0, n_customers, n)
    columns = {name: values[of_row].copy() for name, values in table.items()}
    odd = rng.random(n) < 0.03                      # rows whose sixth / seventh column differs from their customer's
    columns["c_address"][odd] += 1
    columns["c_comment"][rng.random(n) < 0.02] = 77
    nulls = {"c_acctbal": rng.random(n) < 0.01, "c_comment": rng.random(n) < 0.01}
    revenue = (rng.random(n) * 1e4).astype(np.float32)
    quantity = rng.integers(1, 51, n).astype(np.int32)
    groupby = [build_column(columns[name], nulls.get(name), chunk, abi.ENC_DICTIONARY if name in ("n_name", "c_comment", "c_acctbal") else abi.ENC_UNENCODED)
               for name in ("c_custkey", "c_name", "c_acctbal", "c_phone", "n_name", "c_address", "c_comment")]
    aggregates = [(abi.AGG_SUM, build_column(revenue, None, chunk, abi.ENC_UNENCODED)), (abi.AGG_COUNT, None),
                  (abi.AGG_MIN, build_column(quantity, rng.random(n) < 0.05, chunk, abi.ENC_FRAME_OF_REFERENCE)), (abi.AGG_AVG, build_column(quantity, None, chunk, abi.ENC_DICTIONARY))]
    got = run_both(groupby, aggregates, f"Q10 shape, {n_customers} customers")
    assert got.n_groups > n_customers
    if n_customers == 120_000:
        assert aggregate_path() != 0 and finished_on_device() == 1   # the hash-partitioned path, ordered and written by the finish kernels


def test_q18_shaped_five_columns(device):
    """GROUP BY c_name, c_custkey, o_orderkey, o_orderdate, o_totalprice with SUM(l_quantity): one group per order."""
    rng = np.random.default_rng(18)
    n_orders, n = 50_000, 200_000
    orders = {"c_name": rng.integers(0, 5000, n_orders).astype(np.int64) + (1 << 40), "c_custkey": rng.integers(0, 5000, n_orders).astype(np.int32),
              "o_orderkey": (np.arange(n_orders, dtype=np.int32) * 4 + 1), "o_orderdate": rng.integers(0, 2406, n_orders).astype(np.int32),
              "o_totalprice": rng.random(n_orders) * 5e5}
    of_row = np.sort(rng.integers(0, n_orders, n))
    groupby = [build_column(orders[name][of_row], None, 65_535, abi.ENC_DICTIONARY if name == "o_orderdate" else abi.ENC_UNENCODED)
               for name in ("c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice")]
    quantity = build_column(rng.integers(1, 51, n).astype(np.int32), None, 65_535, abi.ENC_DICTIONARY)
    got = run_both(groupby, [(abi.AGG_SUM, quantity)], "Q18 shape")
    assert got.n_groups == len(np.unique(of_row))


@pytest.mark.parametrize("n_columns", [5, 6, 8])
def test_groups_that_differ_in_the_last_column_only(device, n_columns):
    """Tuples equal in every column but the last; eight columns = the most the nine-word build takes."""
    rng = np.random.default_rng(n_columns)
    n = 100_000
    constant = [build_column(np.full(n, 5 + g, dtype=np.int32 if g % 2 else np.int64), None, 30_000, abi.ENC_DICTIONARY if g % 3 == 0 else abi.ENC_UNENCODED) for g in range(n_columns - 1)]
    last_values = rng.integers(0, 9, n).astype(np.int32)
    last = build_column(last_values, rng.random(n) < 0.1, 30_000, abi.ENC_UNENCODED)
    values = build_column(rng.integers(-5, 5, n).astype(np.int32), None, 30_000, abi.ENC_UNENCODED)
    got = run_both(constant + [last], [(abi.AGG_SUM, values), (abi.AGG_COUNT, None), (abi.AGG_ANY, last)], f"{n_columns} columns")
    assert got.n_groups == 10
    run_both(constant + [last], [], f"DISTINCT over {n_columns} columns")


def test_count_distinct_behind_four_groupby_columns(device):
    """COUNT(DISTINCT x) groups by (GROUP BY columns, x): with four GROUP BY columns that is a five-column key."""
    rng = np.random.default_rng(4)
    n = 120_000
    keys = [build_column(rng.integers(0, 3, n).astype(np.int32), rng.random(n) < 0.02 if g == 1 else None, 50_000, abi.ENC_DICTIONARY if g < 2 else abi.ENC_UNENCODED) for g in range(4)]
    x = build_column(rng.integers(0, 40, n).astype(np.int32), rng.random(n) < 0.1, 50_000, abi.ENC_UNENCODED)
    run_both(keys, [(abi.AGG_COUNT_DISTINCT, x), (abi.AGG_SUM, x), (abi.AGG_STDDEV_SAMP, x)], "COUNT(DISTINCT), four GROUP BY columns")


FIXTURES_WITH_GROUPS = [c for c in CASES if c["groupby"]]


@pytest.mark.parametrize("case", FIXTURES_WITH_GROUPS, ids=[f"L{c['line']}" for c in FIXTURES_WITH_GROUPS])
def test_reference_aggregate_fixture_with_padded_key_lists(device, case):
    """Every aggregate_test.cpp case with GROUP BY columns, its key list padded to six columns by repeating its own columns: the same
    groups in the same order with the same rows -- through the nine-word kernels; expected values: the oracle on the UNPADDED list."""
    columns = AggregateCase(case)
    padded = list(columns.groupby)
    while len(padded) < 6:
        padded.append(columns.groupby[len(padded) % len(columns.groupby)])
    cache = {}

    def dev(col):
        if id(col) not in cache:
            cache[id(col)] = DeviceColumn(col)
        return cache[id(col)]

    got = aggregate_hash([dev(c) for c in padded], [(f, dev(c) if c is not None else None) for f, c in columns.aggregates])
    want = oracle_aggregate(columns.groupby, columns.aggregates)
    # (a single int32 GROUP BY column takes the reference's immediate-key shortcut and comes out in KEY order, aggregate_hash.cpp:706-727;
    #  more than one column: first occurrence -- compare as the oracle orders the padded list)
    want_padded = oracle_aggregate(padded, columns.aggregates)
    assert_aggregate_equal(got, want_padded, len(columns.aggregates), f"aggregate_test.cpp:{case['line']} padded")
    assert got.n_groups == want.n_groups
    assert sorted(map(tuple, got.row_ids[:got.n_groups].tolist())) == sorted(map(tuple, want.row_ids[:want.n_groups].tolist()))


def test_fused_scan_project_aggregate_over_five_columns(device):
    """hy_scan_project_aggregate with a five-column GROUP BY against the operator chain on the CPU oracle."""
    rng = np.random.default_rng(55)
    n, chunk = 150_000, 40_000
    hosts = {f"k{g}": build_column(rng.integers(0, 3, n).astype(np.int32), rng.random(n) < 0.01 if g == 2 else None, chunk, abi.ENC_DICTIONARY if g % 2 else abi.ENC_UNENCODED) for g in range(5)}
    hosts["date"] = build_column(rng.integers(0, 2000, n).astype(np.int32), None, chunk, abi.ENC_DICTIONARY)
    hosts["price"] = build_column((rng.random(n) * 1000).astype(np.float32), None, chunk, abi.ENC_UNENCODED)
    hosts["discount"] = build_column((rng.integers(0, 11, n) / 100.0).astype(np.float32), None, chunk, abi.ENC_DICTIONARY)
    devices = {name: DeviceColumn(column) for name, column in hosts.items()}
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 1200)

    def plan(columns):
        return ([(columns["date"], predicate)], [columns[f"k{g}"] for g in range(5)],
                [(abi.AGG_SUM, (abi.ARITH_MUL, columns["price"], columns["discount"])), (abi.AGG_COUNT, None), (abi.AGG_MAX, columns["price"])])

    chain = oracle_chain(*plan(hosts))
    got = scan_project_aggregate(*plan(devices))
    assert_matches_chain(got, chain, 3, "fused, five GROUP BY columns")


@pytest.mark.parametrize("n_customers", [40, 3_000], ids=["lds", "global"])
def test_q10_shape_padded_to_twelve_columns(device, n_customers):
    """TPC-H Q10's seven GROUP BY columns and five more of the customer's (the reference takes any number, aggregate_hash.cpp:1184-1198):
    nine to sixteen columns run on the seventeen-word build of the kernels (csrc/aggregate_widest.hip) -- groups, order, representative
    rows and cells against the CPU oracle, like the narrower plans."""
    rng = np.random.default_rng(1200 + n_customers)
    n, chunk = 300_000, 65_535
    table = customers(rng, n_customers)
    for extra in range(5):
        table[f"c_extra{extra}"] = rng.integers(0, 4, n_customers).astype(np.int32 if extra % 2 else np.int64) - 1
    of_row = rng.integers(0, n_customers, n)
    columns = {name: values[of_row].copy() for name, values in table.items()}
    columns["c_extra4"][rng.random(n) < 0.02] = 9       # (groups that differ in the twelfth column only)
    nulls = {"c_acctbal": rng.random(n) < 0.01, "c_extra2": rng.random(n) < 0.01}
    names = ["c_custkey", "c_name", "c_acctbal", "c_phone", "n_name", "c_address", "c_comment"] + [f"c_extra{e}" for e in range(5)]
    groupby = [build_column(columns[name], nulls.get(name), chunk, abi.ENC_DICTIONARY if name in ("n_name", "c_comment", "c_acctbal", "c_extra1") else abi.ENC_UNENCODED) for name in names]
    revenue = (rng.random(n) * 1e4).astype(np.float32)
    quantity = rng.integers(1, 51, n).astype(np.int32)
    aggregates = [(abi.AGG_SUM, build_column(revenue, None, chunk, abi.ENC_UNENCODED)), (abi.AGG_COUNT, None),
                  (abi.AGG_MIN, build_column(quantity, rng.random(n) < 0.05, chunk, abi.ENC_FRAME_OF_REFERENCE)), (abi.AGG_AVG, build_column(quantity, None, chunk, abi.ENC_DICTIONARY))]
    got = run_both(groupby, aggregates, f"Q10 shape + five columns, {n_customers} customers")
    assert got.n_groups > n_customers
    run_both(groupby, [], "DISTINCT over twelve columns")


def test_sixteen_columns_and_count_distinct_behind_fifteen(device):
    """Sixteen GROUP BY columns = the most the device takes; COUNT(DISTINCT x) behind fifteen is a sixteen-column key."""
    rng = np.random.default_rng(16)
    n = 80_000
    keys = [build_column(rng.integers(0, 2, n).astype(np.int32 if g % 2 else np.int64), rng.random(n) < 0.01 if g == 11 else None, 30_000, abi.ENC_DICTIONARY if g % 3 == 0 else abi.ENC_UNENCODED)
            for g in range(16)]
    x = build_column(rng.integers(0, 20, n).astype(np.int32), rng.random(n) < 0.1, 30_000, abi.ENC_UNENCODED)
    run_both(keys, [(abi.AGG_SUM, x), (abi.AGG_COUNT, None), (abi.AGG_MAX, x)], "sixteen columns")
    run_both(keys[:15], [(abi.AGG_COUNT_DISTINCT, x), (abi.AGG_SUM, x)], "COUNT(DISTINCT), fifteen GROUP BY columns")


def test_fused_scan_project_aggregate_over_ten_columns(device):
    """hy_scan_project_aggregate with a ten-column GROUP BY against the operator chain on the CPU oracle."""
    rng = np.random.default_rng(1010)
    n, chunk = 120_000, 40_000
    hosts = {f"k{g}": build_column(rng.integers(0, 2, n).astype(np.int32), rng.random(n) < 0.01 if g == 7 else None, chunk, abi.ENC_DICTIONARY if g % 2 else abi.ENC_UNENCODED) for g in range(10)}
    hosts["date"] = build_column(rng.integers(0, 2000, n).astype(np.int32), None, chunk, abi.ENC_DICTIONARY)
    hosts["price"] = build_column((rng.random(n) * 1000).astype(np.float32), None, chunk, abi.ENC_UNENCODED)
    devices = {name: DeviceColumn(column) for name, column in hosts.items()}
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 1200)

    def plan(columns):
        return ([(columns["date"], predicate)], [columns[f"k{g}"] for g in range(10)], [(abi.AGG_SUM, columns["price"]), (abi.AGG_COUNT, None), (abi.AGG_MAX, columns["price"])])

    chain = oracle_chain(*plan(hosts))
    got = scan_project_aggregate(*plan(devices), group_capacity=4096)
    assert_matches_chain(got, chain, 3, "fused, ten GROUP BY columns")


def test_seventeen_columns_are_refused(device):
    column = DeviceColumn(build_column(np.arange(100, dtype=np.int32), None, 50, abi.ENC_UNENCODED))
    with pytest.raises(abi.HyriseAmdError) as error:
        aggregate_hash([column] * 17, [(abi.AGG_COUNT, None)])
    assert error.value.status == abi.ERR_UNSUPPORTED
