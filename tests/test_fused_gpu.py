"""hy_scan_project_aggregate (one pass: TableScan(s) -> Projection -> AggregateHash) against the operator chain on the CPU oracle
(support.oracle_chain, pinned by tests/test_oracle_chain.py) and against the same chain run operator by operator on the device.
Groups, group order and representative rows identical; integer results identical; SUM / AVG of floating-point inputs within 1e-9."""
import numpy as np
import pytest

import fused_cases
from hyrise_amd import abi, storage
from hyrise_amd.operators import make_predicate, scan_project_aggregate, string_predicate
from hyrise_amd.storage import DeviceColumn
from support import build_column, oracle_chain
from test_aggregate_gpu import FLOAT_TOLERANCE

pytestmark = pytest.mark.gpu


def assert_matches_chain(got, chain, n_aggregates, context):
    want, base_rows, sizes = chain
    assert got.n_groups == want.n_groups, f"group count {context}"
    n = want.n_groups
    if len(base_rows):   # the chain's representative rows are rows of its filtered table: the data table's rows behind them
        first_of_chunk = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        flat = first_of_chunk[want.row_ids[:n, 0].astype(np.int64)] + want.row_ids[:n, 1].astype(np.int64)
        np.testing.assert_array_equal(got.row_ids[:n], base_rows[flat], err_msg=f"group order / representative rows {context}")
    for a in range(n_aggregates if n else 0):   # (the oracle leaves the type of an empty result column unset)
        assert got.columns[a].data_type == want.columns[a].data_type, f"result type of aggregate {a} {context}"
        for x, y in zip(got.column(a), want.column(a)):
            if x is None or y is None:
                assert x is None and y is None, f"NULL mismatch aggregate {a} {context}"
            elif isinstance(y, float):
                assert abs(x - y) <= FLOAT_TOLERANCE * max(1.0, abs(y)), f"aggregate {a}: {x} vs {y} {context}"
            else:
                assert x == y, f"aggregate {a}: {x} vs {y} {context}"


@pytest.mark.parametrize("encoded", [True, False], ids=["encoded", "unencoded"])
@pytest.mark.parametrize("with_nulls", [False, True], ids=["not_null", "nullable"])
def test_fused_plans_match_the_oracle_chain(device, encoded, with_nulls):
    _, _, hosts = fused_cases.lineitem(n=60_000, chunk=10_000, encoded=encoded, with_nulls=with_nulls)
    devices = {name: DeviceColumn(column) for name, column in hosts.items()}
    for plan in fused_cases.plans(with_nulls):
        chain = oracle_chain(*plan.on(hosts))
        got = scan_project_aggregate(*plan.on(devices))
        assert_matches_chain(got, chain, len(plan.aggregates), f"plan {plan.name}")


def test_fused_matches_the_device_chain_on_q6(device):
    """... and the chain as the device runs it operator by operator (tpch.run_q6 on the HIP executor)."""
    from hyrise_amd import tpch
    from hyrise_amd.distributed import HipExecutor
    table, _, hosts = fused_cases.lineitem(n=300_000, chunk=65_535)
    import torch
    ex = HipExecutor(torch.device("cuda", 0))
    columns = {name: ex.column(hosts[name]) for name in ("l_shipdate", "l_discount", "l_quantity", "l_extendedprice")}
    revenue, qualifying = tpch.run_q6(ex, columns, date_from=731, date_to=1096)
    devices = {name: DeviceColumn(hosts[name]) for name in columns}
    plan = next(p for p in fused_cases.plans(False) if p.name == "q6")
    got = scan_project_aggregate(*plan.on(devices))
    assert got.column(1) == [qualifying]
    assert abs(got.column(0)[0] - revenue) <= FLOAT_TOLERANCE * abs(revenue)


def test_fused_filter_on_a_string_dictionary_column(device):
    """The reference's schema keeps l_shipdate as a string: the literal is resolved per chunk on the host (value ids), the kernel
    tests the attribute vectors -- the same jobs hy_table_scan gets."""
    from hyrise_amd import tpch
    table, _, hosts = fused_cases.lineitem(n=40_000, chunk=7_000)
    string_column, dictionaries = tpch.string_date_column(hosts["l_shipdate"])
    predicate = string_predicate(abi.PRED_LESS_THAN_EQUALS, dictionaries, tpch.iso_date(2436))
    int_predicate = make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, 2436)
    names = ("l_returnflag", "l_quantity", "l_extendedprice", "l_discount")
    devices = {name: DeviceColumn(hosts[name]) for name in names}
    aggregates = [(abi.AGG_SUM, fused_cases.bind(fused_cases.DISC_PRICE, devices)), (abi.AGG_COUNT, None)]
    on_strings = scan_project_aggregate([(DeviceColumn(string_column), predicate)], [devices["l_returnflag"]], aggregates)
    on_ints = scan_project_aggregate([(DeviceColumn(hosts["l_shipdate"]), int_predicate)], [devices["l_returnflag"]], aggregates)
    assert on_strings.n_groups == on_ints.n_groups == 3
    np.testing.assert_array_equal(on_strings.row_ids[:3], on_ints.row_ids[:3])
    assert on_strings.column(1) == on_ints.column(1)
    for x, y in zip(on_strings.column(0), on_ints.column(0)):
        assert abs(x - y) <= FLOAT_TOLERANCE * abs(y)


def test_fused_slices_with_more_groups_than_their_tables(device):
    """Every slice holds thousands of groups: 256 go through the workgroup's table, the rest to the global table row by row; the
    global table grows (64 Ki slots, then 2 Mi) when the groups do not fit."""
    rng = np.random.default_rng(11)
    n = 400_000
    keys = rng.integers(0, 150_000, n).astype(np.int32)
    other = rng.integers(0, 3, n).astype(np.int64)
    values = rng.integers(-1000, 1000, n).astype(np.int32)
    weights = rng.random(n).astype(np.float64)
    hosts = {"k": build_column(keys, None, 65_535, abi.ENC_DICTIONARY), "o": build_column(other, None, 65_535, abi.ENC_DICTIONARY),
             "v": build_column(values, None, 65_535, abi.ENC_FRAME_OF_REFERENCE), "w": build_column(weights, rng.random(n) < 0.1, 65_535, abi.ENC_UNENCODED)}
    devices = {name: DeviceColumn(column) for name, column in hosts.items()}
    plan = fused_cases.Plan("groups", [("v", make_predicate(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_INT, -900, 900))], ["k", "o"],
                            [(abi.AGG_SUM, (fused_cases.MUL, "v", (abi.TYPE_LONG, 3))), (abi.AGG_SUM, (fused_cases.MUL, "w", "v")), (abi.AGG_MIN, "w"), (abi.AGG_COUNT, "w"),
                             (abi.AGG_COUNT, None)])
    chain = oracle_chain(*plan.on(hosts))
    got = scan_project_aggregate(*plan.on(devices))
    assert got.n_groups > 200_000
    assert_matches_chain(got, chain, len(plan.aggregates), "many groups per slice")


def test_fused_rejects_what_the_chain_must_run(device):
    from hyrise_amd.abi import HyriseAmdError
    _, _, hosts = fused_cases.lineitem(n=5_000, chunk=1_000)
    d = {name: DeviceColumn(hosts[name]) for name in ("l_partkey", "l_quantity", "l_suppkey")}
    with pytest.raises(HyriseAmdError) as e:   # STDDEV_SAMP, COUNT(DISTINCT), ANY: the operator chain
        scan_project_aggregate([], [], [(abi.AGG_STDDEV_SAMP, d["l_quantity"])])
    assert e.value.status == abi.ERR_UNSUPPORTED
    deep = (fused_cases.ADD, d["l_partkey"], (fused_cases.ADD, d["l_partkey"], (fused_cases.ADD, d["l_partkey"], d["l_suppkey"])))   # four operands on the stack
    with pytest.raises(HyriseAmdError) as e:
        scan_project_aggregate([], [], [(abi.AGG_SUM, deep)])
    assert e.value.status == abi.ERR_UNSUPPORTED
    reference = DeviceColumn(storage.make_reference_column(hosts["l_partkey"], [0, 1, 2, 3, 4]), refs={id(hosts["l_partkey"]): d["l_partkey"]})
    with pytest.raises(HyriseAmdError) as e:
        scan_project_aggregate([(reference, make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 100))], [], [(abi.AGG_COUNT, None)])
    assert e.value.status == abi.ERR_UNSUPPORTED
    other_table = DeviceColumn(build_column(np.arange(10, dtype=np.int32), None, 5, abi.ENC_UNENCODED))
    with pytest.raises(HyriseAmdError) as e:
        scan_project_aggregate([], [d["l_partkey"]], [(abi.AGG_SUM, other_table)])
    assert e.value.status == abi.ERR_INVALID
    with pytest.raises(HyriseAmdError) as e:   # literal of another type than the column: hy_predicate_cast first, like hy_table_scan
        scan_project_aggregate([(d["l_quantity"], make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 24))], [], [(abi.AGG_COUNT, None)])
    assert e.value.status == abi.ERR_INVALID
