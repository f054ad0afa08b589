"""The checker of the fused kernel: TableScan(s) -> Projection -> AggregateHash run operator by operator on the CPU oracle
(support.oracle_chain), pinned here against an independent numpy evaluation of the same plans.  CPU only."""
import numpy as np
import pytest

import fused_cases
from hyrise_amd import abi
from support import oracle_chain


def numpy_plan(table, nulls, plan):
    """Independent evaluation: masks for the filters, float32 / int arithmetic by numpy, groups by python dicts in first-occurrence order."""
    n = len(next(iter(table.values())))
    keep = np.ones(n, dtype=bool)
    for name, predicate in plan.filters:
        from support import brute_force_scan
        value = {abi.TYPE_INT: predicate.value.i32, abi.TYPE_LONG: predicate.value.i64, abi.TYPE_FLOAT: predicate.value.f32, abi.TYPE_DOUBLE: predicate.value.f64}[predicate.value_type]
        value2 = {abi.TYPE_INT: predicate.value2.i32, abi.TYPE_LONG: predicate.value2.i64, abi.TYPE_FLOAT: predicate.value2.f32, abi.TYPE_DOUBLE: predicate.value2.f64}[predicate.value_type]
        keep &= brute_force_scan(table[name], nulls[name], predicate.condition, value, value2)
    rows = np.nonzero(keep)[0]
    return rows


@pytest.mark.parametrize("with_nulls", [False, True], ids=["not_null", "nullable"])
def test_chain_filters_and_row_ids(with_nulls):
    table, nulls, hosts = fused_cases.lineitem(n=20_000, chunk=3_000, with_nulls=with_nulls)
    for plan in fused_cases.plans(with_nulls):
        filters, groupby, aggregates = plan.on(hosts)
        result, base_rows, sizes = oracle_chain(filters, groupby, aggregates)
        rows = numpy_plan(table, nulls, plan)
        np.testing.assert_array_equal(base_rows[:, 0].astype(np.int64) * 3_000 + base_rows[:, 1], rows, err_msg=plan.name)
        assert sum(sizes) == len(rows) or (len(rows) == 0 and sizes == [0])


def test_chain_q6_and_q1_values():
    table, nulls, hosts = fused_cases.lineitem(n=20_000, chunk=3_000)
    by_name = {p.name: p for p in fused_cases.plans(False)}
    # Q6: float32 products summed in double
    result, _, _ = oracle_chain(*by_name["q6"].on(hosts))
    rows = numpy_plan(table, nulls, by_name["q6"])
    revenue = (table["l_extendedprice"][rows] * table["l_discount"][rows]).astype(np.float32).astype(np.float64).sum()
    assert result.n_groups == 1 and result.column(1) == [len(rows)]
    assert abs(result.column(0)[0] - revenue) <= 1e-9 * abs(revenue)
    # Q1: groups in first-occurrence order, float32 expressions node by node (1 - l_discount in float, the product in float)
    result, _, _ = oracle_chain(*by_name["q1"].on(hosts))
    rows = numpy_plan(table, nulls, by_name["q1"])
    groups = {}
    for r in rows:
        groups.setdefault((int(table["l_returnflag"][r]), int(table["l_linestatus"][r])), []).append(r)
    assert result.n_groups == len(groups)
    one = np.float32(1)
    for g, (key, members) in enumerate(groups.items()):
        members = np.array(members)
        price, discount, tax = table["l_extendedprice"][members], table["l_discount"][members], table["l_tax"][members]
        disc_price = (price * (one - discount).astype(np.float32)).astype(np.float32)
        charge = (disc_price * (one + tax).astype(np.float32)).astype(np.float32)
        want = [table["l_quantity"][members].astype(np.float64).sum(), price.astype(np.float64).sum(), disc_price.astype(np.float64).sum(), charge.astype(np.float64).sum(),
                table["l_quantity"][members].astype(np.float64).mean(), price.astype(np.float64).mean(), discount.astype(np.float64).mean(), len(members)]
        for a, w in enumerate(want):
            got = result.column(a)[g]
            assert abs(got - w) <= 1e-9 * max(1.0, abs(w)), f"group {key} aggregate {a}: {got} vs {w}"
