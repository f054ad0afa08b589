"""hy_scan_project_aggregate on the TPC-H Q1 shape -- the plans fused_small_domain takes (csrc/fused_small.hpp: a handful of groups over 1-byte
dictionary keys, filters on value ids, float expressions over DictionarySegment<float> columns) and the ones it must hand back to fused_rows
at run time (a fifth group in a chunk, a NULL in an input column) -- against the operator chain on the CPU oracle (support.oracle_chain).
Which kernel produced the groups is asserted (hy_debug_aggregate_small_domain: 2 = fused_small_domain), so that a plan that silently falls
back does not pass for the kernel.  Groups, group order and representative rows identical; COUNTs identical; SUM / AVG within 1e-9."""
import numpy as np
import pytest

from fused_cases import ADD, MUL, SUB, Plan
from hyrise_amd import abi
from hyrise_amd.operators import make_predicate, scan_project_aggregate
from hyrise_amd.storage import DeviceColumn
from support import build_column, oracle_chain
from test_fused_gpu import assert_matches_chain

pytestmark = pytest.mark.gpu

F32 = np.float32
DISC_PRICE = (MUL, "l_extendedprice", (SUB, (abi.TYPE_INT, 1), "l_discount"))
CHARGE = (MUL, DISC_PRICE, (ADD, (abi.TYPE_INT, 1), "l_tax"))


def table(n, chunk, flags=2, null_share=0.0, distinct_prices=10_000_000, seed=11):
    """lineitem's Q1 columns, every one a dictionary segment: `flags` x 2 groups; `distinct_prices` bounds the price dictionary of a chunk."""
    rng = np.random.default_rng(seed)
    values = {
        "l_shipdate": rng.integers(0, 2526, n).astype(np.int32),                     # 2-byte value ids
        "l_discount": (rng.integers(0, 11, n) / 100.0).astype(F32),
        "l_tax": (rng.integers(0, 9, n) / 100.0).astype(F32),
        "l_quantity": rng.integers(1, 51, n).astype(F32),
        "l_extendedprice": (rng.integers(90_000, 90_000 + distinct_prices, n) / 100.0).astype(F32),
        "l_returnflag": rng.integers(0, flags, n).astype(np.int64),
        "l_linestatus": rng.integers(0, 2, n).astype(np.int64),
    }
    nulls = {name: (rng.random(n) < null_share if null_share and name in ("l_discount", "l_extendedprice") else None) for name in values}
    return {name: build_column(v, nulls[name], chunk, abi.ENC_DICTIONARY, nullable=nulls[name] is not None) for name, v in values.items()}


def plans():
    p = make_predicate
    q1 = [(abi.AGG_SUM, "l_quantity"), (abi.AGG_SUM, "l_extendedprice"), (abi.AGG_SUM, DISC_PRICE), (abi.AGG_SUM, CHARGE), (abi.AGG_AVG, "l_quantity"),
          (abi.AGG_AVG, "l_extendedprice"), (abi.AGG_AVG, "l_discount"), (abi.AGG_COUNT, None)]
    return [
        Plan("q1", [("l_shipdate", p(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, 2436))], ["l_returnflag", "l_linestatus"], q1),
        Plan("q1_unfiltered", [], ["l_returnflag", "l_linestatus"], q1),
        # one key; a filter on 1-byte and one on 2-byte value ids; inputs over 1-byte columns only; COUNT of an expression; a float literal
        Plan("one_key_two_filters", [("l_quantity", p(abi.PRED_LESS_THAN, abi.TYPE_FLOAT, 24.0)), ("l_shipdate", p(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, 300, 2000))],
             ["l_linestatus"], [(abi.AGG_SUM, (MUL, "l_discount", "l_tax")), (abi.AGG_COUNT, "l_quantity"), (abi.AGG_AVG, (ADD, "l_tax", (abi.TYPE_FLOAT, F32(1.5)))), (abi.AGG_COUNT, None)]),
        # no GROUP BY: one group; Q6's expression behind two of its filters
        Plan("no_groups", [("l_shipdate", p(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, 731, 1096)), ("l_discount", p(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_FLOAT, F32(0.05), F32(0.07)))],
             [], [(abi.AGG_SUM, (MUL, "l_extendedprice", "l_discount")), (abi.AGG_COUNT, None)]),
        # a filter that empties some chunks' jobs and an inverted range
        Plan("not_equals", [("l_tax", p(abi.PRED_NOT_EQUALS, abi.TYPE_FLOAT, F32(0.04)))], ["l_returnflag"], [(abi.AGG_AVG, "l_extendedprice"), (abi.AGG_SUM, (SUB, "l_extendedprice", "l_quantity"))]),
        Plan("nothing_passes", [("l_shipdate", p(abi.PRED_GREATER_THAN, abi.TYPE_INT, 5000))], ["l_returnflag"], [(abi.AGG_SUM, "l_quantity"), (abi.AGG_COUNT, None)]),
    ]


def run(lib, hosts, plan):
    devices = {name: DeviceColumn(column) for name, column in hosts.items()}
    got = scan_project_aggregate(*plan.on(devices))
    return got, lib.hy_debug_aggregate_small_domain()


@pytest.mark.parametrize("layout", ["three_chunks", "full_chunks"])
def test_plans_of_the_q1_shape_run_on_their_kernel(device, layout):
    """four groups at most, no NULLs: every plan is answered by fused_small_domain -- small chunks (every price in the LDS window) and chunks of
    65535 rows whose price dictionaries (about 64 K entries) reach past the window (the gathers through the L2)"""
    lib = abi.load_library()
    hosts = table(25_000, 10_000) if layout == "three_chunks" else table(150_000, 65_535)
    for plan in plans():
        chain = oracle_chain(*plan.on(hosts))
        got, kernel = run(lib, hosts, plan)
        assert kernel == 2, f"plan {plan.name}: the groups did not come from fused_small_domain"
        assert_matches_chain(got, chain, len(plan.aggregates), f"plan {plan.name} ({layout})")


def test_what_the_kernel_refuses_falls_back(device):
    """a fifth group in a chunk (3 x 2 flags) and NULLs in the input columns: FLAG_SMALL_REFUSED, fused_rows answers -- same results"""
    lib = abi.load_library()
    for hosts, what in ((table(40_000, 10_000, flags=3), "six groups"), (table(40_000, 10_000, null_share=0.05), "NULL inputs")):
        for plan in plans()[:3]:
            chain = oracle_chain(*plan.on(hosts))
            got, kernel = run(lib, hosts, plan)
            if what == "six groups" and plan.name == "one_key_two_filters":
                assert kernel == 2   # (l_linestatus alone: two groups)
            elif what == "six groups" or plan.name != "one_key_two_filters":
                assert kernel == 0, f"plan {plan.name}, {what}: expected the fallback"
            assert_matches_chain(got, chain, len(plan.aggregates), f"plan {plan.name}, {what}")


def test_the_switch_that_turns_the_kernel_off(device, options):
    lib = abi.load_library()
    hosts = table(25_000, 10_000)
    plan = plans()[0]
    options.set(abi.OPT_FUSED_SMALL_DOMAIN, 0)
    got, kernel = run(lib, hosts, plan)
    assert kernel == 0
    assert_matches_chain(got, oracle_chain(*plan.on(hosts)), len(plan.aggregates), "fused_rows")
