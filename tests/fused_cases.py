"""Plans for hy_scan_project_aggregate (TableScan(s) -> Projection -> AggregateHash of one data table), shared by the CPU test that
pins the oracle chain against numpy and the GPU parity test.  A plan names its columns; `bind` replaces the names by the
HostColumns (oracle) or DeviceColumns (device) of a table."""
import numpy as np

from hyrise_amd import abi
from hyrise_amd.operators import make_predicate
from support import build_column

ADD, SUB, MUL, DIV, MOD = abi.ARITH_ADD, abi.ARITH_SUB, abi.ARITH_MUL, abi.ARITH_DIV, abi.ARITH_MOD


def bind(tree, columns):
    if isinstance(tree, str):
        return columns[tree]
    if tree is None or len(tree) == 2:
        return tree
    return (tree[0], bind(tree[1], columns), bind(tree[2], columns))


class Plan:
    def __init__(self, name, filters, groupby, aggregates):
        self.name, self.filters, self.groupby, self.aggregates = name, filters, groupby, aggregates

    def on(self, columns):
        return ([(columns[c], p) for c, p in self.filters], [columns[g] for g in self.groupby],
                [(f, bind(tree, columns)) for f, tree in self.aggregates])


def lineitem(n=60_000, chunk=10_000, seed=7, encoded=True, with_nulls=False):
    """A small lineitem look-alike: the columns TPC-H Q1 / Q6 read, with the reference's types (float decimals, int dates)."""
    rng = np.random.default_rng(seed)
    table = {
        "l_shipdate": rng.integers(0, 2526, n).astype(np.int32),
        "l_discount": (rng.integers(0, 11, n) / 100.0).astype(np.float32),
        "l_tax": (rng.integers(0, 9, n) / 100.0).astype(np.float32),
        "l_quantity": rng.integers(1, 51, n).astype(np.float32),
        "l_extendedprice": (rng.integers(90_000, 10_500_000, n) / 100.0).astype(np.float32),
        "l_returnflag": rng.integers(0, 3, n).astype(np.int64),
        "l_linestatus": rng.integers(0, 2, n).astype(np.int64),
        "l_orderkey": np.sort(rng.integers(0, n // 4, n)).astype(np.int32),
        "l_partkey": rng.integers(0, 20_000, n).astype(np.int32),
        "l_suppkey": rng.integers(-50, 50, n).astype(np.int64),
    }
    nulls = {name: (rng.random(n) < 0.07 if with_nulls and name not in ("l_orderkey",) else None) for name in table}
    encodings = {}
    for name, values in table.items():
        if not encoded:
            encodings[name] = abi.ENC_UNENCODED
        elif values.dtype == np.int32 and name != "l_shipdate":
            encodings[name] = [abi.ENC_FRAME_OF_REFERENCE, abi.ENC_DICTIONARY, abi.ENC_UNENCODED]   # a mix of encodings inside one column, the last chunks unencoded
        else:
            encodings[name] = abi.ENC_DICTIONARY
    hosts = {name: build_column(values, nulls[name], chunk, encodings[name], nullable=with_nulls) for name, values in table.items()}
    return table, nulls, hosts


F32 = np.float32
REVENUE = (MUL, "l_extendedprice", "l_discount")
DISC_PRICE = (MUL, "l_extendedprice", (SUB, (abi.TYPE_INT, 1), "l_discount"))
CHARGE = (MUL, DISC_PRICE, (ADD, (abi.TYPE_INT, 1), "l_tax"))


def plans(nullable):
    p = lambda *a: make_predicate(*a, nullable=nullable)
    return [
        Plan("q6", [("l_shipdate", p(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, 731, 1096)),
                    ("l_discount", p(abi.PRED_BETWEEN_INCLUSIVE, abi.TYPE_FLOAT, F32(0.05), F32(0.07))),
                    ("l_quantity", p(abi.PRED_LESS_THAN, abi.TYPE_FLOAT, 24.0))],
             [], [(abi.AGG_SUM, REVENUE), (abi.AGG_COUNT, None)]),
        Plan("q1", [("l_shipdate", p(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, 2436))], ["l_returnflag", "l_linestatus"],
             [(abi.AGG_SUM, "l_quantity"), (abi.AGG_SUM, "l_extendedprice"), (abi.AGG_SUM, DISC_PRICE), (abi.AGG_SUM, CHARGE), (abi.AGG_AVG, "l_quantity"),
              (abi.AGG_AVG, "l_extendedprice"), (abi.AGG_AVG, "l_discount"), (abi.AGG_COUNT, None)]),
        Plan("no_filter_no_groups", [], [], [(abi.AGG_MIN, "l_extendedprice"), (abi.AGG_MAX, (SUB, "l_shipdate", "l_partkey")), (abi.AGG_COUNT, "l_tax"),
                                             (abi.AGG_AVG, "l_partkey"), (abi.AGG_SUM, (MUL, "l_suppkey", "l_partkey"))]),
        Plan("many_groups", [("l_quantity", p(abi.PRED_GREATER_THAN_EQUALS, abi.TYPE_FLOAT, 10.0))], ["l_partkey"],
             [(abi.AGG_SUM, "l_suppkey"), (abi.AGG_MIN, "l_shipdate"), (abi.AGG_MAX, (ADD, "l_quantity", (abi.TYPE_DOUBLE, 0.5))), (abi.AGG_AVG, "l_orderkey"),
              (abi.AGG_COUNT, "l_discount"), (abi.AGG_COUNT, None)]),
        Plan("three_keys", [("l_suppkey", p(abi.PRED_NOT_EQUALS, abi.TYPE_LONG, 7)), ("l_tax", p(abi.PRED_IS_NOT_NULL, abi.TYPE_FLOAT))],
             ["l_linestatus", "l_suppkey", "l_discount"], [(abi.AGG_SUM, (MOD, "l_partkey", (abi.TYPE_INT, 7))), (abi.AGG_MAX, "l_extendedprice")]),
        Plan("division", [("l_partkey", p(abi.PRED_LESS_THAN, abi.TYPE_INT, 5000))], ["l_returnflag"],
             [(abi.AGG_SUM, (DIV, "l_partkey", "l_suppkey")), (abi.AGG_COUNT, (DIV, "l_partkey", "l_suppkey")), (abi.AGG_AVG, (DIV, "l_extendedprice", (abi.TYPE_INT, 0))),
              (abi.AGG_MIN, (MOD, "l_suppkey", "l_linestatus")), (abi.AGG_SUM, (ADD, "l_quantity", None))]),
        Plan("nothing_passes", [("l_shipdate", p(abi.PRED_GREATER_THAN, abi.TYPE_INT, 5000))], [], [(abi.AGG_SUM, REVENUE), (abi.AGG_COUNT, None), (abi.AGG_MIN, "l_tax")]),
        Plan("nothing_passes_grouped", [("l_shipdate", p(abi.PRED_GREATER_THAN, abi.TYPE_INT, 5000))], ["l_returnflag"], [(abi.AGG_COUNT, None)]),
        Plan("null_test", [("l_discount", p(abi.PRED_IS_NULL, abi.TYPE_FLOAT))], ["l_linestatus"], [(abi.AGG_COUNT, None), (abi.AGG_COUNT, "l_discount"), (abi.AGG_SUM, "l_partkey")]),
        # the immediate-key shortcut (aggregate_hash.cpp:770-804) is decided on the row count of the aggregate's INPUT: a dense int key
        # behind a filter that keeps most rows (ascending keys, last rows) and behind one that keeps few (first-occurrence order)
        Plan("immediate_key", [("l_quantity", p(abi.PRED_GREATER_THAN_EQUALS, abi.TYPE_FLOAT, 2.0))], ["l_orderkey"], [(abi.AGG_COUNT, None), (abi.AGG_SUM, "l_partkey")]),
        Plan("no_immediate_key", [("l_quantity", p(abi.PRED_EQUALS, abi.TYPE_FLOAT, 2.0))], ["l_orderkey"], [(abi.AGG_COUNT, None), (abi.AGG_SUM, "l_partkey")]),
    ]
