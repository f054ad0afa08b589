"""TPC-H Q1, the whole query (tpch_queries.cpp:60-80), as the reference plans it -- TableScan, a Projection of two expressions over the
survivors, AggregateHash of eight aggregates GROUP BY l_returnflag, l_linestatus (hyrise_amd/tpch.py run_q1) -- on the CPU oracle at
scale factor 0.01: against SQLite over the same rows (hyriseBenchmarkTPCH --verify's practice), against numpy doing the reference's
arithmetic (float expressions node by node, sums in double), and against the plan of the fused pass run the reference's way.  The GPU
runs both at scale factor 10 (tests/test_full_size_gpu.py, bench.py `q1`)."""
import sqlite3

import numpy as np

from hyrise_amd import abi, tpch
from oracle_executor import OracleExecutor

Q1_SQL = ("SELECT l_returnflag, l_linestatus, SUM(l_quantity), SUM(l_extendedprice), SUM(l_extendedprice * (1 - l_discount)), "
          "SUM(l_extendedprice * (1 - l_discount) * (1 + l_tax)), AVG(l_quantity), AVG(l_extendedprice), AVG(l_discount), COUNT(*) "
          "FROM lineitem WHERE l_shipdate <= :to GROUP BY l_returnflag, l_linestatus")


def sqlite_q1(data):
    db = sqlite3.connect(":memory:")
    db.execute("create table lineitem (l_shipdate integer, l_returnflag integer, l_linestatus integer, l_quantity real, l_extendedprice real, l_discount real, l_tax real)")
    as_real = lambda a: a.astype(np.float64).tolist()
    db.executemany("insert into lineitem values (?, ?, ?, ?, ?, ?, ?)",
                   zip(data.l_shipdate.tolist(), data.l_returnflag.tolist(), data.l_linestatus.tolist(), as_real(data.l_quantity), as_real(data.l_extendedprice),
                       as_real(data.l_discount), as_real(data.l_tax)))
    return {(int(r[0]), int(r[1])): r[2:] for r in db.execute(Q1_SQL, {"to": tpch.DAY_1998_09_02})}


def numpy_q1(data):
    keep = data.l_shipdate <= tpch.DAY_1998_09_02
    one = np.float32(1)
    disc_price = (data.l_extendedprice * (one - data.l_discount)).astype(np.float32)
    charge = (disc_price * (one + data.l_tax)).astype(np.float32)
    out = {}
    for r in np.nonzero(keep)[0]:   # first-occurrence order
        out.setdefault((int(data.l_returnflag[r]), int(data.l_linestatus[r])), []).append(r)
    result = []
    for key, members in out.items():
        m = np.array(members)
        total = lambda a: float(a[m].astype(np.float64).sum())
        n = len(m)
        result.append((key, m[0], [total(data.l_quantity), total(data.l_extendedprice), total(disc_price), total(charge), total(data.l_quantity) / n,
                                   total(data.l_extendedprice) / n, total(data.l_discount) / n, n]))
    return result


def test_q1_oracle_plan_matches_numpy_and_sqlite():
    data = tpch.TpchData(scale_factor=0.01, seed=21)
    chunk = 5000
    columns = tpch.q1_columns(data, chunk_size=chunk)
    ex = OracleExecutor()
    chain = tpch.run_q1(ex, columns)
    want = numpy_q1(data)
    by_sql = sqlite_q1(data)
    assert chain.n_groups == len(want) == len(by_sql) == 4
    for g, (key, first_row, cells) in enumerate(want):
        for a, expected in enumerate(cells):
            got = chain.column(a)[g]
            assert abs(got - expected) <= 1e-9 * max(1.0, abs(expected)), f"group {key} {tpch.Q1_AGGREGATES[a]}: {got} vs numpy {expected}"
            assert abs(got - by_sql[key][a]) <= 1e-6 * max(1.0, abs(by_sql[key][a])), f"group {key} {tpch.Q1_AGGREGATES[a]}: {got} vs SQLite {by_sql[key][a]}"
    # the plan of the fused pass (what hy_scan_project_aggregate is given), run the reference's way: the same groups in the same order,
    # their representative rows named in the DATA table
    from hyrise_amd.operators import make_predicate
    one = (abi.TYPE_INT, 1)
    disc_price = (abi.ARITH_MUL, columns["l_extendedprice"], (abi.ARITH_SUB, one, columns["l_discount"]))
    charge = (abi.ARITH_MUL, disc_price, (abi.ARITH_ADD, one, columns["l_tax"]))
    fused = ex.scan_project_aggregate([(columns["l_shipdate"], make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, tpch.DAY_1998_09_02))],
                                      [columns["l_returnflag"], columns["l_linestatus"]],
                                      tpch._q1_aggregates(columns["l_quantity"], columns["l_extendedprice"], columns["l_discount"], disc_price, charge))
    assert fused.n_groups == 4
    for g, (key, first_row, cells) in enumerate(want):
        assert int(fused.row_ids[g][0]) * chunk + int(fused.row_ids[g][1]) == first_row
        for a in range(8):
            assert abs(fused.column(a)[g] - chain.column(a)[g]) <= 1e-12 * max(1.0, abs(chain.column(a)[g]))
