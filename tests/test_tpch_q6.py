"""configs[0] of BASELINE.json: TPC-H Q6 -- "the reference's own CPU-runnable case ... verify results against SQLite"
(hyriseBenchmarkTPCH --verify).  tpch.run_q6 is the reference's plan (three chained TableScans, Projection, AggregateHash SUM);
it runs through the CPU oracle here (scale factor 0.01) and through libhyrise_amd.so at scale factor 1 on the GPU, and both
must agree with what SQLite computes from the query text over the same rows.

Tolerances.  The reference multiplies two float columns in float and sums the products in double (expression_functors.hpp,
aggregate_hash.cpp AggregateTraits<float, Sum>): against numpy doing exactly that the bar is the north star's 1e-9 relative.
SQLite multiplies in double, so against SQLite the bar is the float rounding of the products: 1e-6 relative."""
import sqlite3

import numpy as np
import pytest

from hyrise_amd import tpch
from oracle_executor import OracleExecutor


def sqlite_q6(data):
    db = sqlite3.connect(":memory:")
    db.execute("create table lineitem (l_shipdate integer, l_discount real, l_quantity real, l_extendedprice real)")
    db.executemany("insert into lineitem values (?, ?, ?, ?)",
                   zip(data.l_shipdate.tolist(), data.l_discount.astype(np.float64).tolist(), data.l_quantity.astype(np.float64).tolist(),
                       data.l_extendedprice.astype(np.float64).tolist()))
    # the float literals a Hyrise plan compares a float column with are floats (lossless_predicate_cast.cpp:40-73)
    parameters = {"from": tpch.DAY_1994_01_01, "to": tpch.DAY_1995_01_01, "low": float(np.float32(0.05)), "high": float(np.float32(0.07)), "quantity": 24.0}
    return db.execute(tpch.Q6_SQL, parameters).fetchone()


def numpy_q6(data):
    keep = (data.l_shipdate >= tpch.DAY_1994_01_01) & (data.l_shipdate < tpch.DAY_1995_01_01) & (data.l_discount >= np.float32(0.05)) & \
           (data.l_discount <= np.float32(0.07)) & (data.l_quantity < 24)
    return float((data.l_extendedprice[keep] * data.l_discount[keep]).astype(np.float64).sum()), int(keep.sum())


def check(data, revenue, qualifying):
    want_revenue, want_rows = sqlite_q6(data)
    exact_revenue, exact_rows = numpy_q6(data)
    assert qualifying == want_rows == exact_rows > 0
    assert abs(revenue - exact_revenue) <= 1e-9 * exact_revenue
    assert abs(revenue - want_revenue) <= 1e-6 * want_revenue


def test_q6_oracle_plan_matches_sqlite():
    data = tpch.TpchData(scale_factor=0.01, seed=11)
    revenue, qualifying = tpch.run_q6(OracleExecutor(), tpch.q6_columns(data, chunk_size=5000))
    check(data, revenue, qualifying)


def test_q6_no_qualifying_row_is_null():
    data = tpch.TpchData(scale_factor=0.002, seed=3)
    revenue, qualifying = tpch.run_q6(OracleExecutor(), tpch.q6_columns(data, chunk_size=5000), quantity=0.5)
    assert revenue is None and qualifying == 0


@pytest.mark.gpu
def test_q6_sf1_on_device_matches_sqlite(device):
    import torch
    from hyrise_amd.distributed import HipExecutor
    from hyrise_amd.storage import DeviceColumn
    data = tpch.TpchData(scale_factor=1.0, seed=42)
    host = tpch.q6_columns(data)
    columns = {name: DeviceColumn(column) for name, column in host.items()}
    revenue, qualifying = tpch.run_q6(HipExecutor(torch.device("cuda", 0)), columns)
    check(data, revenue, qualifying)
    oracle_revenue, oracle_rows = tpch.run_q6(OracleExecutor(), host)      # ... and the same plan on the CPU oracle
    assert oracle_rows == qualifying and abs(oracle_revenue - revenue) <= 1e-9 * abs(oracle_revenue)
