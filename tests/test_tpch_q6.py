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


def mvcc_for(data, rng, chunk_size):
    """MvccData of a lineitem table that has seen some transactions: 3 % of the rows deleted before the snapshot, 2 % after it, 1 %
    inserted after it, a few in-flight rows of our own and of another transaction.  -> (mvcc column, visible mask for tid 7 / snapshot 10)"""
    from hyrise_amd import storage
    n = data.n_lineitems
    tids = np.zeros(n, dtype=np.uint32)
    begin = np.full(n, 3, dtype=np.uint32)
    end = np.full(n, storage.MAX_COMMIT_ID, dtype=np.uint32)
    pick = rng.random(n)
    end[pick < 0.03] = 8                                   # deleted before the snapshot: invisible
    end[(pick >= 0.03) & (pick < 0.05)] = 12               # deleted after it: still visible
    begin[(pick >= 0.05) & (pick < 0.06)] = 11             # inserted after it: invisible
    ours = (pick >= 0.06) & (pick < 0.062)                 # our own uncommitted inserts: visible to us
    begin[ours], tids[ours] = storage.MAX_COMMIT_ID, 7
    theirs = (pick >= 0.062) & (pick < 0.064)              # somebody else's: invisible
    begin[theirs], tids[theirs] = storage.MAX_COMMIT_ID, 9
    snapshot, our_tid = 10, 7
    visible = (snapshot < end) & ((snapshot >= begin) != (tids == our_tid))
    return storage.make_mvcc_column(tids, begin, end, chunk_size), visible


def storage_mvcc_all_visible(n, chunk_size=65_535):
    """MvccData of a table nobody has touched since it was loaded: every chunk immutable and entirely visible (validate.cpp:57-68)."""
    from hyrise_amd import storage
    return storage.make_mvcc_column(np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.full(n, storage.MAX_COMMIT_ID, np.uint32), chunk_size)


def numpy_q6_visible(data, visible):
    keep = visible & (data.l_shipdate >= tpch.DAY_1994_01_01) & (data.l_shipdate < tpch.DAY_1995_01_01) & (data.l_discount >= np.float32(0.05)) & \
           (data.l_discount <= np.float32(0.07)) & (data.l_quantity < 24)
    return float((data.l_extendedprice[keep] * data.l_discount[keep]).astype(np.float64).sum()), int(keep.sum())


def test_q6_behind_validate_on_the_oracle():
    """GetTable -> Validate -> TableScan x3 -> Projection -> AggregateHash, the plan shape of hyriseBenchmarkTPCH."""
    data = tpch.TpchData(scale_factor=0.01, seed=12)
    mvcc, visible = mvcc_for(data, np.random.default_rng(1), 5000)
    revenue, qualifying = tpch.run_q6(OracleExecutor(), tpch.q6_columns(data, chunk_size=5000), mvcc=mvcc, transaction=(7, 10))
    exact_revenue, exact_rows = numpy_q6_visible(data, visible)
    assert qualifying == exact_rows > 0 and qualifying < numpy_q6(data)[1]
    assert abs(revenue - exact_revenue) <= 1e-9 * exact_revenue


@pytest.mark.gpu
def test_q6_behind_validate_on_device(device):
    import torch
    from hyrise_amd.distributed import HipExecutor
    from hyrise_amd.storage import DeviceColumn
    data = tpch.TpchData(scale_factor=0.5, seed=42)
    mvcc, visible = mvcc_for(data, np.random.default_rng(2), 65_535)
    host = tpch.q6_columns(data)
    columns = {name: DeviceColumn(column) for name, column in host.items()}
    mvcc_device = DeviceColumn(mvcc)
    revenue, qualifying = tpch.run_q6(HipExecutor(torch.device("cuda", 0)), columns, mvcc=mvcc_device, transaction=(7, 10))
    exact_revenue, exact_rows = numpy_q6_visible(data, visible)
    assert qualifying == exact_rows > 0
    assert abs(revenue - exact_revenue) <= 1e-9 * exact_revenue
    oracle_revenue, oracle_rows = tpch.run_q6(OracleExecutor(), host, mvcc=mvcc, transaction=(7, 10))
    assert oracle_rows == qualifying and abs(oracle_revenue - revenue) <= 1e-9 * abs(oracle_revenue)
    # the same plan in ONE pass: Validate is the first filter of hy_scan_project_aggregate (HY_FILTER_VALIDATE)
    fused_revenue, fused_rows = tpch.q6_fused(columns, mvcc=mvcc_device, transaction=(7, 10))
    assert fused_rows == qualifying and abs(fused_revenue - revenue) <= 1e-9 * abs(revenue)
    # ... for another transaction (nobody's uncommitted rows are ours), and with the entirely-visible-chunk shortcut taken by every chunk
    for our_tid, snapshot in ((3, 10), (7, 2)):
        chain = tpch.run_q6(HipExecutor(torch.device("cuda", 0)), columns, mvcc=mvcc_device, transaction=(our_tid, snapshot))
        fused = tpch.q6_fused(columns, mvcc=mvcc_device, transaction=(our_tid, snapshot))
        assert fused[1] == chain[1] and (chain[0] is None) == (fused[0] is None)
        if chain[0] is not None:
            assert abs(fused[0] - chain[0]) <= 1e-9 * abs(chain[0])
    n = data.n_lineitems
    clean = DeviceColumn(storage_mvcc_all_visible(n))
    assert tpch.q6_fused(columns, mvcc=clean, transaction=(7, 10))[1] == numpy_q6(data)[1]
    from hyrise_amd import abi
    short = DeviceColumn(storage_mvcc_all_visible(1000))   # the MvccData of some other table
    with pytest.raises(abi.HyriseAmdError):
        tpch.q6_fused(columns, mvcc=short, transaction=(7, 10))


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
