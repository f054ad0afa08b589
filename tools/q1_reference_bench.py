#!/usr/bin/env python3
"""Ad-hoc timing of the TPC-H Q1 core the way a Hyrise plan runs it: TableScan l_shipdate <= 1998-09-02 (98.6 % of the rows), then
AggregateHash over the REFERENCE table the scan produced (every column read through the scan's PosLists) -- beside the same aggregate
over the data table (debug aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyrise_amd import abi, storage, tpch
from hyrise_amd.distributed import HipExecutor
from hyrise_amd.operators import make_predicate
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
data = tpch.TpchData(scale_factor=float(os.environ.get("SF", "10")), seed=42)
groupby_host, measures_host, _ = tpch.q1_core_columns(data)
shipdate = DeviceColumn(storage.make_column(data.l_shipdate, None, abi.ENC_DICTIONARY))
groupby = [DeviceColumn(c) for c in groupby_host]
measures = {name: DeviceColumn(c) for name, c in measures_host.items()}
ex = HipExecutor(torch.device("cuda", 0))


def spec(columns):
    return [(abi.AGG_SUM, columns["l_quantity"]), (abi.AGG_SUM, columns["l_extendedprice"]), (abi.AGG_AVG, columns["l_quantity"]),
            (abi.AGG_AVG, columns["l_extendedprice"]), (abi.AGG_AVG, columns["l_discount"]), (abi.AGG_COUNT, None)]


def timed(name, run, repeats=5):
    run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(repeats):
        out = run()
    torch.cuda.synchronize()
    print(f"{name:44s} {(time.perf_counter() - t) / repeats * 1e3:8.3f} ms")
    return out


direct = timed("aggregate over the data table", lambda: ex.aggregate(groupby, spec(measures)))
lists = timed("scan l_shipdate <= 1998-09-02 (device PosLists)", lambda: ex.scan_chunked(shipdate, make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, tpch.DAY_1998_09_02)))
refs = timed("reference columns over the PosLists (5)", lambda: ([ex.reference_column_chunked(c, lists) for c in groupby], {n: ex.reference_column_chunked(c, lists) for n, c in measures.items()}))
through = timed("aggregate over the reference table", lambda: ex.aggregate(refs[0], spec(refs[1])))
print("rows", lists.total, "groups", direct.n_groups, through.n_groups, "counts", direct.column(5), through.column(5))
