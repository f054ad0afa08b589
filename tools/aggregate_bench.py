#!/usr/bin/env python3
"""Ad-hoc timing of the TPC-H Q1 core on one GPU (config 4 of BASELINE.json): GROUP BY l_returnflag, l_linestatus with
SUM / AVG / COUNT(*) over SF10 lineitem (debug aid; bench.py reports the headline)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import aggregate_hash
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
sf = float(os.environ.get("SF", "10"))
data = tpch.TpchData(scale_factor=sf, seed=42)
n = data.n_lineitems
if os.environ.get("PLAIN"):   # round 1's shape: int32 dictionary keys, unencoded float measures
    cols = {
        "l_returnflag": storage.make_column(data.l_returnflag, None, abi.ENC_DICTIONARY),
        "l_linestatus": storage.make_column(data.l_linestatus, None, abi.ENC_DICTIONARY),
        "l_quantity": storage.make_column(data.l_quantity, None, abi.ENC_UNENCODED),
        "l_extendedprice": storage.make_column(data.l_extendedprice, None, abi.ENC_UNENCODED),
        "l_discount": storage.make_column(data.l_discount, None, abi.ENC_UNENCODED),
    }
    bytes_total = n * (1 + 1 + 4 + 4 + 4)
else:                         # config 4 as specified: string keys as key names, dictionary-encoded float measures
    groupby_host, measures_host, bytes_total = tpch.q1_core_columns(data)
    cols = dict(measures_host, l_returnflag=groupby_host[0], l_linestatus=groupby_host[1])
dev = {k: DeviceColumn(v) for k, v in cols.items()}
aggregates = [(abi.AGG_SUM, dev["l_quantity"]), (abi.AGG_SUM, dev["l_extendedprice"]), (abi.AGG_AVG, dev["l_quantity"]),
              (abi.AGG_AVG, dev["l_extendedprice"]), (abi.AGG_AVG, dev["l_discount"]), (abi.AGG_COUNT, None)]
if os.environ.get("EXTREMES"):   # the kernel's round-5 widening: MIN / MAX (the smallest / largest value id a group counted) next to the sums
    aggregates = [(abi.AGG_SUM, dev["l_quantity"]), (abi.AGG_MIN, dev["l_extendedprice"]), (abi.AGG_MAX, dev["l_extendedprice"]), (abi.AGG_AVG, dev["l_extendedprice"]),
                  (abi.AGG_MIN, dev["l_discount"]), (abi.AGG_MAX, dev["l_quantity"]), (abi.AGG_COUNT, None)]
n_aggs = int(os.environ.get("AGGS", str(len(aggregates))))
aggregates = aggregates[:n_aggs] if n_aggs else [(abi.AGG_COUNT, None)]
for i in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    result = aggregate_hash([dev["l_returnflag"], dev["l_linestatus"]], aggregates, group_capacity=64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    lib.hy_debug_aggregate_small_domain.restype = int
    print(f"aggregate ms {dt * 1e3:.3f} groups {result.n_groups} rows/s {n / dt:.3g} algorithmic GB/s {bytes_total / dt / 1e9:.0f} small-domain kernel {lib.hy_debug_aggregate_small_domain()}")
if os.environ.get("HY_AGG_TRACE"):
    lib.hy_debug_aggregate_trace.argtypes = [C.c_void_p, C.c_uint32]
    lib.hy_debug_aggregate_trace.restype = C.c_int
    buf = np.zeros((1 << 14, 12), dtype=np.uint64)
    ns = lib.hy_debug_aggregate_trace(buf.ctypes.data, 1 << 14)
    t = buf[:ns].astype(np.int64)
    t = t[t[:, 10] > 0]
    n_acc = int(((t[0, 3:8]) > 0).sum())   # (slot 8: end of the first accumulator's row loop)
    marks = [0, 1, 2] + [3 + g for g in range(n_acc)] + [9, 10]
    names = ["pass 1 (keys)", "dense prep"] + [f"accumulator {g}" for g in range(n_acc)] + ["pass 3 + merge"]
    names = ["pass 1 (keys)", "dense prep", "(loop entry)"] + [f"accumulator {g}" for g in range(n_acc)] + ["pass 3 + merge"]
    print("aggregate_rows slices", len(t), "span us", (t[:, 10].max() - t[:, 0].min()) / 100.0)
    for i in range(len(marks) - 1):
        d = (t[:, marks[i + 1]] - t[:, marks[i]]) / 100.0
        print(f"  {names[i]:16s} mean {d.mean():7.2f} p50 {np.percentile(d, 50):7.2f} p90 {np.percentile(d, 90):7.2f}")
    print("  total            mean %.2f" % ((t[:, 10] - t[:, 0]).mean() / 100.0))
    print("  (pass 1: setup before the row loop mean %.2f)" % ((t[:, 11] - t[:, 0]).mean() / 100.0))
    if n_acc <= 5:
        print("  (accumulator 0: rows %.2f, reductions + merge into the slice's table %.2f)" % ((t[:, 8] - t[:, 3]).mean() / 100.0, (t[:, 4] - t[:, 8]).mean() / 100.0))
