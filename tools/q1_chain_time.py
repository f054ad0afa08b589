#!/usr/bin/env python3
"""TPC-H Q1 as the operator chain (TableScan, four ArithmeticExpressions materialised over the reference table, AggregateHash of eight
aggregates; tpch.run_q1) at SF10, a few runs -- under `rocprofv3 --kernel-trace --stats` this shows which operator's kernels the chain's
4.3 ms are (bench.py's q1 leg reports the chain beside the fused pass).   usage: python tools/q1_chain_time.py [runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyrise_amd import abi, tpch
from hyrise_amd.distributed import HipExecutor
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
dev = torch.device("cuda:0")
data = tpch.TpchData(scale_factor=float(os.environ.get("SF", "10")), seed=42)
columns = {name: DeviceColumn(column) for name, column in tpch.q1_columns(data).items()}
ex = HipExecutor(dev)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tpch.run_q1(ex, columns)
for _ in range(runs):
    torch.cuda.synchronize()
    t = time.perf_counter()
    result = tpch.run_q1(ex, columns)
    torch.cuda.synchronize()
    print(f"Q1 chain {1e3 * (time.perf_counter() - t):.3f} ms, {result.n_groups} groups")
