#!/usr/bin/env python3
"""Ad-hoc timing of TPC-H Q6 as a device-resident operator chain (hyrise_amd/tpch.py run_q6), step by step (debug aid)."""
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
data = tpch.TpchData(scale_factor=float(os.environ.get("SF", "10")), seed=42)
columns = {name: DeviceColumn(column) for name, column in tpch.q6_columns(data).items()}
ex = HipExecutor(torch.device("cuda", 0))


class Timed:
    """Wraps the executor: every operator call is followed by a device synchronise and its wall time is recorded."""

    def __init__(self, inner):
        self.inner, self.log = inner, []

    def __getattr__(self, name):
        f = getattr(self.inner, name)

        def call(*args, **kwargs):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = f(*args, **kwargs)
            torch.cuda.synchronize()
            self.log.append((name, (time.perf_counter() - t) * 1e3))
            return out

        return call


for _ in range(3):
    tpch.run_q6(ex, columns)
timed = Timed(ex)
print(tpch.run_q6(timed, columns))
for name, ms in timed.log:
    print(f"  {name:18s} {ms:8.3f} ms")
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    tpch.run_q6(ex, columns)
torch.cuda.synchronize()
print("q6 ms", (time.perf_counter() - t) * 100)
