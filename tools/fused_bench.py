#!/usr/bin/env python3
"""TPC-H Q6 and Q1 at SF (default 10): the operator chain on the device beside hy_scan_project_aggregate (one pass), wall clock
per query and the fused kernel's HIP-event time (debug aid; bench.py reports the same numbers in its `q6` / `q1` objects)."""
import ctypes as C
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
ex = HipExecutor(torch.device("cuda", 0))
repeats = int(os.environ.get("REPEATS", "10"))


def timed(name, run):
    run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(repeats):
        out = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / repeats * 1e3
    abi.check(lib.hy_set_profiling(1))
    run()
    kernel_ms, launches = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(kernel_ms), C.byref(launches)))
    abi.check(lib.hy_set_profiling(0))
    print(f"{name:34s} {ms:8.3f} ms   (dominant kernels of its operators: {kernel_ms.value:7.3f} ms in {launches.value} launches)")
    return out, ms


q6 = {name: DeviceColumn(column) for name, column in tpch.q6_columns(data).items()}
chain, _ = timed("Q6 operator chain", lambda: tpch.run_q6(ex, q6))
fused, _ = timed("Q6 fused", lambda: tpch.q6_fused(q6))
print("   chain", chain, "fused", fused)
assert chain[1] == fused[1] and abs(chain[0] - fused[0]) <= 1e-9 * abs(chain[0])
q6_bytes = sum(s.size * s.width for name in ("l_shipdate", "l_discount", "l_quantity") for s in tpch.q6_columns(data)[name].segments) if os.environ.get("BYTES") else 0

q1 = {name: DeviceColumn(column) for name, column in tpch.q1_columns(data).items()}
chain, _ = timed("Q1 operator chain", lambda: tpch.run_q1(ex, q1))
fused, _ = timed("Q1 fused", lambda: tpch.q1_fused(q1))
assert chain.n_groups == fused.n_groups
for a, name in enumerate(tpch.Q1_AGGREGATES):
    for x, y in zip(chain.column(a), fused.column(a)):
        assert abs(x - y) <= 1e-9 * max(1.0, abs(y)), (name, x, y)
print("   groups", fused.n_groups, "count_order", fused.column(7))
