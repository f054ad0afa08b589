#!/usr/bin/env python3
"""Where the time of the fused kernel goes: TPC-H Q1 / Q6 at SF10 with parts of fused_rows switched off (HY_FUSED_DEBUG; results are wrong then)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyrise_amd import abi, tpch
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
data = tpch.TpchData(scale_factor=float(os.environ.get("SF", "10")), seed=42)
q1 = {name: DeviceColumn(column) for name, column in tpch.q1_columns(data).items()}
q6 = {name: DeviceColumn(column) for name, column in tpch.q6_columns(data).items()}


def kernel_ms(run):
    run()
    abi.check(lib.hy_set_profiling(1))
    run()
    ms, launches = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(ms), C.byref(launches)))
    abi.check(lib.hy_set_profiling(0))
    return ms.value


for flags in (0, 1, 3, 7, 15, 16):
    os.environ["HY_FUSED_DEBUG"] = str(flags)
    try:
        a = kernel_ms(lambda: tpch.q1_fused(q1))
    except Exception as e:   # (a result the host cannot read back: the kernel time is what matters here)
        a = float("nan")
    try:
        b = kernel_ms(lambda: tpch.q6_fused(q6))
    except Exception as e:
        b = float("nan")
    print(f"HY_FUSED_DEBUG={flags:2d}   Q1 {a:8.3f} ms   Q6 {b:8.3f} ms", flush=True)
