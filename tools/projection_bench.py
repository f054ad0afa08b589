#!/usr/bin/env python3
"""Timing of hy_projection_arithmetic on SF10 lineitem (debug aid): l_extendedprice * l_discount and
l_extendedprice * (1 - l_discount) as two chained projections, float value segments."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import projection_arithmetic
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
data = tpch.TpchData(scale_factor=10.0, seed=42)
n = data.n_lineitems
price = DeviceColumn(storage.make_column(data.l_extendedprice, None, abi.ENC_UNENCODED))
discount = DeviceColumn(storage.make_column(data.l_discount, None, abi.ENC_UNENCODED))
for _ in range(2):
    projection_arithmetic(abi.ARITH_MUL, price, discount).close()
abi.check(lib.hy_set_profiling(1))
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 10
for _ in range(steps):
    projection_arithmetic(abi.ARITH_MUL, price, discount).close()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
km, ln = C.c_float(0), C.c_uint32(0)
abi.check(lib.hy_profile_read(C.byref(km), C.byref(ln)))
abi.check(lib.hy_set_profiling(0))
kernel_ms = km.value / max(1, ln.value)
bytes_moved = n * (4 + 4 + 4) + n // 8
print(f"price * discount: {dt * 1e3:.3f} ms per call, kernel {kernel_ms * 1e3:.1f} us, {bytes_moved / kernel_ms / 1e6:.0f} GB/s algorithmic")
product = projection_arithmetic(abi.ARITH_MUL, price, discount)
values, nulls = product.read()
expected = data.l_extendedprice * data.l_discount
print("check:", "ok" if values.tobytes() == expected.tobytes() and not nulls.any() else "MISMATCH")
