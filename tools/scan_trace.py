#!/usr/bin/env python3
"""Per-workgroup timeline of scan_slices (debug aid, not part of the product): runs the bench.py scan with
HY_SCAN_TRACE=1 and prints when workgroups start / finish (wall_clock64 ticks, 100 MHz -> 10 ns)."""
import ctypes as C
import os
import sys

os.environ["HY_SCAN_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
from hyrise_amd import abi, tpch
from hyrise_amd.operators import make_predicate
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
rows = tpch.LINEITEM_ROWS_SF10
days, host_column = tpch.shipdate_column(rows, seed=42)
column = DeviceColumn(host_column)
n_chunks = host_column.n_chunks
dev = torch.device("cuda", 0)
matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
offsets = torch.zeros(n_chunks + 1, dtype=torch.int64, device=dev)
counts = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
result = abi.ScanResult()
result.mem = abi.MEM_DEVICE
result.matches, result.capacity = matches.data_ptr(), rows
result.flags = abi.SCAN_CHUNK_REGIONS
result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()
value = int(sys.argv[1]) if len(sys.argv) > 1 else tpch.DAY_1995_01_01
predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, value)
lib.hy_debug_scan_trace.argtypes = [C.c_void_p, C.c_uint32]
lib.hy_debug_scan_trace.restype = C.c_int
for rep in range(4):
    abi.check(lib.hy_table_scan(column.handle, C.byref(predicate), None, 0, C.byref(result)))
    torch.cuda.synchronize()
buf = np.zeros((4096, 4), dtype=np.uint64)
n = lib.hy_debug_scan_trace(buf.ctypes.data, 4096)
t = buf[:n].astype(np.int64)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0
first_part = (t[:, 1] - t0) / 100.0
end = (t[:, 3] - t0) / 100.0
pct = lambda a: " ".join(f"{np.percentile(a, p):7.2f}" for p in (0, 10, 50, 90, 100))
print(f"workgroups {n}")
print("start  us  (p0 p10 p50 p90 p100):", pct(start))
print("end    us                       :", pct(end))
print("duration us                     :", pct(end - start))
print("first part done us              :", pct(first_part))
order = np.argsort(start)
print("by blockIdx (every 64th): start, end")
for b in range(0, n, 64):
    print(f"  wg {b:4d}: {start[b]:7.2f} {end[b]:7.2f}")

