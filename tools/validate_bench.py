#!/usr/bin/env python3
"""Timing of hy_validate on an SF10-lineitem-shaped table (debug aid): 59 986 052 rows of MvccData, 5 % of the rows
invalidated before the snapshot, chunk shortcut off (every row is tested) and on."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hyrise_amd import abi, storage, tpch
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
rows = tpch.LINEITEM_ROWS_SF10
rng = np.random.default_rng(8)
tids = np.zeros(rows, dtype=np.uint32)
begins = rng.integers(1, 1000, rows).astype(np.uint32)
ends = np.where(rng.random(rows) < 0.05, rng.integers(1, 2000, rows), storage.MAX_COMMIT_ID).astype(np.uint32)
host = storage.make_mvcc_column(tids, begins, ends)
column = DeviceColumn(host)
dev = torch.device("cuda", 0)
matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
offsets = torch.zeros(host.n_chunks + 1, dtype=torch.int64, device=dev)
counts = torch.zeros(host.n_chunks, dtype=torch.int32, device=dev)
result = abi.ScanResult()
result.mem = abi.MEM_DEVICE
result.matches, result.capacity = matches.data_ptr(), rows
result.flags = abi.SCAN_CHUNK_REGIONS
result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()
snapshot, our_tid = 1500, 7
expected = int(((snapshot < ends) & ((snapshot >= begins) != (tids == our_tid))).sum())
for shortcut in (0, 1):
    for _ in range(3):
        abi.check(lib.hy_validate(column.handle, our_tid, snapshot, shortcut, C.byref(result)))
    abi.check(lib.hy_set_profiling(1))
    torch.cuda.synchronize()
    for _ in range(10):
        abi.check(lib.hy_validate(column.handle, our_tid, snapshot, shortcut, C.byref(result)))
    km, ln = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(km), C.byref(ln)))
    abi.check(lib.hy_set_profiling(0))
    visible = int(counts.sum().item())
    kernel_ms = km.value / max(1, ln.value)
    print(f"shortcut {shortcut}: visible {visible} ({'ok' if visible == expected else 'MISMATCH ' + str(expected)}) kernel {kernel_ms * 1e3:.1f} us "
          f"{(rows * 12 + visible * 8) / kernel_ms / 1e6:.0f} GB/s algorithmic")
