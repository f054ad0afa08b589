#!/usr/bin/env python3
"""Scan throughput per segment layout (debug aid): the same 60 M-row column as u8 / u16 / u32 dictionary codes, as an
unencoded int32 value segment and as FrameOfReference, predicate selectivity ~43 %."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hyrise_amd import abi, storage, tpch
from hyrise_amd.operators import make_predicate
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
rows = int(os.environ.get("ROWS", tpch.LINEITEM_ROWS_SF10))
rng = np.random.default_rng(5)
dev = torch.device("cuda", 0)
matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
cases = [("dict u8", rng.integers(0, 200, rows).astype(np.int32), abi.ENC_DICTIONARY, 86),
         ("dict u16", rng.integers(0, 2500, rows).astype(np.int32), abi.ENC_DICTIONARY, 1075),
         ("dict u32", None, abi.ENC_DICTIONARY, 0),
         ("value i32", rng.integers(0, 2500, rows).astype(np.int32), abi.ENC_UNENCODED, 1075),
         ("FoR u8", (np.arange(rows) // 4 + rng.integers(0, 200, rows)).astype(np.int32), abi.ENC_FRAME_OF_REFERENCE, None),
         ("FoR u16", (np.arange(rows) // 4 + rng.integers(0, 2500, rows)).astype(np.int32), abi.ENC_FRAME_OF_REFERENCE, None)]
for name, values, encoding, literal in cases:
    chunk = abi.CHUNK_DEFAULT_SIZE
    if values is None:   # > 65535 distinct values per chunk need bigger chunks
        chunk = 1 << 20
        values = rng.integers(0, 200_000, rows).astype(np.int32)
        literal = 86_000
    if literal is None:
        literal = int(np.quantile(values[::997], 0.43))
    host = storage.make_column(values, None, encoding, chunk_size=chunk)
    column = DeviceColumn(host)
    n_chunks = host.n_chunks
    offsets = torch.zeros(n_chunks + 1, dtype=torch.int64, device=dev)
    counts = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    result = abi.ScanResult()
    result.mem = abi.MEM_DEVICE
    result.matches, result.capacity = matches.data_ptr(), rows
    result.flags = abi.SCAN_CHUNK_REGIONS
    result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, literal)
    for _ in range(3):
        abi.check(lib.hy_table_scan(column.handle, C.byref(predicate), None, 0, C.byref(result)))
    abi.check(lib.hy_set_profiling(1))
    torch.cuda.synchronize()
    for _ in range(10):
        abi.check(lib.hy_table_scan(column.handle, C.byref(predicate), None, 0, C.byref(result)))
    km, ln = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(km), C.byref(ln)))
    abi.check(lib.hy_set_profiling(0))
    m = int(counts.sum().item())
    expected = int((values < literal).sum())
    width = host.segments[0].width
    kernel_ms = km.value / max(1, ln.value)
    print(f"{name:10s} width {width} chunks {n_chunks} matches {m} ({'ok' if m == expected else 'MISMATCH ' + str(expected)}) kernel {kernel_ms * 1e3:.1f} us "
          f"{(rows * width + m * 8) / kernel_ms / 1e6:.0f} GB/s")
    column.close()
