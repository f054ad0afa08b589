#!/usr/bin/env python3
"""Ad-hoc timing of a join whose build keys have duplicates (every probe row has four partners: the tiles go through
probe_emit_generic) -- debug aid."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hyrise_amd import abi, storage
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
rng = np.random.default_rng(5)
distinct, copies, probe_rows = 1_000_000, 4, 16_000_000
build_keys = np.repeat(np.arange(distinct, dtype=np.int32), copies)
rng.shuffle(build_keys)
probe_keys = rng.integers(0, distinct, probe_rows).astype(np.int32)
build = DeviceColumn(storage.make_column(build_keys, None, abi.ENC_UNENCODED))
probe = DeviceColumn(storage.make_column(probe_keys, None, abi.ENC_UNENCODED))
n = probe_rows * copies
dev = torch.device("cuda")
left = torch.empty((n, 2), dtype=torch.int32, device=dev)
right = torch.empty((n, 2), dtype=torch.int32, device=dev)
so = torch.zeros(4000, dtype=torch.int64, device=dev)
r = abi.JoinResult()
r.mem, r.radix_bits, r.left_pos, r.right_pos, r.capacity, r.slice_offsets, r.slice_capacity = abi.MEM_DEVICE, 0xFFFFFFFF, left.data_ptr(), right.data_ptr(), n, so.data_ptr(), 3990
for i in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    abi.check(lib.hy_join_hash(build.handle, probe.handle, abi.JOIN_INNER, C.byref(r)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("join ms %.3f pairs %d slices %d radix %d pairs/s %.3g" % (dt * 1e3, r.n_pairs, r.n_slices, r.radix_bits, r.n_pairs / dt))
