#!/usr/bin/env python3
"""Ad-hoc timing of AggregateHash with many groups (debug aid): GROUP BY one int32 key with G distinct values over N rows,
SUM + COUNT(*) of a float column."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hyrise_amd import abi, storage
from hyrise_amd.operators import aggregate_hash
from hyrise_amd.storage import DeviceColumn

lib = abi.load_library()
abi.check(lib.hy_init(0))
rng = np.random.default_rng(9)
n = int(os.environ.get("ROWS", "60000000"))
values = DeviceColumn(storage.make_column(rng.random(n).astype(np.float32), None, abi.ENC_UNENCODED))
for groups in [int(g) for g in os.environ.get("NGROUPS", "4,64,1000,100000,4000000").split(",")]:
    keys = DeviceColumn(storage.make_column(rng.integers(0, groups, n).astype(np.int32), None, abi.ENC_UNENCODED))
    result = None
    for i in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        result = aggregate_hash([keys], [(abi.AGG_SUM, values), (abi.AGG_COUNT, None)], group_capacity=groups + 16, result=result if os.environ.get("FRESH") is None else None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(f"groups {groups:8d}: {dt * 1e3:8.3f} ms  {n / dt:.3g} rows/s  ({result.n_groups} groups)", flush=True)
    keys.close()
