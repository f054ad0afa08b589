#!/usr/bin/env python3
"""hy_column_create's upload rate: pageable numpy segments -> one arena in HBM, through the pinned windows.  Usage: python tools/upload_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    rng = np.random.default_rng(1)
    n = tpch.LINEITEM_ROWS_SF10
    for name, values, encoding in (("int32 values, 240 MB", rng.integers(0, 1 << 30, n).astype(np.int32), abi.ENC_UNENCODED),
                                   ("u16 value ids + dictionaries, 130 MB", rng.integers(0, 2500, n).astype(np.int32), abi.ENC_DICTIONARY)):
        host = storage.make_column(values, None, encoding)
        payload = sum(sum(int(a.nbytes) for a in (seg.data, seg.aux, seg.nulls) if a is not None and hasattr(a, "nbytes")) for seg in host.segments)
        DeviceColumn(host).close()
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            column = DeviceColumn(host)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            column.close()
        dt = sorted(times)[2]
        print(f"{name:40s} {payload / 1e6:7.1f} MB  {dt * 1e3:7.2f} ms  {payload / dt / 1e9:6.1f} GB/s")


if __name__ == "__main__":
    main()
