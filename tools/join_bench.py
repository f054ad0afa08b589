#!/usr/bin/env python3
"""Config 3 (JoinHash orders x lineitem, SF10) under the library's debug switches, one process: ms per join and the HIP-event time of
the timed kernels.  Usage: python tools/join_bench.py [steps]   (not part of the product; the switches are options, include/hyrise_amd.h HY_OPT_*; the "debug:" variants and HY_JOIN_TRACE need a -DHY_DEBUG_SWITCHES build)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders = DeviceColumn(storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED))
    lineitem = DeviceColumn(storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE))
    n = data.n_lineitems
    variants = [("default", {}), ("no key hint (two-pass build)", {"HY_JOIN_NO_HINT": "1"}), ("fill: one workgroup per slice", {"HY_JOIN_FILL_WGS_PER_CU": "0"}),
                ("fill: waves, 2 per CU", {"HY_JOIN_FILL_WGS_PER_CU": "2"}), ("fill: waves, 8 per CU", {"HY_JOIN_FILL_WGS_PER_CU": "8"}),
                ("general rank-table kernels (round 2)", {"HY_JOIN_NO_PKFK": "1", "HY_JOIN_NO_HINT": "1"}), ("default again", {})]
    for name, env in variants:
        switches = abi.switches(env)
        switches.__enter__()
        run, r, keep = bench.device_join(lib, torch, dev, orders, lineitem, n)
        dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
        print(f"{name:40s} {dt * 1e3:7.3f} ms/join  pairs {int(r.n_pairs)}  " + "  ".join(f"{k} {v[0] * 1e3:6.1f} us" for k, v in kinds.items() if v[1]), flush=True)
        switches.__exit__(None, None, None)
        del keep
    if os.environ.get("HY_JOIN_TRACE"):   # per-tile phase stamps of pk_emit of the last join (wall clock, 100 MHz)
        import numpy as np
        lib.hy_debug_join_trace.restype = C.c_int
        run, r, keep = bench.device_join(lib, torch, dev, orders, lineitem, n)
        run()
        stamps = np.zeros((1 << 15, 6), dtype=np.uint64)
        tiles = lib.hy_debug_join_trace(stamps.ctypes.data_as(C.c_void_p), C.c_uint32(1 << 15))
        t = stamps[:tiles].astype(np.float64) / 100.0   # us
        t = t[t[:, 5] > 0]
        phases = ("evaluate", "reserve + rank", "prefix", "stage", "copy out")
        print(f"pk_emit trace over {len(t)} tiles: span {t[:, 5].max() - t[:, 0].min():.1f} us, per tile " +
              ", ".join(f"{name} {np.median(t[:, i + 1] - t[:, i]):.2f}" for i, name in enumerate(phases)) + f", total {np.median(t[:, 5] - t[:, 0]):.2f} us (medians)")
        del keep
    for name, left, right, capacity in (("semi: probe lineitem, build orders", lineitem, orders, n), ("semi: probe orders, build lineitem", orders, lineitem, data.n_orders)):
        run, r, keep = bench.device_join(lib, torch, dev, left, right, capacity, abi.JOIN_SEMI)
        dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
        print(f"{name:40s} {dt * 1e3:7.3f} ms/join  matches {int(r.n_pairs)}  " + "  ".join(f"{k} {v[0] * 1e3:6.1f} us" for k, v in kinds.items() if v[1]), flush=True)
        del keep


if __name__ == "__main__":
    main()
