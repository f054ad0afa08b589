#!/usr/bin/env python3
"""Config 3 (JoinHash orders x lineitem, SF10) under the library's debug switches, one process: ms per join and the HIP-event time of
the timed kernels.  Usage: python tools/join_bench.py [steps]   (not part of the product; the switches are documented in DESIGN.md section 6)"""
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
    variants = [("default", {}), ("plain stores", {"HY_JOIN_PLAIN_STORES": "1"}), ("no key hint (two-pass build)", {"HY_JOIN_NO_HINT": "1"}),
                ("general rank-table kernels (round 2)", {"HY_JOIN_NO_PKFK": "1", "HY_JOIN_NO_HINT": "1"}), ("default again", {})]
    for name, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        run, r, keep = bench.device_join(lib, torch, dev, orders, lineitem, n)
        dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
        print(f"{name:40s} {dt * 1e3:7.3f} ms/join  pairs {int(r.n_pairs)}  " + "  ".join(f"{k} {v[0] * 1e3:6.1f} us" for k, v in kinds.items() if v[1]), flush=True)
        for k in env:
            del os.environ[k]
        del keep
    for name, left, right, capacity in (("semi: probe lineitem, build orders", lineitem, orders, n), ("semi: probe orders, build lineitem", orders, lineitem, data.n_orders)):
        run, r, keep = bench.device_join(lib, torch, dev, left, right, capacity, abi.JOIN_SEMI)
        dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
        print(f"{name:40s} {dt * 1e3:7.3f} ms/join  matches {int(r.n_pairs)}  " + "  ".join(f"{k} {v[0] * 1e3:6.1f} us" for k, v in kinds.items() if v[1]), flush=True)
        del keep


if __name__ == "__main__":
    main()
