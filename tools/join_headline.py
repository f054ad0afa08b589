#!/usr/bin/env python3
"""Config 3 (JoinHash orders x lineitem, SF10) on the default path, a few joins: something for rocprofv3 to look at (tools/sq_counters.sh).
Usage: python tools/join_headline.py [joins]   (not part of the product)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    joins = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders = DeviceColumn(storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED))
    lineitem = DeviceColumn(storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE))
    run, r, keep = bench.device_join(lib, torch, dev, orders, lineitem, data.n_lineitems, asynchronous=True)
    dt, kinds = bench.timed_kernel(lib, torch, run, joins, all_kinds=True)
    run.finish()
    print(f"{dt * 1e3:7.3f} ms/join  pairs {int(r.n_pairs)}  " + "  ".join(f"{k} {v[0] * 1e3:6.1f} us" for k, v in kinds.items() if v[1]), flush=True)


if __name__ == "__main__":
    main()
