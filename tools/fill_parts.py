#!/usr/bin/env python3
"""Where the time of the hinted build goes (rank_table_fill_waves), parts switched off one after the other: needs a -DHY_DEBUG_SWITCHES build
(tools/build_variant.sh debug -DHY_DEBUG_SWITCHES; HY_LIBRARY=hyrise_amd/variants/lib_debug.so).  Results are wrong with a part off:
only the kernel's HIP-event time is printed.  Usage: HY_LIBRARY=... python tools/fill_parts.py [steps]"""
import os
import sys

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
    run, r, keep = bench.device_join(lib, torch, dev, orders, lineitem, n)
    run()
    run()
    for name, debug in (("everything", 0), ("no filter atomics", 1), ("no table stores", 2), ("no filter, no table stores", 3), ("no LDS work (and nothing after it)", 4),
                        ("loads and order checks only", 8), ("everything", 0)):
        os.environ["HY_JOIN_FILL_DEBUG"] = str(debug)
        abi.check(lib.hy_set_profiling(1))
        torch.cuda.synchronize()
        for _ in range(steps):
            try:
                run()
            except abi.HyriseAmdError:
                pass
        torch.cuda.synchronize()
        kinds = bench.kernel_times(lib)
        abi.check(lib.hy_set_profiling(0))
        print(f"{name:40s} fill {kinds['join_build'][0] * 1e3:6.1f} us over {kinds['join_build'][1]} launches", flush=True)
    del keep


if __name__ == "__main__":
    main()
