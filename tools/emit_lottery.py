#!/usr/bin/env python3
"""What decides pk_emit's time from process to process (230 .. 305 us for the same join)?  One process, the SF10 join into several output
arenas that are alive at the same time, with the library's temporaries (rank table, count cells) re-allocated in between: if the time
follows the arena, it is where the output lists lie; if it follows the temporaries, it is those.
Usage: python tools/emit_lottery.py [arenas] [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    arenas = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders_host = storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)
    lineitem_host = storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
    n = data.n_lineitems
    orders, lineitem = [], []
    if os.environ.get("ARENAS_FIRST"):   # the output arenas before anything else is in device memory
        joins = [bench.device_join(lib, torch, dev, orders, lineitem, n) for _ in range(arenas)]
    orders += [DeviceColumn(orders_host) for _ in range(3)]
    lineitem += [DeviceColumn(lineitem_host) for _ in range(3)]
    if not os.environ.get("ARENAS_FIRST"):
        joins = [bench.device_join(lib, torch, dev, orders, lineitem, n) for _ in range(arenas)]
    if os.environ.get("SHARE_SMALL"):   # every join writes its PosList offsets and its status into the buffers of join number SHARE_SMALL
        donor = joins[int(os.environ["SHARE_SMALL"])][1]
        for run, r, keep in joins:
            r.slice_offsets, r.status = donor.slice_offsets, donor.status
    for run, r, keep in joins:
        for _ in range(6):
            run()

    def measure(label):
        out = []
        for i, (run, r, keep) in enumerate(joins):
            dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
            out.append(kinds["join_probe"][0] * 1e3)
        print(f"{label:44s} pk_emit per arena: " + "  ".join(f"{v:6.1f}" for v in out) + f"   (arena addresses mod 1 GiB: " + " ".join(f"{(k[3].data_ptr() >> 21) & 511}" for _, _, k in joins) + ")", flush=True)

    if os.environ.get("TILE_GROUPS"):   # HY_OPT_JOIN_EMIT_TILE_GROUP: the same arenas under every setting
        for group in [int(g) for g in os.environ["TILE_GROUPS"].split(",")] * 2:
            abi.check(lib.hy_set_option(abi.OPT_JOIN_EMIT_TILE_GROUP, group))
            measure(f"tile group {group}")
        return
    measure("as allocated")
    measure("again")
    # the library's temporaries: hy_shutdown releases this thread's pool, the next join allocates anew
    pads = []
    for round_ in range(3):
        lib.hy_shutdown()
        pads.append(torch.empty((round_ + 1) * 37 * (1 << 20) + 12345, dtype=torch.uint8, device=dev))   # (moves what hipMalloc hands out next)
        abi.check(lib.hy_init(0))
        for run, r, keep in joins[:1]:
            for _ in range(6):
                run()
        measure(f"temporaries re-allocated ({round_ + 1})")


if __name__ == "__main__":
    main()
