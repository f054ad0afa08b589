#!/usr/bin/env python3
"""pk_emit against where the output lists start: ONE allocation, both lists shifted together by k x 2 MiB (and by odd amounts), and the second
list alone -- is a slow arena (tools/emit_lottery.py) slow because of the address bits above 2 MiB, or because of the memory behind it?
Usage: python tools/emit_shift.py [arenas] [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    arenas = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders = [DeviceColumn(storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)) for _ in range(3)]
    lineitem = [DeviceColumn(storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)) for _ in range(3)]
    n = data.n_lineitems
    MiB = 1 << 20
    list_bytes = 8 * n
    slack = 80 * MiB
    slice_offsets = torch.zeros(8192, dtype=torch.int64, device=dev)
    turn = [0]
    for a in range(arenas):
        arena = torch.empty(2 * list_bytes + slack, dtype=torch.uint8, device=dev)
        base = -arena.data_ptr() % (2 * MiB)
        second_base = (list_bytes + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
        print(f"arena {a} at 2 MiB page {(arena.data_ptr() + base) >> 21}")

        def time_at(first, second):
            r = abi.JoinResult()
            r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
            r.left_pos, r.right_pos, r.capacity = arena.data_ptr() + first, arena.data_ptr() + second, n
            r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000

            def run():
                r.radix_bits = 0xFFFFFFFF
                i = turn[0] % 3
                turn[0] += 1
                abi.check(lib.hy_join_hash(orders[i].handle, lineitem[i].handle, abi.JOIN_INNER, C.byref(r)))
            dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
            return kinds["join_probe"][0] * 1e3

        for _ in range(6):
            time_at(base, base + second_base + 5 * MiB // 4)
        shifts = [0, 2, 4, 6, 8, 16, 32, 3, 1, 0]
        print("  both lists shifted by k MiB:      " + "  ".join(f"k={k}: {time_at(base + k * MiB, base + second_base + 5 * MiB // 4 + k * MiB):6.1f}" for k in shifts), flush=True)
        print("  second list alone shifted by k MiB: " + "  ".join(f"k={k}: {time_at(base, base + second_base + 5 * MiB // 4 + k * MiB):6.1f}" for k in (0, 2, 4, 8, 16, 32, 64)), flush=True)
        keep = arena   # (the next arena lies elsewhere)
        globals().setdefault("_keep", []).append(keep)


if __name__ == "__main__":
    main()
