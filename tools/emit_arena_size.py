#!/usr/bin/env python3
"""pk_emit against the SIZE of the allocation that holds the output lists: does memory that the driver can back with one large contiguous
block (a power-of-two size) translate addresses more cheaply than 966 MB pieced together?  Usage: python tools/emit_arena_size.py [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
    second_base = (list_bytes + 2 * MiB - 1) // (2 * MiB) * (2 * MiB) + 5 * MiB // 4
    need = second_base + list_bytes
    slice_offsets = torch.zeros(8192, dtype=torch.int64, device=dev)
    turn = [0]
    keep = []
    sizes = [need + 6 * MiB, 1024 * MiB, 2048 * MiB, need + 6 * MiB, 1024 * MiB, 2048 * MiB, 4096 * MiB, 1024 * MiB, need + 6 * MiB, 2048 * MiB]
    first = True
    for size in sizes:
        arena = torch.empty(size, dtype=torch.uint8, device=dev)
        keep.append(arena)
        base = -arena.data_ptr() % (2 * MiB)
        if base + need > size:
            base = 0
        r = abi.JoinResult()
        r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
        r.left_pos, r.right_pos, r.capacity = arena.data_ptr() + base, arena.data_ptr() + base + second_base, n
        r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000

        def run():
            r.radix_bits = 0xFFFFFFFF
            i = turn[0] % 3
            turn[0] += 1
            abi.check(lib.hy_join_hash(orders[i].handle, lineitem[i].handle, abi.JOIN_INNER, C.byref(r)))
        if first:
            for _ in range(6):
                run()
            first = False
        dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
        print(f"arena of {size / MiB:7.1f} MiB at {arena.data_ptr():#x} (aligned to {arena.data_ptr() & -arena.data_ptr():#x}): pk_emit {kinds['join_probe'][0] * 1e3:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
