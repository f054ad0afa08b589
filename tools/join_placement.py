#!/usr/bin/env python3
"""Config 3 with the two output PosLists carved out of ONE allocation at different relative offsets: does pk_emit's time depend on where
the build-side list sits relative to the probe-side list (both are written at the same pair index at the same time)?
Usage: python tools/join_placement.py [steps]   (not part of the product)"""
import ctypes as C
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
    list_bytes = 8 * n
    slack = 64 << 20
    arena = torch.empty(2 * list_bytes + 2 * slack, dtype=torch.uint8, device=dev)
    base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)   # 2 MiB aligned
    slice_offsets = torch.zeros(8192, dtype=torch.int64, device=dev)
    print(f"arena at {arena.data_ptr():#x}, first list at {base:#x}, lists of {list_bytes} bytes")
    skews = [k << 18 for k in range(0, 17)] + [5 << 20, 7 << 20, 9 << 20, 17 << 20, 33 << 20, 63 << 20]
    if len(sys.argv) > 2 and sys.argv[2] == "absolute":   # both lists moved together, 1.25 MiB apart (mod 2 MiB): does the absolute position matter?
        origin = base
        for shift in [k << 18 for k in range(0, 33, 2)] + [(16 << 20) + (k << 20) for k in (0, 1, 3, 8, 16, 31)]:
            first = origin + shift
            second = (first + list_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20) + (shift & ((2 << 20) - 1)) + (5 << 18)
            r = abi.JoinResult()
            r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
            r.left_pos, r.right_pos, r.capacity = first, second, n
            r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000

            def run_shifted():
                r.radix_bits = 0xFFFFFFFF
                abi.check(lib.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(r)))
            dt, kinds = bench.timed_kernel(lib, torch, run_shifted, steps, all_kinds=True)
            print(f"both lists + {shift:9d} B: {dt * 1e3:7.3f} ms/join  " + "  ".join(f"{k} {v[0] * 1e3:6.1f} us" for k, v in kinds.items() if v[1]), flush=True)
        return
    if len(sys.argv) > 2:
        base += int(sys.argv[2]) << 18   # (the first list itself off the 2 MiB grid, in units of 256 KiB)
    for skew in skews:
        second = (base + list_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20) + skew   # 2 MiB aligned + skew
        r = abi.JoinResult()
        r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
        r.left_pos, r.right_pos, r.capacity = base, second, n
        r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000

        def run():
            r.radix_bits = 0xFFFFFFFF
            abi.check(lib.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(r)))
        dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
        print(f"second list at 2 MiB boundary + {skew:9d} B: {dt * 1e3:7.3f} ms/join  " + "  ".join(f"{k} {v[0] * 1e3:6.1f} us" for k, v in kinds.items() if v[1]), flush=True)


if __name__ == "__main__":
    main()
