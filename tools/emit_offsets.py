#!/usr/bin/env python3
"""pk_emit against the placement of the join's two output PosLists, with everything else fixed: ONE process, ONE arena aligned to 1 GiB, the
first list at the arena's start (+ an optional shift), the second `1 GiB + offset` behind it -- the SF10 orders x lineitem join, its
kernels timed with HIP events.  (tools/emit_lottery.py showed that arenas aligned to 1 GiB all behave alike: the time is a function of
the address bits, not of the allocation.)
Usage: python tools/emit_offsets.py [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

GIB = 1 << 30


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders_host = storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)
    lineitem_host = storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
    orders = [DeviceColumn(orders_host) for _ in range(3)]
    lineitem = [DeviceColumn(lineitem_host) for _ in range(3)]
    n = data.n_lineitems
    far = int(os.environ.get("FAR_GIB", "0"))   # FAR_GIB=N: only the sweep of both lists through an N GiB arena, in steps of 512 MiB
    arena = torch.empty((far if far else 5) * GIB, dtype=torch.uint8, device=dev)
    base = arena.data_ptr() + (-arena.data_ptr() % GIB)
    slice_offsets = torch.zeros(8192, dtype=torch.int64, device=dev)
    status = torch.zeros(4, dtype=torch.int64, device=dev)
    print(f"arena base {base:#x}", flush=True)

    def measure(shift, offset):
        r = abi.JoinResult()
        r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
        r.left_pos, r.right_pos, r.capacity = base + shift, base + shift + GIB + offset, n
        r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000
        r.flags, r.status = abi.JOIN_ASYNC, status.data_ptr()
        turn = [0]

        def run():
            r.radix_bits = 0xFFFFFFFF
            i = turn[0] % 3
            turn[0] += 1
            abi.check(lib.hy_join_hash(orders[i].handle, lineitem[i].handle, abi.JOIN_INNER, C.byref(r)))

        dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
        return kinds["join_probe"][0] * 1e3

    KIB, MIB = 1 << 10, 1 << 20
    if os.environ.get("FINE"):
        print("## second list at first + 1 GiB + offset: fine steps")
        for offset in [k * 256 for k in range(0, 17)] + [k * 4 * KIB for k in range(2, 33)] + [k * 16 * KIB for k in range(9, 33)]:
            print(f"offset {offset / KIB:9.2f} KiB   pk_emit {measure(0, offset):6.1f} us", flush=True)
        return
    if far:
        print("## both lists shifted together through the arena (second list 1 GiB + 1.25 MiB behind the first)")
        for half in range(2 * int(os.environ.get("FAR_FROM", "0")), 2 * (far - 3)):
            print(f"shift {half / 2:6.1f} GiB   pk_emit {measure(half * GIB // 2, 5 << 18):6.1f} us", flush=True)
        return
    print("## second list at first + 1 GiB + offset (first list at the 1 GiB boundary)")
    offsets = [k * 128 * KIB for k in range(0, 17)] + [k * MIB for k in (3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768)]
    for offset in offsets:
        print(f"offset {offset / MIB:9.3f} MiB   pk_emit {measure(0, offset):6.1f} us", flush=True)
    print("## both lists shifted together (offset 0)")
    for shift in [k * 256 * KIB for k in range(0, 9)] + [k * MIB for k in (4, 8, 16, 32, 64, 128, 256, 512)]:
        print(f"shift {shift / MIB:9.3f} MiB   pk_emit {measure(shift, 0):6.1f} us", flush=True)
    print("## second list `offset` BEFORE first + 1 GiB")
    for offset in [k * 256 * KIB for k in range(1, 9)] + [k * MIB for k in (4, 16, 64, 256)]:
        print(f"offset {-offset / MIB:9.3f} MiB   pk_emit {measure(0, -offset):6.1f} us", flush=True)


if __name__ == "__main__":
    main()
