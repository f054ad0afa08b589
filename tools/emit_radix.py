#!/usr/bin/env python3
"""pk_emit against the number of output streams: the SF10 join with radix_bits 0 .. 8 (1 .. 256 partitions; dbgen's order keys populate a
quarter of them) into three arenas.  Usage: python tools/emit_radix.py [steps]"""
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
    from hyrise_amd.operators import pair_lists
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders = [DeviceColumn(storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)) for _ in range(3)]
    lineitem = [DeviceColumn(storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)) for _ in range(3)]
    n = data.n_lineitems
    slice_offsets = torch.zeros(8192, dtype=torch.int64, device=dev)
    turn = [0]
    arenas = [pair_lists(torch, dev, n) for _ in range(3)]
    for bits in (7, 0, 2, 4, 5, 6, 7, 8):
        line = []
        for left, right, arena in arenas:
            r = abi.JoinResult()
            r.mem = abi.MEM_DEVICE
            r.left_pos, r.right_pos, r.capacity = left.data_ptr(), right.data_ptr(), n
            r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000

            def run():
                r.radix_bits = bits
                i = turn[0] % 3
                turn[0] += 1
                abi.check(lib.hy_join_hash(orders[i].handle, lineitem[i].handle, abi.JOIN_INNER, C.byref(r)))
            for _ in range(3):
                run()
            dt, kinds = bench.timed_kernel(lib, torch, run, steps, all_kinds=True)
            line.append(kinds["join_probe"][0] * 1e3)
        print(f"radix_bits {bits}: pk_emit per arena " + "  ".join(f"{v:6.1f}" for v in line), flush=True)


if __name__ == "__main__":
    main()
