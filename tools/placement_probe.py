#!/usr/bin/env python3
"""pk_emit in EVERY candidate placement of one process, under a few settings: which allocations are fast, and does a setting move the slow
ones?  Usage: python tools/placement_probe.py [candidates]   (SF10 orders x lineitem, HIP events around four asynchronous joins per cell)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_candidates = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    import ctypes as C
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
    candidates = [pair_lists(torch, dev, n) for _ in range(n_candidates)]
    slice_offsets = torch.zeros(8192, dtype=torch.int64, device=dev)
    status = torch.zeros(4, dtype=torch.int64, device=dev)
    r = abi.JoinResult()
    r.mem, r.capacity = abi.MEM_DEVICE, n
    r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000
    r.flags, r.status = abi.JOIN_ASYNC, status.data_ptr()
    turn = [0]

    def join():
        r.radix_bits = 0xFFFFFFFF
        i = turn[0] % 3
        turn[0] += 1
        abi.check(lib.hy_join_hash(orders[i].handle, lineitem[i].handle, abi.JOIN_INNER, C.byref(r)))

    def cell(left, right):
        r.left_pos, r.right_pos = left.data_ptr(), right.data_ptr()
        for _ in range(3):
            join()
        started, stopped = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        started.record()
        for _ in range(6):
            join()
        stopped.record()
        torch.cuda.synchronize()
        return started.elapsed_time(stopped) / 6

    print("addresses (GiB):", " ".join(f"{c[0].data_ptr() / 2**30:7.2f}" for c in candidates))
    for name, switches in (("default", {}), ("tile group 8", {"HY_JOIN_EMIT_TILE_GROUP": "8"}), ("tile group 512", {"HY_JOIN_EMIT_TILE_GROUP": "512"}),
                           ("write-back stores", {"HY_JOIN_STORES": "1"}), ("default again", {})):
        with abi.switches(switches):
            print(f"{name:20s}", " ".join(f"{cell(left, right):7.4f}" for left, right, _ in candidates), flush=True)
    # every list an allocation of its own (2 n of them, allocated now): pairs (2 k, 2 k + 1), then each against the FIRST of them
    singles = [torch.empty((n, 2), dtype=torch.int32, device=dev) for _ in range(2 * n_candidates)]
    print("single addresses (GiB):", " ".join(f"{t.data_ptr() / 2**30:7.2f}" for t in singles))
    print("singles 2k, 2k+1      ", " ".join(f"{cell(singles[2 * k], singles[2 * k + 1]):7.4f}" for k in range(n_candidates)), flush=True)
    print("single 0 with single j", " ".join(f"{cell(singles[0], singles[j]):7.4f}" for j in range(1, 2 * n_candidates)), flush=True)
    print("single j with single 0", " ".join(f"{cell(singles[j], singles[0]):7.4f}" for j in range(1, 2 * n_candidates)), flush=True)
    # the lists of candidate i and i + 1 crossed: is it the left list's place, the right one's, or the pair?
    print("left of i, right of i+1", " ".join(f"{cell(candidates[i][0], candidates[(i + 1) % n_candidates][1]):7.4f}" for i in range(n_candidates)), flush=True)


if __name__ == "__main__":
    main()
