#!/usr/bin/env python3
"""Config 2 (TableScan l_shipdate < 1995-01-01, SF10, three column copies in rotation) under the scan's debug switches, one process:
ms per scan and the HIP-event time of scan_slices.  Usage: python tools/scan_ab.py [steps]   (not part of the product)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import torch
    from hyrise_amd import abi, tpch
    from hyrise_amd.operators import make_predicate
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    rows = tpch.LINEITEM_ROWS_SF10
    days, host = tpch.shipdate_column(rows, seed=42)
    columns = [DeviceColumn(host) for _ in range(3)]
    matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
    offsets = torch.zeros(host.n_chunks + 1, dtype=torch.int64, device=dev)
    counts = torch.zeros(host.n_chunks, dtype=torch.int32, device=dev)
    result = abi.ScanResult()
    result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS
    result.matches, result.capacity = matches.data_ptr(), rows
    result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()
    turn = [0]
    from hyrise_amd import storage
    import numpy as np
    rng = np.random.default_rng(44)
    widths = {"u16 value ids": (columns, 2)}
    if len(sys.argv) > 2:   # the other streaming instantiations; argv[2] = copies in rotation (1: a 60 - 240 MB column partly lives in the 256 MiB Infinity Cache)
        n_copies = int(sys.argv[2])
        for label, host_column, width in (("int32 values", storage.make_column(days, None, abi.ENC_UNENCODED), 4),
                                          ("u8 value ids", storage.make_column((days % 200).astype(np.int32), None, abi.ENC_DICTIONARY), 1),
                                          ("FrameOfReference u16", storage.make_column(np.sort(rng.integers(1, 60_000_000, rows).astype(np.int32)), None, abi.ENC_FRAME_OF_REFERENCE), 2)):
            widths[f"{label} x{n_copies}"] = ([DeviceColumn(host_column) for _ in range(n_copies)], width)
    for label, (copies, width) in widths.items():
        literal_sets = [("sel 0.43", abi.PRED_LESS_THAN, tpch.DAY_1995_01_01, None), ("sel 0.15", abi.PRED_BETWEEN_UPPER_EXCLUSIVE, tpch.DAY_1994_01_01, tpch.DAY_1995_01_01),
                        ("sel 0.99", abi.PRED_LESS_THAN_EQUALS, tpch.DAY_1998_09_02, None)]
        if label.startswith("u8 value ids"):
            literal_sets = [("sel 0.43", abi.PRED_LESS_THAN, 86, None), ("sel 0.15", abi.PRED_LESS_THAN, 30, None)]
        if label.startswith("FrameOfReference u16"):
            literal_sets = [("sel 0.33", abi.PRED_LESS_THAN, 20_000_000, None)]
        for name, condition, low, high in literal_sets:
            pred = make_predicate(condition, abi.TYPE_INT, low, high)

            def step():
                abi.check(lib.hy_table_scan(copies[turn[0] % len(copies)].handle, C.byref(pred), None, 0, C.byref(result)))
                turn[0] += 1
            for variant, env in (("write-back stores (default)", {}), ("nontemporal stores", {"HY_SCAN_NT_STORES": "1"}), ("write-back again", {}), ("nontemporal again", {"HY_SCAN_NT_STORES": "1"})):
                _switches = abi.switches(env)
                _switches.__enter__()
                dt, km = bench.timed_kernel(lib, torch, step, steps, 4, kind="scan")
                m = int(counts.sum().item())
                print(f"{label:24s} {name:9s} {variant:28s} {dt * 1e6:7.1f} us/scan  scan_slices {km * 1e3:6.1f} us  {(rows * width + m * 8) / (km * 1e-3) / 1e9:6.0f} GB/s on algorithmic bytes", flush=True)
                _switches.__exit__(None, None, None)


if __name__ == "__main__":
    main()
