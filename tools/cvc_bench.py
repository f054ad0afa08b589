#!/usr/bin/env python3
"""ColumnVsColumn l_commitdate < l_receiptdate at SF10 (two dictionary columns, u16 value ids): the two-stream kernel (scan_two_columns) against
the generic instantiation (HY_OPT_SCAN_TWO_COLUMNS = 0), kernel time by HIP events.  Usage: python tools/cvc_bench.py [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    import numpy as np
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    dev = torch.device("cuda", 0)
    rows = tpch.LINEITEM_ROWS_SF10
    rng = np.random.default_rng(43)
    orderdate = rng.integers(0, tpch.LAST_ORDERDATE + 1, rows, dtype=np.int32)
    for encoding, name in ((abi.ENC_DICTIONARY, "dictionary u16 x dictionary u16"), (abi.ENC_FRAME_OF_REFERENCE, "FrameOfReference x FrameOfReference")):
        commit = DeviceColumn(storage.make_column((orderdate + rng.integers(30, 91, rows, dtype=np.int32)).astype(np.int32), None, encoding))
        receipt = DeviceColumn(storage.make_column((orderdate + rng.integers(2, 152, rows, dtype=np.int32)).astype(np.int32), None, encoding))
        matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
        offsets = torch.zeros(commit.n_chunks + 1, dtype=torch.int64, device=dev)
        counts = torch.zeros(commit.n_chunks, dtype=torch.int32, device=dev)
        result = abi.ScanResult()
        result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS
        result.matches, result.capacity = matches.data_ptr(), rows
        result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()

        def run():
            abi.check(lib.hy_table_scan_columns(commit.handle, receipt.handle, abi.PRED_LESS_THAN, C.byref(result)))

        for two in (1, 0, 1):
            with abi.option(abi.OPT_SCAN_TWO_COLUMNS, two):
                dt, km = bench.timed_kernel(lib, torch, run, steps, 1)
            m = int(counts.sum().item())
            width = 2 if encoding == abi.ENC_DICTIONARY else commit.host.segments[0].width
            print(f"{name:40s} two_columns={two}  {dt * 1e3:7.3f} ms/scan  kernel {km * 1e3:7.1f} us  matches {m}  {(rows * 2 * width + m * 8) / (km * 1e-3) / 1e9:7.1f} GB/s", flush=True)
        del commit, receipt


if __name__ == "__main__":
    main()
