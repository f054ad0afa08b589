#!/usr/bin/env python3
"""Results in HOST memory (what the INTEGRATION.md adapter asks for): the SF10 join's two PosLists (0.96 GB) and the SF10 scan's PosLists (0.2 GB),
into buffers whose pages exist and into fresh ones (first touch: the page faults are the caller's).  Usage: python tools/host_result_bench.py
Round 4: 17.5 ms / 3.9 ms = 55 / 53 GB/s of PCIe 5 x16 into touched buffers -- a pinned-tile pipeline with helper threads (tried) moved 50 GB/s:
the runtime's own staging of pageable copies is the better one; the 56 ms of round 3 were first-touch page faults of a fresh numpy buffer."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.operators import HostJoinResult, HostScanResult, make_predicate
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    data = tpch.TpchData(10.0, 42, keys_only=True)
    orders = DeviceColumn(storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED))
    lineitem = DeviceColumn(storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE))
    n = data.n_lineitems
    days, host_column = tpch.shipdate_column(n, seed=42)
    shipdate = DeviceColumn(host_column)
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)
    join_result = HostJoinResult(n, n // 131070 + 1000)
    join_result.left[:] = 1
    join_result.right[:] = 1          # (the pages exist: a Hyrise operator writes into pooled memory)
    scan_result = HostScanResult(host_column.n_chunks, n, 0)
    scan_result.matches[:] = 1
    for buffers in ("touched", "fresh", "touched"):
        if buffers == "fresh":
            join_result = HostJoinResult(n, n // 131070 + 1000)
            scan_result = HostScanResult(host_column.n_chunks, n, 0)
        for _ in range(1 if buffers == "fresh" else 2):
            t0 = time.perf_counter()
            abi.check(lib.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(join_result.c)))
            dt_join = time.perf_counter() - t0
            t0 = time.perf_counter()
            abi.check(lib.hy_table_scan(shipdate.handle, C.byref(predicate), None, 0, C.byref(scan_result.c)))
            dt_scan = time.perf_counter() - t0
        pairs, matches = int(join_result.c.n_pairs), int(scan_result.c.total_matches)
        print(f"{buffers:8s} buffers: join {dt_join * 1e3:7.2f} ms ({pairs * 16 / dt_join / 1e9:5.1f} GB/s over the link)   scan {dt_scan * 1e3:6.2f} ms ({matches * 8 / dt_scan / 1e9:5.1f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
