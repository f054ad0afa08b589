#!/usr/bin/env python3
"""TPC-H Q1 at SF10 through hy_scan_project_aggregate: the Q1-shaped kernel (fused_small_domain) beside fused_rows, results compared,
ms per query and the kernel's HIP-event time.  Usage: python tools/q1_fused_time.py [steps]   (not part of the product)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    import torch
    from hyrise_amd import abi, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    data = bench.sf10_tables()
    columns = {name: DeviceColumn(column) for name, column in tpch.q1_columns(data).items()}
    results = {}
    for label, env in (("fused_small_domain", {}), ("fused_rows", {"HY_FUSED_NO_SMALL_DOMAIN": "1"})):
        _switches = abi.switches(env)
        _switches.__enter__()
        results[label] = tpch.q1_fused(columns)
        which = lib.hy_debug_aggregate_small_domain()
        dt, km = bench.timed_kernel(lib, torch, lambda: tpch.q1_fused(columns), steps)
        print(f"{label:20s} kernel flag {which}  {dt * 1e3:7.3f} ms/query  kernel {km:7.3f} ms", flush=True)
        _switches.__exit__(None, None, None)
    a, b = results["fused_small_domain"], results["fused_rows"]
    assert a.n_groups == b.n_groups, (a.n_groups, b.n_groups)
    assert (a.row_ids[:a.n_groups] == b.row_ids[:b.n_groups]).all(), "group order / representative rows"
    worst = 0.0
    for i, name in enumerate(tpch.Q1_AGGREGATES):
        for x, y in zip(a.column(i), b.column(i)):
            worst = max(worst, abs(x - y) / max(1.0, abs(y)))
    print(f"groups {a.n_groups}, count_order {a.column(7)}, largest relative difference {worst:.3e}")
    assert worst <= 1e-9


if __name__ == "__main__":
    main()
