#!/usr/bin/env python3
"""SSB SF30 Q2.1 or Q4.1 as ONE hy_star_join_aggregate call, N times (for rocprofv3 --kernel-trace --stats: the kernels of that plan alone).
Usage: python tools/ssb_star_time.py 2.1|4.1 [steps] [sf]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    query = sys.argv[1] if len(sys.argv) > 1 else "2.1"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sf = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    import torch
    from hyrise_amd import abi, ssb
    from hyrise_amd.distributed import HipExecutor
    from hyrise_amd.operators import star_join_aggregate
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    named = {k: v for k, v in os.environ.items() if k in abi._SWITCHES}   # A/B: the library's named switches from the environment
    abi.switches(named).__enter__()
    ex = HipExecutor(torch.device("cuda", 0))
    data = ssb.SsbData(scale_factor=sf, seed=7)
    columns = {name: ex.column(c) for name, c in data.host_columns().items()}
    dimensions, groupby, aggregates = ssb.star_plan(columns, query)
    result, joined = star_join_aggregate(dimensions, groupby, aggregates)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        result, joined = star_join_aggregate(dimensions, groupby, aggregates, result=result)
    torch.cuda.synchronize()
    lib.hy_debug_aggregate_path.restype = int
    print(f"Q{query} SF{sf:g}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per query, {joined} joined rows, {result.n_groups} groups, switches {named}, aggregate path {lib.hy_debug_aggregate_path()}")


if __name__ == "__main__":
    main()
