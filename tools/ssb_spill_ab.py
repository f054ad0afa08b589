#!/usr/bin/env python3
"""SSB Q2.1 / Q4.1 at SF30 under different give-up thresholds of aggregate_rows (HY_AGG_SPILL_SHIFT: rows >> shift spilled rows before the
partitioned path takes over).  One process, one data set.  Usage: python tools/ssb_spill_ab.py [sf]   (not part of the product)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    sf = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    import torch
    from hyrise_amd import abi, ssb
    from hyrise_amd.distributed import HipExecutor, aggregate_groups
    device = torch.device("cuda", 0)
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    ex = HipExecutor(device)
    data = ssb.SsbData(scale_factor=sf, seed=7)
    columns = {name: ex.column(column) for name, column in data.host_columns().items()}
    reference = {}
    for shift in ("4", "3", "2", "4"):
        abi.check(lib.hy_set_option(abi.OPT_AGG_SPILL_SHIFT, int(shift)))
        for query in ("2.1", "4.1"):
            def once():
                groupby, aggregates, joined = ssb.run_query(ex, columns, query)
                return aggregate_groups(ex, groupby, aggregates)
            groups = once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                once()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            rows = sorted(ssb.result_rows(groups))
            same = reference.setdefault(query, rows) == rows
            print(f"shift {shift}  Q{query}  {ms:7.3f} ms  {len(rows)} groups  partition path {lib.hy_debug_aggregate_path()}  same groups {same}", flush=True)


if __name__ == "__main__":
    main()
