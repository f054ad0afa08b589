#!/usr/bin/env python3
"""bench.py's join cases (SF10 sizes) one by one, for a kernel trace: shuffled probe side, shuffled unique build side, every build key four
times, a selective dimension build.  Usage: python tools/join_cases.py [case ...] [--steps N]; under rocprofv3 --kernel-trace --stats the
per-kernel table says where each case's time goes."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = 3
    if "--steps" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--steps") + 1])
        args = [a for a in args if a != str(steps)]
    cases = args or ["shuffled_probe", "shuffled_build", "duplicate_build_x4", "selective_dimension"]
    import numpy as np
    import torch
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    if "--no-hand-over" in sys.argv:   # pass 2 looks every probe key up again
        abi.check(lib.hy_set_option(abi.OPT_JOIN_HAND_OVER_RANKS, 0))
    dev = torch.device("cuda", 0)
    data = tpch.TpchData(10.0, 42, keys_only=True)
    n = data.n_lineitems
    rng = np.random.default_rng(7)

    def column(values, encoding):
        return DeviceColumn(storage.make_column(values, None, encoding))

    def measure(name, build, probe, capacity):
        run, r, keep = bench.device_join(lib, torch, dev, build, probe, capacity)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"{name:22s} {dt * 1e3:7.3f} ms/join  pairs {int(r.n_pairs)}  rank table {lib.hy_debug_join_used_rank_table()}  pkfk {lib.hy_debug_join_used_pkfk()}", flush=True)
        if os.environ.get("HY_JOIN_TRACE"):   # (a -DHY_DEBUG_SWITCHES build: per-tile phase stamps of pk_emit, wall clock at 100 MHz)
            lib.hy_debug_join_trace.restype = C.c_int
            stamps = np.zeros((1 << 15, 6), dtype=np.uint64)
            tiles = lib.hy_debug_join_trace(stamps.ctypes.data_as(C.c_void_p), C.c_uint32(1 << 15))
            t = stamps[:tiles].astype(np.float64) / 100.0
            t = t[t[:, 5] > 0]
            phases = ("evaluate", "reserve + rank", "prefix", "stage", "copy out")
            if len(t):
                print(f"   pk_emit trace over {len(t)} tiles: span {t[:, 5].max() - t[:, 0].min():.1f} us, per tile " +
                      ", ".join(f"{p} {np.median(t[:, i + 1] - t[:, i]):.2f}" for i, p in enumerate(phases)) + f", total {np.median(t[:, 5] - t[:, 0]):.2f} us (medians)")

    for case in cases:
        if case == "shuffled_probe":
            measure(case, column(data.o_orderkey, abi.ENC_UNENCODED), column(data.l_orderkey[rng.permutation(n)], abi.ENC_FRAME_OF_REFERENCE), n)
        elif case == "shuffled_build":
            measure(case, column(data.o_orderkey[rng.permutation(data.n_orders)], abi.ENC_UNENCODED), column(data.l_orderkey, abi.ENC_FRAME_OF_REFERENCE), n)
        elif case == "duplicate_build_x4":
            dup_keys = np.repeat(data.o_orderkey[:3_750_000], 4)[rng.permutation(15_000_000)]
            measure(case, column(dup_keys, abi.ENC_UNENCODED), column(data.l_orderkey[:16_000_000], abi.ENC_FRAME_OF_REFERENCE), 16_000_000 * 4 + 1024)
        elif case == "unencoded_probe":   # the headline's join with the foreign keys as a ValueSegment<int32>: clustered, but nothing in the layout says so
            measure(case, column(data.o_orderkey, abi.ENC_UNENCODED), column(data.l_orderkey, abi.ENC_UNENCODED), n)
        elif case == "selective_dimension":
            # an SSB-shaped star join: 1 000 of a dimension's 1 000 000 keys survive its filter, 180 M fact rows carry random foreign keys
            keys = np.sort(rng.choice(np.arange(1, 1_000_001, dtype=np.int32), 1000, replace=False))
            fact = rng.integers(1, 1_000_001, 180_000_000).astype(np.int32)
            measure(case, column(keys, abi.ENC_UNENCODED), column(fact, abi.ENC_UNENCODED), 1_000_000)
        else:
            raise SystemExit(f"unknown case {case}")


if __name__ == "__main__":
    main()
