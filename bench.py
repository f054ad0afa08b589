#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload at every N: config 2 of BASELINE.json -- TableScan ColumnVsValue on a TPC-H SF10 lineitem `l_shipdate`
column (59 986 052 rows, 916 chunks of 65 535, DictionarySegment<int32> + FixedWidthInteger u16 attribute vectors),
predicate `l_shipdate < 1995-01-01` (the reference's own micro-benchmark predicate,
src/benchmark/tpch_data_micro_benchmark.cpp:65-67).  One step = one hy_table_scan over the whole column with the
column resident in HBM and the PosLists written to HBM.  With N GPUs every rank scans its own SF10-shaped shard of an
N x SF10 table (chunks shard naturally, no data-path collective: SURVEY.md section 8(e)) -> weak scaling.

Prints ONE JSON line on rank 0: rows/s over the whole job, plus `roofline` (scan_slices kernel: algorithmic bytes /
HIP-event duration vs. the 8 TB/s HBM peak) and `cpu_baseline` (the CPU restatement of the Hyrise operator on the
host cores, rank 0, N=1 only).  At N=1 the line also carries `join` and `aggregate` objects: the SF10 orders x lineitem
JoinHash and the TPC-H Q1-core AggregateHash on the same GPU (configs 3 and 4 of BASELINE.json), outside the timed region.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_EVERY = 4               # every 4th scan of the timed region carries the HIP event pair (a timed launch costs ~7 us of stream time)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=0, help="override row count (debug); 0 = SF10 lineitem")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cases", action="store_true", help="also time the Q1/Q6/point predicates (extra JSON field)")
    ap.add_argument("--no-join", action="store_true", help="skip the JoinHash orders x lineitem leg (extra JSON field)")
    ap.add_argument("--no-aggregate", action="store_true", help="skip the AggregateHash Q1-core leg (extra JSON field)")
    return ap.parse_args()


def cpu_baseline(host_column, predicate, rows, budget_s=12.0):
    """CPU restatement of the Hyrise TableScan (oracle/, kind 'port'), all host cores, one job per chunk range
    (table_scan.cpp:223-229).  The ONLY place bench.py touches the oracle: a reported baseline, never the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support
    from hyrise_amd.operators import HostScanResult
    cores = os.cpu_count() or 1
    # column descriptors and the PosList buffers are set up once (a Hyrise operator writes into pooled memory, it does not
    # page-fault 480 MB of fresh output per scan); the first call below touches every page
    column = support.OracleCol(host_column)
    result = HostScanResult(host_column.n_chunks, host_column.rows, 0)
    scan = support.oracle().hyo_table_scan

    def run():
        status = scan(C.byref(column.c), C.byref(predicate), C.byref(result.c), cores)
        if status != 0:
            raise SystemExit(f"oracle scan failed with {status}")

    run()
    run()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 25):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    times.sort()
    median = times[len(times) // 2]
    return {"value": rows / median, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"full {rows}-row l_shipdate column, same predicate, median of {len(times)} runs "
                      f"({median * 1e3:.1f} ms each), CPU restatement of Hyrise's TableScan (not Hyrise itself)"}


def committed_traffic():
    """HBM bytes per scan_slices launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled as
    the gfx950 guide prescribes, + WRITE_SIZE; collected by tools/profile_scan.sh on the same command).  None when no
    profile of this round is committed."""
    path = os.path.join(ROOT, "profiles", "scan_pmc.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def join_leg(lib, torch, dev, steps):
    """Config 3 of BASELINE.json: JoinHash(orders, lineitem) on the order key, SF10 -- o_orderkey unencoded int32 (build),
    l_orderkey FrameOfReference + u16 offsets (probe); PosList pairs written to HBM.  Reported beside the scan."""
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.storage import DeviceColumn
    data = tpch.TpchData(scale_factor=10.0, seed=42)
    orders = DeviceColumn(storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED))
    lineitem = DeviceColumn(storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE))
    n = data.n_lineitems
    left = torch.empty((n, 2), dtype=torch.int32, device=dev)
    right = torch.empty((n, 2), dtype=torch.int32, device=dev)
    slice_offsets = torch.zeros(4096, dtype=torch.int64, device=dev)
    r = abi.JoinResult()
    r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
    r.left_pos, r.right_pos, r.capacity = left.data_ptr(), right.data_ptr(), n
    r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 4000
    steps = max(3, min(steps, 10))
    for _ in range(2):
        abi.check(lib.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(r)))
    abi.check(lib.hy_set_profiling(1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        abi.check(lib.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(r)))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    km, ln = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(km), C.byref(ln)))
    abi.check(lib.hy_set_profiling(0))
    algorithmic = data.n_orders * 4 + n * 2 + int(r.n_pairs) * 16      # build keys + probe keys + 16 B/pair
    return {"workload": "configs[2]: JoinHash orders x lineitem on the order key, SF10, Inner", "rows_per_s": (data.n_orders + n) / dt,
            "ms_per_join": dt * 1e3, "pairs": int(r.n_pairs), "radix_bits": int(r.radix_bits), "output_pos_lists": int(r.n_slices),
            "algorithmic_bytes": algorithmic, "achieved_GBps_whole_join": algorithmic / dt / 1e9,
            "probe_emit_kernel_ms": km.value / max(1, ln.value)}


def aggregate_leg(lib, torch, steps):
    """Config 4 of BASELINE.json on one GPU: AggregateHash, TPC-H Q1 core -- GROUP BY l_returnflag, l_linestatus (dictionary
    segments, u8 attribute vectors) with SUM / AVG over l_quantity, l_extendedprice, l_discount (float value segments) and
    COUNT(*), SF10 lineitem.  Reported beside the scan."""
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.operators import aggregate_hash
    from hyrise_amd.storage import DeviceColumn
    data = tpch.TpchData(scale_factor=10.0, seed=42)
    n = data.n_lineitems
    flag = DeviceColumn(storage.make_column(data.l_returnflag, None, abi.ENC_DICTIONARY))
    status = DeviceColumn(storage.make_column(data.l_linestatus, None, abi.ENC_DICTIONARY))
    quantity = DeviceColumn(storage.make_column(data.l_quantity, None, abi.ENC_UNENCODED))
    price = DeviceColumn(storage.make_column(data.l_extendedprice, None, abi.ENC_UNENCODED))
    discount = DeviceColumn(storage.make_column(data.l_discount, None, abi.ENC_UNENCODED))
    aggregates = [(abi.AGG_SUM, quantity), (abi.AGG_SUM, price), (abi.AGG_AVG, quantity), (abi.AGG_AVG, price), (abi.AGG_AVG, discount),
                  (abi.AGG_COUNT, None)]
    steps = max(3, min(steps, 10))
    for _ in range(2):
        result = aggregate_hash([flag, status], aggregates, group_capacity=64)
    abi.check(lib.hy_set_profiling(1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        result = aggregate_hash([flag, status], aggregates, group_capacity=64)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    km, ln = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(km), C.byref(ln)))
    abi.check(lib.hy_set_profiling(0))
    algorithmic = n * (1 + 1 + 4 + 4 + 4)   # two u8 attribute vectors + three float columns, each read once
    return {"workload": "configs[3] on one GPU: AggregateHash Q1 core, GROUP BY l_returnflag, l_linestatus, SF10 lineitem",
            "rows_per_s": n / dt, "ms_per_aggregate": dt * 1e3, "groups": int(result.n_groups), "algorithmic_bytes": algorithmic,
            "achieved_GBps_whole_operator": algorithmic / dt / 1e9, "aggregate_rows_kernel_ms": km.value / max(1, ln.value)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    from hyrise_amd import abi, tpch
    from hyrise_amd.operators import make_predicate
    from hyrise_amd.storage import DeviceColumn

    # HY_BENCH_SHARE_GPU=1 (debug): several ranks on ONE GPU with the gloo backend, to exercise the N > 1 control flow on
    # a single-GPU box; the real thing is one rank per GPU over RCCL.
    share_gpu = bool(os.environ.get("HY_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    lib = abi.load_library()          # raises if the HIP library is missing: no fallback
    abi.check(lib.hy_init(local_rank))
    stream = torch.cuda.current_stream()
    abi.check(lib.hy_set_stream(C.c_void_p(stream.cuda_stream)))

    # ---- data: this rank's SF10-shaped shard (seed differs per rank), encoded like Hyrise, uploaded once -------------
    rows = args.rows or tpch.LINEITEM_ROWS_SF10
    days, host_column = tpch.shipdate_column(rows, seed=42 + rank)
    column = DeviceColumn(host_column)
    n_chunks = host_column.n_chunks
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)

    dev = torch.device("cuda", local_rank)
    matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
    offsets = torch.zeros(n_chunks + 1, dtype=torch.int64, device=dev)
    result = abi.ScanResult()
    result.mem = abi.MEM_DEVICE
    result.matches, result.capacity = matches.data_ptr(), rows
    counts = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    result.flags = abi.SCAN_CHUNK_REGIONS  # chunk c's PosList at matches[offsets[c] : offsets[c] + counts[c]]
    result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()

    def step(pred=predicate):
        abi.check(lib.hy_table_scan(column.handle, C.byref(pred), None, 0, C.byref(result)))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    abi.check(lib.hy_set_profiling(0 if os.environ.get("HY_BENCH_NO_EVENTS") else PROFILE_EVERY))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(kernel_ms), C.byref(launches)))
    abi.check(lib.hy_set_profiling(0))

    n_matches = int(counts.sum().item())
    expected = int((days < tpch.DAY_1995_01_01).sum())
    if n_matches != expected:
        raise SystemExit(f"rank {rank}: scan produced {n_matches} matches, numpy says {expected}")

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    total_rows = rows * world
    value = total_rows / (elapsed / args.steps)

    # roofline of the dominant kernel (scan_slices): algorithmic bytes per launch = attribute vector + RowIDs written
    width = host_column.segments[0].width
    algorithmic_bytes = rows * width + n_matches * 8
    kernel_s = (kernel_ms.value / max(1, launches.value)) * 1e-3
    achieved = algorithmic_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0

    extra_cases = None
    if args.cases and rank == 0:
        extra_cases = {}
        for name, pred in (("q1_le_1998-09-02", make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, tpch.DAY_1998_09_02)),
                           ("q6_between_1994", make_predicate(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, tpch.DAY_1994_01_01, tpch.DAY_1995_01_01)),
                           ("point_eq_1995-06-17", make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, tpch.CURRENT_DATE))):
            for _ in range(3):
                step(pred)
            abi.check(lib.hy_set_profiling(PROFILE_EVERY))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(pred)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.steps
            km, ln = C.c_float(0), C.c_uint32(0)
            abi.check(lib.hy_profile_read(C.byref(km), C.byref(ln)))
            abi.check(lib.hy_set_profiling(0))
            m = int(counts.sum().item())
            bytes_case = rows * width + m * 8
            extra_cases[name] = {"rows_per_s": rows / dt, "ms_per_step": dt * 1e3, "matches": m,
                                 "kernel_ms": km.value / max(1, ln.value),
                                 "kernel_GBps": bytes_case / (km.value / max(1, ln.value) * 1e-3) / 1e9}

    join_info = None
    if rank == 0 and world == 1 and not args.no_join and not args.rows:   # (single-GPU legs: the N > 1 runs time the sharded scan only)
        join_info = join_leg(lib, torch, dev, args.steps)

    aggregate_info = None
    if rank == 0 and world == 1 and not args.no_aggregate and not args.rows:
        aggregate_info = aggregate_leg(lib, torch, args.steps)

    if rank == 0:
        line = {
            "metric": "rows/sec TableScan (ColumnVsValue, l_shipdate < 1995-01-01) on TPC-H SF10 lineitem",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "configs[1]: TableScan ColumnVsValue, SF10 lineitem l_shipdate, DictionarySegment<int32> "
                                   "+ u16 attribute vectors, 916 chunks x 65535 rows, column and PosLists resident in HBM",
                       "rows_per_gpu": rows, "chunks_per_gpu": n_chunks, "selectivity": n_matches / rows,
                       "parallelism": f"chunk-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "scan_slices",
                         "algorithmic_bytes_per_launch": algorithmic_bytes,
                         "kernel_ms": kernel_s * 1e3, "launches_timed": int(launches.value)},
        }
        if extra_cases:
            line["cases"] = extra_cases
        if join_info:
            line["join"] = join_info
        if aggregate_info:
            line["aggregate"] = aggregate_info
        line["roofline"]["traffic"] = committed_traffic()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_column, predicate, rows)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
