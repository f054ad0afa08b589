#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X hot path (BASELINE.json metric: rows/sec TableScan+JoinHash, TPC-H SF10).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = ONE TableScan + ONE JoinHash on the same GPU, inputs and outputs resident in HBM:
  * TableScan ColumnVsValue (config 2): TPC-H SF10 lineitem `l_shipdate` (59 986 052 rows, 916 chunks of 65 535, DictionarySegment<int32>
    + FixedWidthInteger u16 attribute vectors), predicate `l_shipdate < 1995-01-01` (the reference's own micro-benchmark predicate,
    src/benchmark/tpch_data_micro_benchmark.cpp:65-67), PosLists written to HBM; the steps rotate over three copies of the column;
  * JoinHash (config 3): orders x lineitem on the order key, Inner (o_orderkey ValueSegment<int32> = build side, 15 000 000 rows;
    l_orderkey FrameOfReference + u16 offsets = probe side, 59 986 052 rows), both PosLists (0.96 GB) written to HBM; three copies of
    both key columns in rotation too.  The join is called with HY_JOIN_ASYNC: it returns when its kernels are queued, its pair count
    stays in device memory (hy_join_status) and is read and checked once, after the timed region (hy_join_hash_finish).
Each step moves > 1.4 GB through the memory-side cache (256 MiB): nothing it reads is still there when it comes round again.
`value` = (scanned rows + build rows + probe rows) / step time.  With N GPUs every rank runs the step on its own SF10-shaped shard
of an N x SF10 database (chunks shard naturally, orders and lineitem co-partitioned by order key range: no data-path collective,
SURVEY.md section 8(e)) -> weak scaling; the line then also carries the strong-scaling scan (ONE SF10 table split over the ranks), the
sharded AggregateHash (per-rank partials + one all-reduce over RCCL) and the sharded JoinHash (broadcast-build by all-gather, and
hash repartition by all-to-all) -- `multi_gpu` object, hyrise_amd/distributed.py.

Prints ONE JSON line on rank 0: `roofline` = the step's algorithmic bytes (scan 2 B/row + 8 B/match, join build
keys + probe keys + 16 B/pair) over the step time vs. the 8 TB/s HBM peak, with the HIP-event durations of the step's kernels
(`kernels`: scan_slices, pk_emit, pk_count, rank_table_fill_waves -- events taken INSIDE the timed region, every 4th step) and
`dominant_kernel` = pk_emit; `cpu_baseline` = the CPU restatement of the two Hyrise operators on the host cores (rank 0, N = 1
only).  At N = 1 the line also carries `scan` (config 2 alone), `join` (config 3 alone, + Semi legs + cases) and `aggregate`
(config 4), each with its own roofline and cpu_baseline, `cases` (the other predicates / encodings SURVEY.md 8(d) lists), `q6`
(configs[0]'s query as a device-resident operator chain, and `q6.fused`), `q1` and `ssb` (config 5).
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
COLUMN_COPIES = 3               # > 256 MiB of column data in rotation


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=0, help="override row count (debug); 0 = SF10 lineitem")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cases", action="store_true", help="skip the extra scan / join cases (N = 1)")
    ap.add_argument("--no-join", action="store_true", help="skip the JoinHash orders x lineitem leg")
    ap.add_argument("--no-aggregate", action="store_true", help="skip the AggregateHash Q1-core leg")
    ap.add_argument("--no-multi", action="store_true", help="N > 1: skip the strong-scaling / aggregate / join legs")
    ap.add_argument("--no-ssb", action="store_true", help="skip the SSB SF30 star-join leg (config 5)")
    ap.add_argument("--switch", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: one of the library's named switches (hyrise_amd/abi.py _SWITCHES, e.g. HY_JOIN_NO_HINT=1) for the whole run")
    ap.add_argument("--placements", type=int, default=32, help="result-buffer placements the join's output pool is calibrated over before the timed region (1 = take the first)")
    ap.add_argument("--headline-only", action="store_true", help="only the timed TableScan + JoinHash step (no legs, no CPU baselines)")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"),
                    help="file the full result object goes to (every leg, every case, prose); stdout gets the compact line only")
    return ap.parse_args(argv)


def oracle_support():
    """tests/support.py: the ctypes side of the CPU oracle.  The ONLY use of oracle/ in this file is the cpu_baseline
    objects: reported baselines, never the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support
    return support


def median_time(run, budget_s, min_runs=3, max_runs=25):
    run()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < min_runs or (time.perf_counter() < t_end and len(times) < max_runs):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def cpu_baseline_scan(host_column, predicate, rows, budget_s=10.0):
    """CPU restatement of the Hyrise TableScan (oracle/, kind 'port'), all host cores, one job per chunk range
    (table_scan.cpp:223-229)."""
    support = oracle_support()
    from hyrise_amd.operators import HostScanResult
    cores = os.cpu_count() or 1
    # column descriptors and the PosList buffers are set up once (a Hyrise operator writes into pooled memory, it does not
    # page-fault 480 MB of fresh output per scan); the first call touches every page
    column = support.OracleCol(host_column)
    result = HostScanResult(host_column.n_chunks, host_column.rows, 0)
    scan = support.oracle().hyo_table_scan

    def run():
        if scan(C.byref(column.c), C.byref(predicate), C.byref(result.c), cores) != 0:
            raise SystemExit("oracle scan failed")

    run()
    median, n = median_time(run, budget_s)
    return {"value": rows / median, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"full {rows}-row l_shipdate column, same predicate, median of {n} runs ({median * 1e3:.1f} ms each), "
                      "CPU restatement of Hyrise's TableScan (not Hyrise itself)"}


def cpu_baseline_step(host_column, predicate, rows, orders_host, lineitem_host):
    """The headline's CPU baseline: the oracle's TableScan and the oracle's JoinHash, one after the other like a step, all host cores."""
    scan = cpu_baseline_scan(host_column, predicate, rows, budget_s=6.0)
    join_rows = orders_host.rows + lineitem_host.rows
    join = cpu_baseline_join(orders_host, lineitem_host, join_rows, budget_s=10.0)
    seconds = rows / scan["value"] + join_rows / join["value"]
    return {"value": (rows + join_rows) / seconds, "unit": "rows/s", "cores": scan["cores"], "kind": "port",
            "sample": "one full step on the host: " + scan["sample"] + " | then " + join["sample"],
            "scan": scan, "join": join}


def cpu_baseline_join(orders, lineitem, rows, budget_s=12.0):
    """The oracle's JoinHash (materialise -> radix-partition -> build -> probe with the reference's radix_bits, one job per
    chunk / partition: join_hash.cpp:270-572), all host cores."""
    support = oracle_support()
    from hyrise_amd import abi
    cores = os.cpu_count() or 1

    def run():
        support.oracle_join(orders, lineitem, abi.JOIN_INNER, None, threads=cores, capacity=lineitem.rows + 1024)

    median, n = median_time(run, budget_s, min_runs=2, max_runs=5)
    return {"value": rows / median, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"full SF10 orders x lineitem Inner join, median of {n} runs ({median * 1e3:.0f} ms each, fresh output buffers each run), "
                      "CPU restatement of Hyrise's JoinHash"}


def cpu_baseline_aggregate(groupby, aggregates, rows, budget_s=12.0):
    """The oracle's AggregateHash: sequential over chunks and rows like the reference (aggregate_hash.cpp:1016-1176), one core --
    and, beside it, the per-thread-partial variant SURVEY.md 8(d) asks for: every host thread aggregates a contiguous chunk range
    with the same code, the (few) partial groups are added up."""
    support = oracle_support()
    from concurrent.futures import ThreadPoolExecutor
    from hyrise_amd.distributed import shard_column

    def run():
        support.oracle_aggregate(groupby, aggregates, group_capacity=64)

    median, n = median_time(run, budget_s, min_runs=1, max_runs=3)
    threads = max(1, min(os.cpu_count() or 1, 64, groupby[0].n_chunks))
    shards = [([shard_column(c, threads, t)[0] for c in groupby], [(f, shard_column(c, threads, t)[0] if c is not None else None) for f, c in aggregates])
              for t in range(threads)]

    def run_partials():
        with ThreadPoolExecutor(threads) as pool:   # (ctypes releases the GIL for the duration of the C call)
            partials = list(pool.map(lambda shard: support.oracle_aggregate(shard[0], shard[1], group_capacity=64), shards))
        return sum(p.n_groups for p in partials)

    median_partials, n_partials = median_time(run_partials, budget_s / 2, min_runs=2, max_runs=5)
    return {"value": rows / median, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"full SF10 lineitem Q1 core (string keys as key names, dictionary-encoded float measures), median of {n} runs "
                      f"({median * 1e3:.0f} ms each), sequential like the reference's AggregateHash",
            "per_thread_partials": {"value": rows / median_partials, "unit": "rows/s", "cores": threads,
                                    "sample": f"the same aggregate as {threads} chunk-range partials on {threads} threads, median of {n_partials} runs "
                                              f"({median_partials * 1e3:.0f} ms each; merging the partial groups is not timed: a handful of additions)"}}


def cpu_baseline_ssb(threads=None, runs=3):
    """Config 5 on the host cores: the plans of hyrise_amd/ssb.py on the CPU restatement of the operators (tests/oracle_executor.py) over
    the FULL SF30 tables, scans and joins on every core (chunk ranges / radix partitions per thread; the aggregate is sequential like the
    reference's AggregateHash), median of `runs` runs per query."""
    oracle_support()
    from oracle_executor import OracleExecutor
    from hyrise_amd import ssb
    from hyrise_amd.distributed import aggregate_groups
    threads = threads or max(1, os.cpu_count() or 1)
    data = ssb.SsbData(scale_factor=30.0, seed=7)
    columns = data.host_columns()
    ex = OracleExecutor(threads=threads)
    out = {}
    for query in ("2.1", "4.1"):
        times = []
        for _ in range(runs):
            t0 = time.perf_counter()
            groupby, aggregates, joined = ssb.run_query(ex, columns, query)
            groups = aggregate_groups(ex, groupby, aggregates)
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        out[f"q{query}"] = {"_rows": ssb.result_rows(groups), "_joined": joined,
                            "value": data.n_lineorder / dt, "unit": "lineorder rows/s", "cores": threads, "kind": "port",
                            "sample": f"full SF30 ({data.n_lineorder} lineorder rows), median of {runs} runs ({dt:.2f} s each): scan -> JoinHash per dimension -> AggregateHash on the "
                                      f"CPU restatement of Hyrise's operators, scans and joins on {threads} threads, numpy gathers between them; {joined} joined rows, {len(groups)} groups"}
    return out


def committed_traffic(kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes over THIS script that are committed under profiles/
    (tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE in separate runs of `python bench.py`, FETCH_SIZE doubled as the gfx950
    guide prescribes).  None when the committed profile does not list the kernel (rocprofv3 cannot run inside this process: the
    counters are not measured live -- `traffic_source` in the line says which file and commit the figure is from)."""
    path = pmc_file()
    if not path:
        return None
    try:
        with open(path) as fh:
            entry = json.load(fh).get("kernels", {}).get(kernel)
        return entry.get("hbm_bytes_per_launch") if entry else None
    except (OSError, ValueError):
        return None


def pmc_file():
    """The newest committed PMC summary of this script (profiles/rNN_bench_pmc.json, tools/collect_profiles.sh)."""
    for round_ in (6, 5, 4, 3):
        path = os.path.join(ROOT, "profiles", f"r{round_:02d}_bench_pmc.json")
        if os.path.exists(path):
            return path
    return None


def traffic_source():
    path = pmc_file()
    if not path:
        return None
    try:
        with open(path) as fh:
            return "profiles/" + os.path.basename(path) + ", " + str(json.load(fh).get("collected", ""))
    except (OSError, ValueError):
        return None


PCIE_PEAK_GBS = 63.0           # PCIe 5.0 x16, one direction: 32 GT/s x 16 lanes x 128/130


def pcie_roofline(what, payload_bytes, seconds):
    """A result (or an upload) that crosses the host link: payload bytes over the call's wall time against the link's peak."""
    achieved = payload_bytes / seconds / 1e9 if seconds > 0 else 0.0
    return {"bound": "pcie", "achieved": achieved, "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": achieved / PCIE_PEAK_GBS, "traffic": None, "kernel": what,
            "algorithmic_bytes_per_launch": payload_bytes, "kernel_ms": seconds * 1e3}


def timed_upload(name, host_column, uploads):
    """DeviceColumn(host_column) with its H2D time on record: `uploads[name]` = bytes, ms, GB/s of hy_column_create (pageable host buffers ->
    one arena in HBM; resident-column runs -- everything else in this file -- start behind it)."""
    from hyrise_amd.storage import DeviceColumn
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    column = DeviceColumn(host_column)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    payload = sum(sum(int(a.nbytes) for a in (seg.data, seg.aux, seg.nulls) if a is not None and hasattr(a, "nbytes")) for seg in host_column.segments)
    if name not in uploads and payload:
        uploads[name] = dict(pcie_roofline(f"hy_column_create({name}): host segments -> one arena in HBM, through 32 MiB windows of pinned memory", payload, dt), bytes=payload, ms=dt * 1e3)
    return column


def write_details(line, path):
    """The full result object (every leg, case and note) as indented JSON: `path`, and gpurun_out/ beside it when that exists (it travels back)."""
    text = json.dumps(line, indent=1)
    default = os.path.join(ROOT, "bench_details.json")   # (the default file is mirrored into gpurun_out/, which travels back from the GPU box; a named file is not)
    targets = [path] + ([os.path.join(ROOT, "gpurun_out", "bench_details.json")] if path == default and os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])
    for target in targets:
        try:
            with open(target, "w") as fh:
                fh.write(text + "\n")
        except OSError as error:   # (a read-only checkout: the compact line still goes out)
            sys.stderr.write(f"bench.py: could not write {target}: {error}\n")


def _short(x, digits=5):
    """Numbers to `digits` significant figures (the compact line), everything else untouched."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def compact_roofline(r, kernel=None):
    """bound / achieved / peak / unit / frac / traffic / kernel / kernel_ms / bytes -- no prose."""
    if not r:
        return None
    return {"bound": r["bound"], "achieved": _short(r["achieved"]), "peak": r["peak"], "unit": r["unit"], "frac": _short(r["frac"], 4),
            "traffic": _short(float(r["traffic"])) if r.get("traffic") else None, "kernel": kernel or r.get("kernel"),
            "kernel_ms": _short(r.get("kernel_ms")), "bytes": r.get("algorithmic_bytes_per_launch")}


def compact_cpu_baseline(c, sample):
    return {"value": _short(c["value"]), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": sample}


def compact_line(line, details_path=None):
    """The ONE line the driver parses (a few KB): the contract's keys, `roofline` with the step's kernels, `cpu_baseline`, and one number per
    other leg (`legs`); everything else -- workload prose, cases, notes -- is in the --details file."""
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["metric"] = "rows/sec TableScan+JoinHash, TPC-H SF10 lineitem"
    out["value"], out["ms_per_step"] = _short(line["value"], 6), _short(line["ms_per_step"], 6)
    cfg = line["config"]
    out["config"] = {"workload": "configs[1]+configs[2] per step: TableScan l_shipdate<1995-01-01 (Dictionary int32, u16 ids) + JoinHash orders x lineitem Inner "
                                 "(o_orderkey int32 build, l_orderkey FoR u16 probe), SF10, in HBM, 3 column copies in rotation, HY_JOIN_ASYNC, "
                                 "build-key hint from an earlier join",
                     "rows_per_step_per_gpu": cfg["rows_per_step_per_gpu"], "scan_rows": cfg["scan_rows"], "build_rows": cfg["build_rows"], "probe_rows": cfg["probe_rows"],
                     "chunks_per_gpu": cfg["chunks_per_gpu"], "scan_selectivity": _short(cfg["scan_selectivity"], 4), "parallelism": cfg["parallelism"]}   # (join_pairs = probe_rows: in the details)
    if cfg.get("output_placement"):
        out["config"]["output_placement"] = cfg["output_placement"]
        out["config"]["workload"] += (", PosLists from the library's pool (hy_result_pool_calibrate over %d placements before the timed region); "
                                      "ms_per_step_median_placement: the step in the median candidate") % cfg["output_placement"]["candidates"]
        placement = cfg["output_placement"]
        times = sorted(placement.get("join_ms_per_candidate") or [0.0])   # (every candidate's time: bench_details.json)
        out["config"]["output_placement"] = {"candidates": placement["candidates"], "chosen": placement.get("chosen"), "join_ms_fastest": times[0],
                                             "join_ms_median": times[len(times) // 2], "join_ms_slowest": times[-1]}
    r = line["roofline"]
    roof = compact_roofline(r, "TableScan+JoinHash step, host-timed")
    roof["dominant_kernel"] = compact_roofline(r.get("dominant_kernel"))
    roof["kernels"] = {name: compact_roofline(k) for name, k in r.get("kernels", {}).items()}
    roof["launches_timed"] = r.get("launches_timed")
    roof["traffic_source"] = (r.get("traffic_source") or "").split(",")[0] or None
    out["roofline"] = roof
    if "cpu_baseline" in line:
        c = line["cpu_baseline"]
        out["cpu_baseline"] = compact_cpu_baseline(c, "full SF10 step on the host: oracle scan then oracle join (medians), all threads")
        out["cpu_baseline"]["scan"] = _short(c["scan"]["value"])
        out["cpu_baseline"]["join"] = _short(c["join"]["value"])
    legs = {}
    if "scan" in line:
        legs["scan"] = {"ms": _short(line["scan"]["ms_per_scan"]), "frac": _short(line["scan"]["roofline"]["dominant_kernel"]["frac"], 4)}
    join = line.get("join", {})
    if "ms_per_join" in join:
        legs["join"] = {"ms": _short(join["ms_per_join"]), "frac": _short(join["roofline"]["frac"], 4)}
        for name, case in join.get("cases", {}).items():
            legs["join"][name + "_ms"] = _short(case["ms_per_join"])
        if "cpu_baseline" in join:
            legs["join"]["cpu_rows_per_s"] = _short(join["cpu_baseline"]["value"])
    if "first_join_no_hint_ms" in join:
        out["join"] = {"first_join_no_hint_ms": _short(join["first_join_no_hint_ms"])}
    if "aggregate" in line:
        a = line["aggregate"]
        k = a["roofline"]["dominant_kernel"]
        legs["aggregate"] = {"ms": _short(a["ms_per_aggregate"]), "kernel": k["kernel"], "kernel_ms": _short(k["kernel_ms"]), "frac": _short(k["frac"], 4)}
        if "cpu_baseline" in a:
            legs["aggregate"]["cpu_rows_per_s"] = _short(a["cpu_baseline"]["value"])
    if "cases" in line and "column_vs_column_commit_lt_receipt" in line["cases"]:
        legs["column_vs_column_ms"] = _short(line["cases"]["column_vs_column_commit_lt_receipt"]["kernel_ms"])
    if "q6" in line:
        legs["q6"] = {"chain_ms": _short(line["q6"]["ms_per_query"]), "fused_ms": _short(line["q6"]["fused"]["ms_per_query"]), "fused_frac": _short(line["q6"]["fused"]["roofline"]["frac"], 4)}
    if "q1" in line:
        legs["q1"] = {"chain_ms": _short(line["q1"]["chain_ms_per_query"]), "fused_ms": _short(line["q1"]["fused"]["ms_per_query"]), "fused_frac": _short(line["q1"]["fused"]["roofline"]["frac"], 4)}
    if "ssb" in line:
        s = line["ssb"]
        legs["ssb_sf30"] = {q: {"ms": _short(s[q]["ms"]), "frac": _short(s[q]["roofline"]["frac"], 4), "groups": s[q]["groups"], "joined_rows": s[q]["joined_rows"],
                                "oracle_parity": "equal" if s[q].get("oracle_parity") else None} for q in ("q2.1", "q4.1") if q in s}   # (the sentence: bench_details.json; the run fails on a difference)
        if "cpu_baseline" in s:
            for q in ("q2.1", "q4.1"):
                legs["ssb_sf30"][q]["cpu_rows_per_s"] = _short(s["cpu_baseline"][q]["value"])
    if legs:
        out["legs"] = legs
    if "ms_per_step_median_placement" in line:
        out["ms_per_step_median_placement"] = _short(line["ms_per_step_median_placement"], 6)
    if "rccl_ranks" in line:
        out["rccl_ranks"] = line["rccl_ranks"]
    if "cpp_operator_chain_ms" in line:
        chain = line["cpp_operator_chain_ms"]
        legs = out.setdefault("legs", {})
        legs["cpp_operator_chain_ms"] = {k: _short(v) for k, v in chain.items() if isinstance(v, (int, float, bool, str)) and k not in ("pool_candidates", "pool_chosen")}
    if "upload" in line and line["upload"]:
        out["upload_GBps"] = _short(max(u["achieved"] for u in line["upload"].values()), 4)
    if "multi_gpu" in line:
        out["strong_scaling"] = {name: ({k: _short(v) for k, v in leg.items()} if isinstance(leg, dict) else leg)
                                 for name, leg in line["strong_scaling"].items() if name != "note"}
        out["multi_gpu"] = {name: {k: _short(v) for k, v in leg.items() if isinstance(v, (int, float, bool)) or v is None}
                            for name, leg in line["multi_gpu"].items() if isinstance(leg, dict)}
    if details_path:
        out["details"] = os.path.relpath(details_path, ROOT) if details_path.startswith(ROOT) else details_path
    return out


_EVENT_OVERHEAD = []


def event_overhead_ms(lib):
    """What an event pair handed to the launch measures beyond the kernel (hy_profile_event_overhead: an empty kernel, median of 32)."""
    from hyrise_amd import abi
    if not _EVENT_OVERHEAD:
        ms = C.c_float(0)
        abi.check(lib.hy_profile_event_overhead(C.byref(ms)))
        _EVENT_OVERHEAD.append(0.0 if os.environ.get("HY_BENCH_RAW_EVENTS") else float(ms.value))
    return _EVENT_OVERHEAD[0]


def kernel_times(lib):
    """{kernel kind: (ms per timed launch, timed launches)} of the current profiling session (hy_profile_read_kernel).  The scan and
    join kernels are timed by event pairs stamped from their dispatch packets: the elapsed time of an empty kernel measured the same way
    (a few microseconds, `event_overhead_ms` in the line) is taken off every one of them, which is what makes these durations comparable
    with a profiler's per-kernel begin / end timestamps (profiles/r04_bench_kernel_stats.csv)."""
    from hyrise_amd import abi
    out = {}
    overhead = event_overhead_ms(lib)
    for name, kind, packet_events in (("scan", abi.KERNEL_SCAN, True), ("join_probe", abi.KERNEL_JOIN_PROBE, True), ("join_count", abi.KERNEL_JOIN_COUNT, True),
                                      ("join_build", abi.KERNEL_JOIN_BUILD, True), ("aggregate", abi.KERNEL_AGGREGATE, False)):
        km, ln = C.c_float(0), C.c_uint32(0)
        abi.check(lib.hy_profile_read_kernel(kind, C.byref(km), C.byref(ln)))
        per_launch = km.value / ln.value if ln.value else 0.0
        out[name] = (max(per_launch - overhead, 0.0) if packet_events and ln.value else per_launch, int(ln.value))
    return out


_SF10 = []


def sf10_tables():
    """The synthetic TPC-H SF10 orders / lineitem columns, generated once per process (join, aggregate and Q6 legs share them)."""
    from hyrise_amd import tpch
    if not _SF10:
        _SF10.append(tpch.TpchData(scale_factor=10.0, seed=42))
    return _SF10[0]


def join_keys(rank, rows):
    """(o_orderkey, l_orderkey) of this rank's SF10-shaped shard: rank 0 of a full run shares the tables of the other legs."""
    from hyrise_amd import tpch
    if rows:
        data = tpch.TpchData(scale_factor=rows / 5_998_605.2, seed=42 + rank, keys_only=True)
    elif rank == 0 and _SF10:
        data = _SF10[0]
    else:
        data = tpch.TpchData(scale_factor=10.0, seed=42 + rank, keys_only=True)
    return data.o_orderkey, data.l_orderkey


def timed_kernel(lib, torch, run, steps, every=1, kind=None, all_kinds=False):
    """(seconds per call, kernel ms per timed launch): `run` repeated `steps` times after two warm-up calls, in up to five batches whose
    median mean is the figure (the secondary legs only: the headline step is timed as one region, main()).  kind: the kernel whose
    HIP-event time is returned (default: every timed kernel of the call summed / timed launches); all_kinds: the kernel_times dict."""
    from hyrise_amd import abi
    for _ in range(2):
        run()
    abi.check(lib.hy_set_profiling(every))
    batches = min(5, steps)   # (the median of up to five batch means: one descheduled call of twenty otherwise decides a leg's figure)
    means = []
    for b in range(batches):
        calls = steps // batches + (1 if b < steps % batches else 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            run()
        torch.cuda.synchronize()
        means.append((time.perf_counter() - t0) / calls)
    dt = sorted(means)[len(means) // 2]
    kinds = kernel_times(lib)
    km, ln = C.c_float(0), C.c_uint32(0)
    abi.check(lib.hy_profile_read(C.byref(km), C.byref(ln)))
    abi.check(lib.hy_set_profiling(0))
    if all_kinds:
        return dt, kinds
    if kind is not None:
        return dt, kinds[kind][0]
    return dt, km.value / max(1, ln.value)


def roofline_object(kernel, algorithmic_bytes, kernel_ms, traffic=None):
    achieved = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": kernel, "algorithmic_bytes_per_launch": algorithmic_bytes, "kernel_ms": kernel_ms}


class PoolPair:
    """Two PosLists of the library's result-buffer pool (hy_result_pool_acquire_pair): given back on release() / when collected."""

    def __init__(self, lib, rows):
        from hyrise_amd import abi
        self.lib, self.left, self.right = lib, C.c_void_p(), C.c_void_p()
        abi.check(lib.hy_result_pool_acquire_pair(rows, C.byref(self.left), C.byref(self.right)))

    def release(self):
        for side in ("left", "right"):
            pointer = getattr(self, side)
            if pointer is not None and pointer.value:
                self.lib.hy_result_pool_release(pointer)
            setattr(self, side, None)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def device_join(lib, torch, dev, left, right, pairs_capacity, mode=None, asynchronous=False, placements=1):
    """One hy_join_hash with device-memory PosLists from the LIBRARY's result-buffer pool (hy_result_pool_*: what the C++ adapter's JoinHash
    uses too, hyrise_amd/host/hyrise_host.hpp); returns (callable, result struct, buffers).  asynchronous: HY_JOIN_ASYNC -- the call returns
    with its kernels queued (pair count, PosList count and fit flag stay in device memory, hy_join_status); `run.finish()` =
    hy_join_hash_finish waits, reads them and fails like the synchronous call would.  placements > 1: the pool is calibrated first
    (hy_result_pool_calibrate, outside any timed region: `placements` candidate pairs, the fastest is what the pool hands out from then on,
    the median one stays as well -- `run.use_median()` points the join at it)."""
    from hyrise_amd import abi
    mode = abi.JOIN_INNER if mode is None else mode
    # `left` / `right`: a column each, or equally long lists of copies that the calls take in rotation (inputs that come from HBM, not
    # from what the previous call left in the 256 MiB memory-side cache)
    lefts, rights = (left if isinstance(left, (list, tuple)) else [left]), (right if isinstance(right, (list, tuple)) else [right])
    placement = None
    if placements > 1:
        times = (C.c_float * placements)()
        chosen = C.c_uint32(0)
        abi.check(lib.hy_result_pool_calibrate(lefts[0].handle, rights[0].handle, mode, pairs_capacity, placements, abi.POOL_KEEP_MEDIAN, times, C.byref(chosen)))
        placement = {"candidates": placements, "join_ms_per_candidate": [float(f"{t:.4g}") for t in times], "chosen": int(chosen.value),
                     "by": "hy_result_pool_calibrate (the library's result-buffer pool; the C++ adapter's JoinHash draws from the same pool)"}
    best = PoolPair(lib, pairs_capacity)                        # (the calibrated pair if there is one, else a fresh pair placed by the pool's policy)
    median = PoolPair(lib, pairs_capacity) if placement else None   # (second in line: the median candidate)
    slice_offsets = torch.zeros(8192, dtype=torch.int64, device=dev)
    status = torch.zeros(4, dtype=torch.int64, device=dev)
    r = abi.JoinResult()
    r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
    r.capacity = pairs_capacity
    r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), 8000
    if asynchronous:
        r.flags, r.status = abi.JOIN_ASYNC, status.data_ptr()

    def use(pair):
        r.left_pos = pair.left.value
        r.right_pos = pair.right.value if mode == abi.JOIN_INNER else pair.left.value

    use(best)
    turn = [0]

    def run():
        r.radix_bits = 0xFFFFFFFF
        i = turn[0] % len(lefts)
        turn[0] += 1
        abi.check(lib.hy_join_hash(lefts[i].handle, rights[i].handle, mode, C.byref(r)))

    def finish():
        i = (turn[0] - 1) % len(lefts)
        abi.check(lib.hy_join_hash_finish(lefts[i].handle, rights[i].handle, mode, C.byref(r)))

    run.finish = finish
    run.placement = placement
    run.use_median = (lambda: use(median)) if median else None
    run.use_best = lambda: use(best)
    keep = (best, median, slice_offsets, status)
    return run, r, keep


def join_kernels(kinds, n_orders, n, pairs, offset_width=2, pair_bytes=16):
    """The timed kernels of one PK-FK join as roofline objects: HIP-event time per launch against the bytes each must move."""
    out = {}
    if kinds["join_probe"][1]:
        out["pk_emit"] = roofline_object("pk_emit", n * offset_width + pairs * pair_bytes, kinds["join_probe"][0], committed_traffic("pk_emit"))
    if kinds["join_count"][1]:
        out["pk_count"] = roofline_object("pk_count", n * offset_width, kinds["join_count"][0], committed_traffic("pk_count"))
    if kinds["join_build"][1]:
        out["rank_table_fill_waves"] = roofline_object("rank_table_fill_waves", n_orders * 4, kinds["join_build"][0], committed_traffic("rank_table_fill_waves"))
    return out


def join_leg(lib, torch, dev, steps, with_cases, with_cpu, orders_host, lineitem_host, orders, lineitem):
    """Config 3 of BASELINE.json alone: JoinHash(orders, lineitem) on the order key, SF10 -- o_orderkey unencoded int32 (build),
    l_orderkey FrameOfReference + u16 offsets (probe); PosList pairs written to HBM.  Also the reference's two Semi benchmarks
    (tpch_data_micro_benchmark.cpp:299-316)."""
    import numpy as np
    from hyrise_amd import abi, storage
    from hyrise_amd.storage import DeviceColumn
    data = sf10_tables()
    n = data.n_lineitems
    steps = max(3, min(steps, 10))
    run, r, keep = device_join(lib, torch, dev, orders, lineitem, n)   # (the library's result-buffer pool: the pair the headline's calibration left in it)
    dt, kinds = timed_kernel(lib, torch, run, steps, all_kinds=True)
    algorithmic = data.n_orders * 4 + n * 2 + int(r.n_pairs) * 16      # build keys + probe keys + 16 B/pair
    kernels = join_kernels(kinds, data.n_orders, n, int(r.n_pairs))
    info = {"workload": "configs[2]: JoinHash orders x lineitem on the order key, SF10, Inner (o_orderkey int32 values, l_orderkey FrameOfReference u16)"
                        + "; PosLists from the library's result-buffer pool (calibrated once per process: config.output_placement)",
            "rows_per_s": (data.n_orders + n) / dt, "ms_per_join": dt * 1e3, "pairs": int(r.n_pairs), "radix_bits": int(r.radix_bits),
            "output_pos_lists": int(r.n_slices), "algorithmic_bytes": algorithmic,
            "roofline": dict(roofline_object("whole operator (all kernels of one hy_join_hash, host-timed)", algorithmic, dt * 1e3, committed_traffic("hy_join_hash")),
                             dominant_kernel=kernels.get("pk_emit"), kernels=kernels)}
    del keep
    # Semi joins, both directions (BM_HashSemiProbeRelationLarger: lineitem semi-joins orders -- probe lineitem, build orders;
    # BM_HashSemiProbeRelationSmaller: orders semi-joins lineitem -- probe orders, build lineitem with its duplicate keys)
    semi = {}
    run_l, r_l, keep_l = device_join(lib, torch, dev, lineitem, orders, n, abi.JOIN_SEMI)
    dt_l, kinds_l = timed_kernel(lib, torch, run_l, steps, all_kinds=True)
    bytes_l = data.n_orders * 4 + n * 2 + int(r_l.n_pairs) * 8
    semi["probe_relation_larger"] = {"workload": "JoinHash(lineitem, orders, Semi): probe lineitem (FrameOfReference u16), build orders; one PosList of 8 B per match",
                                     "ms_per_join": dt_l * 1e3, "rows_per_s": (data.n_orders + n) / dt_l, "matches": int(r_l.n_pairs), "algorithmic_bytes": bytes_l,
                                     "roofline": dict(roofline_object("whole operator (host-timed)", bytes_l, dt_l * 1e3), kernels=join_kernels(kinds_l, data.n_orders, n, int(r_l.n_pairs), pair_bytes=8))}
    del keep_l
    run_s, r_s, keep_s = device_join(lib, torch, dev, orders, lineitem, data.n_orders, abi.JOIN_SEMI)
    dt_s, kinds_s = timed_kernel(lib, torch, run_s, 3, all_kinds=True)
    bytes_s = data.n_orders * 4 + n * 2 + int(r_s.n_pairs) * 8
    semi["probe_relation_smaller"] = {"workload": "JoinHash(orders, lineitem, Semi): probe orders (int32 values), build lineitem (59 986 052 keys, four per order)",
                                      "ms_per_join": dt_s * 1e3, "rows_per_s": (data.n_orders + n) / dt_s, "matches": int(r_s.n_pairs), "algorithmic_bytes": bytes_s,
                                      "roofline": roofline_object("whole operator (host-timed)", bytes_s, dt_s * 1e3)}
    del keep_s
    info["semi"] = semi
    if with_cases:
        cases = {}
        rng = np.random.default_rng(7)
        # probe side in random order: FrameOfReference offsets widen to 4 bytes, no locality in the lookups (pass 1 hands the ranks to pass 2)
        shuffled = DeviceColumn(storage.make_column(data.l_orderkey[rng.permutation(n)], None, abi.ENC_FRAME_OF_REFERENCE))
        run_s, r_s, keep_s = device_join(lib, torch, dev, orders, shuffled, n)
        dt_s, _ = timed_kernel(lib, torch, run_s, 3)
        cases["shuffled_probe"] = {"ms_per_join": dt_s * 1e3, "rows_per_s": (data.n_orders + n) / dt_s, "pairs": int(r_s.n_pairs),
                                   "GBps_on_algorithmic_bytes": (data.n_orders * 4 + n * 4 + int(r_s.n_pairs) * 16) / dt_s / 1e9}
        del keep_s, shuffled
        # build side in random order (still unique): the (key, RowID) pairs are radix-sorted and the rank table filled from the sorted keys
        build_shuffled = DeviceColumn(storage.make_column(data.o_orderkey[rng.permutation(data.n_orders)], None, abi.ENC_UNENCODED))
        run_b, r_b, keep_b = device_join(lib, torch, dev, build_shuffled, lineitem, n)
        dt_b, _ = timed_kernel(lib, torch, run_b, 3)
        cases["shuffled_build"] = {"ms_per_join": dt_b * 1e3, "rows_per_s": (data.n_orders + n) / dt_b, "pairs": int(r_b.n_pairs)}
        del keep_b, build_shuffled
        # duplicate build keys (every key four times, random order): radix sort + bucket directory, several partners per probe row
        dup_keys = np.repeat(data.o_orderkey[:3_750_000], 4)[rng.permutation(15_000_000)]
        dup_build = DeviceColumn(storage.make_column(dup_keys, None, abi.ENC_UNENCODED))
        probe_rows = 16_000_000
        dup_probe = DeviceColumn(storage.make_column(data.l_orderkey[:probe_rows], None, abi.ENC_FRAME_OF_REFERENCE))
        run_d, r_d, keep_d = device_join(lib, torch, dev, dup_build, dup_probe, probe_rows * 4 + 1024)
        dt_d, _ = timed_kernel(lib, torch, run_d, 3)
        cases["duplicate_build_x4"] = {"ms_per_join": dt_d * 1e3, "rows_per_s": (15_000_000 + probe_rows) / dt_d, "pairs": int(r_d.n_pairs),
                                       "pairs_per_s": int(r_d.n_pairs) / dt_d,
                                       "GBps_on_algorithmic_bytes": (15_000_000 * 4 + probe_rows * 2 + int(r_d.n_pairs) * 16) / dt_d / 1e9}
        del keep_d, dup_build, dup_probe
        # a selective dimension build (the first join of an SSB star plan): 1 000 of a dimension's 1 000 000 keys survive its filter, 180 M
        # fact rows carry random foreign keys -- a key range of 31 250 table words takes the rank table, its bits are staged in LDS
        sel_keys = np.sort(rng.choice(np.arange(1, 1_000_001, dtype=np.int32), 1000, replace=False))
        sel_fact = rng.integers(1, 1_000_001, 180_000_000).astype(np.int32)
        sel_build = DeviceColumn(storage.make_column(sel_keys, None, abi.ENC_UNENCODED))
        sel_probe = DeviceColumn(storage.make_column(sel_fact, None, abi.ENC_UNENCODED))
        run_s, r_s, keep_s = device_join(lib, torch, dev, sel_build, sel_probe, 1_000_000)
        dt_s, _ = timed_kernel(lib, torch, run_s, 3)
        lib.hy_debug_join_used_rank_table.restype = C.c_int
        cases["selective_dimension_build"] = {"ms_per_join": dt_s * 1e3, "rows_per_s": (1000 + 180_000_000) / dt_s, "pairs": int(r_s.n_pairs),
                                              "rank_table": bool(lib.hy_debug_join_used_rank_table()), "GBps_on_algorithmic_bytes": (180_000_000 * 4 + int(r_s.n_pairs) * 16) / dt_s / 1e9,
                                              "note": "1 000 of 1 000 000 dimension keys against 180 M unencoded int32 foreign keys: rank table (not the sorted directory), pk_count_lds / pk_emit<., true>"}
        del keep_s, sel_build, sel_probe, sel_fact
        # the boundary as the adapter uses it today: PosLists returned to HOST memory (0.96 GB over PCIe), one call
        from hyrise_amd.operators import HostJoinResult, join_hash
        t0 = time.perf_counter()
        fresh = join_hash(orders, lineitem, abi.JOIN_INNER)
        dt_fresh = time.perf_counter() - t0
        del fresh
        host = HostJoinResult(n, n // 131070 + 1000)
        host.left[:] = 1
        host.right[:] = 1          # (the pages exist: the adapter writes into pooled PosList memory)
        times = []
        for _ in range(3):
            host.c.radix_bits = 0xFFFFFFFF
            t0 = time.perf_counter()
            abi.check(lib.hy_join_hash(orders.handle, lineitem.handle, abi.JOIN_INNER, C.byref(host.c)))
            times.append(time.perf_counter() - t0)
        dt_h = sorted(times)[1]
        cases["host_memory_result"] = {"ms_per_join": dt_h * 1e3, "rows_per_s": (data.n_orders + n) / dt_h, "pairs": host.n_pairs,
                                       "roofline": pcie_roofline("hy_join_hash with HY_MEM_HOST PosLists: the join's kernels, then both PosLists over the link", host.n_pairs * 16, dt_h),
                                       "ms_into_fresh_buffers": dt_fresh * 1e3,
                                       "note": "HY_MEM_HOST into result buffers whose pages exist (median of 3); ms_into_fresh_buffers: one call that also allocates "
                                               "0.96 GB of numpy memory and takes its first-touch page faults inside the copy (round 3's 56 ms)"}
        del host
        info["cases"] = cases
    if with_cpu:
        info["cpu_baseline"] = cpu_baseline_join(orders_host, lineitem_host, data.n_orders + n)
    return info


def aggregate_kernel_name(lib):
    """Which kernels answered the thread's last hy_aggregate_hash (the library's debug accessor): the Q1 shape takes the two launches of
    csrc/aggregate_small.hpp (their HIP-event bracket spans both)."""
    lib.hy_debug_aggregate_small_domain.restype = C.c_int
    return "sd_groups + sd_wide" if lib.hy_debug_aggregate_small_domain() else "aggregate_rows"


def aggregate_leg(lib, torch, steps, with_cases, with_cpu):
    """Config 4 of BASELINE.json on one GPU, as SURVEY.md 8(d) specifies it: AggregateHash, TPC-H Q1 core -- GROUP BY
    l_returnflag, l_linestatus (string dictionary columns, u8 value ids, passed as their AggregateKey names) with SUM / AVG
    over l_quantity, l_extendedprice, l_discount (DictionarySegment<float>: u8 / u16 / u8 value ids) and COUNT(*), SF10."""
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.operators import aggregate_hash
    from hyrise_amd.storage import DeviceColumn
    data = sf10_tables()
    n = data.n_lineitems
    groupby_host, measures_host, algorithmic = tpch.q1_core_columns(data)
    groupby = [DeviceColumn(c) for c in groupby_host]
    measures = {name: DeviceColumn(c) for name, c in measures_host.items()}

    def spec(columns):
        return [(abi.AGG_SUM, columns["l_quantity"]), (abi.AGG_SUM, columns["l_extendedprice"]), (abi.AGG_AVG, columns["l_quantity"]),
                (abi.AGG_AVG, columns["l_extendedprice"]), (abi.AGG_AVG, columns["l_discount"]), (abi.AGG_COUNT, None)]

    steps = max(3, min(steps, 10))
    holder = {}

    def run():
        holder["result"] = aggregate_hash(groupby, spec(measures), group_capacity=64)

    dt, kernel_ms = timed_kernel(lib, torch, run, steps)
    info = {"workload": "configs[3] on one GPU: AggregateHash Q1 core, GROUP BY l_returnflag, l_linestatus (string dictionaries as key names, u8), "
                        "SUM/AVG over DictionarySegment<float> l_quantity (u8), l_extendedprice (u16), l_discount (u8), COUNT(*), SF10 lineitem",
            "rows_per_s": n / dt, "ms_per_aggregate": dt * 1e3, "groups": int(holder["result"].n_groups), "algorithmic_bytes": algorithmic,
            "roofline": dict(roofline_object("whole operator (host-timed)", algorithmic, dt * 1e3, committed_traffic("hy_aggregate_hash")),
                             dominant_kernel=roofline_object(aggregate_kernel_name(lib), algorithmic, kernel_ms, committed_traffic(aggregate_kernel_name(lib))))}
    if with_cases:   # the same query over unencoded float value segments (round 1's bench shape): 4-byte measures, no dictionary gather
        plain = {name: DeviceColumn(storage.make_column(getattr(data, name), None, abi.ENC_UNENCODED)) for name in ("l_quantity", "l_extendedprice", "l_discount")}

        def run_plain():
            holder["plain"] = aggregate_hash(groupby, spec(plain), group_capacity=64)

        dt_p, kernel_p = timed_kernel(lib, torch, run_plain, steps)
        bytes_p = n * (1 + 1 + 4 + 4 + 4)
        info["cases"] = {"float_value_segments": {"ms_per_aggregate": dt_p * 1e3, "rows_per_s": n / dt_p, "kernel_ms": kernel_p,
                                                  "kernel_GBps": bytes_p / (kernel_p * 1e-3) / 1e9 if kernel_p else None, "algorithmic_bytes": bytes_p}}
    if with_cases:   # many groups: GROUP BY l_orderkey-like int32 keys (100 000 distinct) with SUM + COUNT(*) -- the partitioned path
        import numpy as np
        rng = np.random.default_rng(9)
        many_keys = DeviceColumn(storage.make_column(rng.integers(0, 100_000, n).astype(np.int32), None, abi.ENC_UNENCODED))

        def run_many():
            holder["many"] = aggregate_hash([many_keys], [(abi.AGG_SUM, plain["l_quantity"]), (abi.AGG_COUNT, None)], group_capacity=100_016, result=holder.get("many"))

        dt_m, kernel_m = timed_kernel(lib, torch, run_many, 3)
        info["cases"]["groups_100000"] = {"ms_per_aggregate": dt_m * 1e3, "rows_per_s": n / dt_m, "groups": int(holder["many"].n_groups), "device_ms": kernel_m,
                                          "note": "hash-partitioned path (count, scan, scatter of 16-byte records, one LDS table per partition), the result ordered and written by "
                                                  "kernels; device_ms: the partitioning and aggregating kernels"}
        huge_keys = DeviceColumn(storage.make_column(rng.integers(0, 4_000_000, n).astype(np.int32), None, abi.ENC_UNENCODED))

        def run_huge():
            holder["huge"] = aggregate_hash([huge_keys], [(abi.AGG_SUM, plain["l_quantity"]), (abi.AGG_COUNT, None)], group_capacity=4_000_016, result=holder.get("huge"))

        dt_h, kernel_h = timed_kernel(lib, torch, run_huge, 3)
        info["cases"]["groups_4000000"] = {"ms_per_aggregate": dt_h * 1e3, "rows_per_s": n / dt_h, "groups": int(holder["huge"].n_groups), "device_ms": kernel_h,
                                           "note": "as groups_100000; 100 MB of result columns leave in one copy each into the caller's (already touched) buffers"}
        del huge_keys
        holder.pop("huge", None)
        del many_keys
    if with_cases:   # the plan shape of TPC-H Q1 in Hyrise: TableScan l_shipdate <= 1998-09-02, then AggregateHash over the REFERENCE table it produced
        import torch as _torch
        from hyrise_amd.distributed import HipExecutor
        from hyrise_amd.operators import make_predicate
        ex = HipExecutor(_torch.device("cuda", _torch.cuda.current_device()))
        shipdate = DeviceColumn(storage.make_column(data.l_shipdate, None, abi.ENC_DICTIONARY))
        predicate = make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, tpch.DAY_1998_09_02)

        def run_behind_scan():
            lists = ex.scan_chunked(shipdate, predicate)
            keys = [ex.reference_column_chunked(c, lists) for c in groupby]
            values = {name: ex.reference_column_chunked(c, lists) for name, c in measures.items()}
            holder["behind_scan"] = (ex.aggregate(keys, spec(values)), lists.total)

        dt_s, kernel_s = timed_kernel(lib, torch, run_behind_scan, 3)
        info["cases"]["q1_behind_scan"] = {"ms_per_query": dt_s * 1e3, "rows_per_s": n / dt_s, "qualifying_rows": holder["behind_scan"][1],
                                           "groups": int(holder["behind_scan"][0].n_groups), "aggregate_kernel_ms": kernel_s,
                                           "note": "scan (PosLists stay in HBM) + five reference columns over them + aggregate through the PosLists"}
        del shipdate
    if with_cpu:
        aggregates_host = spec(measures_host)
        info["cpu_baseline"] = cpu_baseline_aggregate(groupby_host, aggregates_host, n)
    return info


def q6_leg(torch, dev, steps):
    """configs[0]'s query on the GPU: TPC-H Q6 at SF10 as the reference plans it -- three TableScans chained through device-resident
    PosLists (hy_table_scan -> hy_poslist_translate -> reference column), Projection l_extendedprice * l_discount, AggregateHash SUM
    (hyrise_amd/tpch.py run_q6); checked against numpy on every run."""
    import numpy as np
    from hyrise_amd import tpch
    from hyrise_amd.distributed import HipExecutor
    from hyrise_amd.storage import DeviceColumn
    data = sf10_tables()
    host = tpch.q6_columns(data)
    columns = {name: DeviceColumn(column) for name, column in host.items()}
    ex = HipExecutor(dev)
    revenue, qualifying = tpch.run_q6(ex, columns)
    keep = (data.l_shipdate >= tpch.DAY_1994_01_01) & (data.l_shipdate < tpch.DAY_1995_01_01) & (data.l_discount >= np.float32(0.05)) & \
           (data.l_discount <= np.float32(0.07)) & (data.l_quantity < 24)
    exact = float((data.l_extendedprice[keep] * data.l_discount[keep]).astype(np.float64).sum())
    if qualifying != int(keep.sum()) or abs(revenue - exact) > 1e-9 * exact:
        raise SystemExit(f"Q6 on the device: {qualifying} rows / {revenue}, numpy says {int(keep.sum())} / {exact}")
    steps = max(3, min(steps, 10))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tpch.run_q6(ex, columns)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    read = sum(s.size * s.width + s.aux_size * 4 for s in host["l_shipdate"].segments)                       # first scan: the whole attribute vector
    # the same query in ONE pass (hy_scan_project_aggregate): no PosList, no product column
    fused_revenue, fused_rows = tpch.q6_fused(columns)
    if fused_rows != qualifying or abs(fused_revenue - exact) > 1e-9 * exact:
        raise SystemExit(f"Q6 fused: {fused_rows} rows / {fused_revenue}, numpy says {int(keep.sum())} / {exact}")
    lib = ex.lib
    fused_dt, fused_kernel_ms = timed_kernel(lib, torch, lambda: tpch.q6_fused(columns), steps)
    # what the fused kernel must read: the three filter columns in full (stored words), price and discount entries of the survivors
    fused_bytes = sum(s.size * s.width for name in ("l_shipdate", "l_discount", "l_quantity") for s in host[name].segments) + qualifying * (4 + 4)
    return {"workload": "configs[0] on one GPU: TPC-H Q6, SF10 lineitem, scan -> scan -> scan -> projection -> aggregate over device-resident PosLists",
            "ms_per_query": dt * 1e3, "lineitem_rows_per_s": data.n_lineitems / dt, "qualifying_rows": qualifying, "revenue": revenue,
            "first_scan_bytes": read, "note": "every intermediate stays in HBM; 8 bytes (a match count) per scan cross to the host",
            "fused": {"workload": "the same query as ONE hy_scan_project_aggregate call (kernel fused_rows): three filters, l_extendedprice * l_discount, SUM + COUNT(*)",
                      "ms_per_query": fused_dt * 1e3, "lineitem_rows_per_s": data.n_lineitems / fused_dt, "speedup_over_the_chain": dt / fused_dt,
                      "roofline": roofline_object("fused_rows", fused_bytes, fused_kernel_ms)}}


def q1_leg(lib, torch, dev, steps):
    """TPC-H Q1, the whole query (tpch_queries.cpp:60-80) at SF10: the operator chain (scan, four ArithmeticExpressions materialised over the
    reference table, AggregateHash of eight aggregates) beside the one-pass hy_scan_project_aggregate; the two results are compared."""
    from hyrise_amd import tpch
    from hyrise_amd.distributed import HipExecutor
    from hyrise_amd.storage import DeviceColumn
    data = sf10_tables()
    host = tpch.q1_columns(data)
    columns = {name: DeviceColumn(column) for name, column in host.items()}
    ex = HipExecutor(dev)
    chain, fused = tpch.run_q1(ex, columns), tpch.q1_fused(columns)
    if chain.n_groups != fused.n_groups:
        raise SystemExit("Q1: the fused pass and the operator chain disagree on the groups")
    for a, name in enumerate(tpch.Q1_AGGREGATES):
        for x, y in zip(chain.column(a), fused.column(a)):
            if abs(x - y) > 1e-9 * max(1.0, abs(y)):
                raise SystemExit(f"Q1 {name}: chain {x}, fused {y}")
    steps = max(3, min(steps, 10))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tpch.run_q1(ex, columns)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fused_dt, fused_kernel_ms = timed_kernel(lib, torch, lambda: tpch.q1_fused(columns), steps)
    kernel = "fused_small_domain" if lib.hy_debug_aggregate_small_domain() == 2 else "fused_rows"   # (which kernel answered the last call)
    fused_bytes = sum(s.size * s.width + s.aux_size * (s.aux.dtype.itemsize if s.aux is not None else 0) for column in host.values() for s in column.segments)
    return {"workload": "TPC-H Q1 (the whole query: l_shipdate <= 1998-09-02, two expressions, eight aggregates, GROUP BY l_returnflag, l_linestatus), SF10 lineitem, "
                        "columns encoded as config 4 specifies",
            "chain_ms_per_query": dt * 1e3, "groups": fused.n_groups, "count_order": fused.column(7),
            "fused": {"workload": f"ONE hy_scan_project_aggregate call (kernel {kernel})", "ms_per_query": fused_dt * 1e3, "lineitem_rows_per_s": data.n_lineitems / fused_dt,
                      "speedup_over_the_chain": dt / fused_dt, "roofline": roofline_object(kernel, fused_bytes, fused_kernel_ms)}}


def scan_cases(lib, torch, dev, steps, days, column, step_fn, counts, rows, width):
    """The other scans SURVEY.md 8(d) lists for config 2, same column unless stated."""
    import numpy as np
    from hyrise_amd import abi, storage, tpch
    from hyrise_amd.operators import make_predicate
    from hyrise_amd.storage import DeviceColumn
    out = {}

    def measure(run, bytes_of):
        dt, km = timed_kernel(lib, torch, run, steps, PROFILE_EVERY)
        m = int(counts.sum().item())
        return {"rows_per_s": rows / dt, "ms_per_step": dt * 1e3, "matches": m, "kernel_ms": km,
                "kernel_GBps": bytes_of(m) / (km * 1e-3) / 1e9 if km else None}

    class Rotating:
        """COLUMN_COPIES device copies of a column, scanned in rotation: like the headline's, a case's input comes from HBM, not from the
        256 MiB Infinity Cache a single repeated scan of a 60 - 240 MB column would partly live in (profiles/r03_scan_stores.txt shows what
        that residency is worth -- and that it decides which store flavour wins)."""

        def __init__(self, host):
            self.host = host
            self.copies = [DeviceColumn(host) for _ in range(COLUMN_COPIES)]
            self.turn = 0

        def scan(self, pred):
            step_fn(pred, self.copies[self.turn % COLUMN_COPIES])
            self.turn += 1

    for name, pred in (("q1_le_1998-09-02", make_predicate(abi.PRED_LESS_THAN_EQUALS, abi.TYPE_INT, tpch.DAY_1998_09_02)),
                       ("q6_between_1994", make_predicate(abi.PRED_BETWEEN_UPPER_EXCLUSIVE, abi.TYPE_INT, tpch.DAY_1994_01_01, tpch.DAY_1995_01_01)),
                       ("point_eq_1995-06-17", make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, tpch.CURRENT_DATE)),
                       ("is_null", make_predicate(abi.PRED_IS_NULL, abi.TYPE_INT))):
        out[name] = measure(lambda p=pred: step_fn(p, None), lambda m: rows * width + m * 8)   # (None: the headline's copies in rotation)
    # the boundary as the adapter uses it today: PosLists returned to HOST memory (PCIe-inclusive, never the headline value)
    from hyrise_amd.operators import HostScanResult, table_scan
    pred = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)
    t0 = time.perf_counter()
    fresh = table_scan(column, pred)
    dt_fresh = time.perf_counter() - t0
    del fresh
    host = HostScanResult(column.n_chunks, rows, 0)
    host.matches[:] = 1            # (the pages exist)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        abi.check(lib.hy_table_scan(column.handle, C.byref(pred), None, 0, C.byref(host.c)))
        times.append(time.perf_counter() - t0)
    dt_h = sorted(times)[1]
    out["host_memory_result_lt_1995"] = {"rows_per_s": rows / dt_h, "ms_per_step": dt_h * 1e3, "matches": host.total,
                                         "roofline": pcie_roofline("hy_table_scan with HY_MEM_HOST PosLists: scan, packing the chunk regions, the PosLists over the link", host.total * 8, dt_h),
                                         "ms_into_fresh_buffers": dt_fresh * 1e3,
                                         "note": "HY_MEM_HOST into a result buffer whose pages exist (median of 3); ms_into_fresh_buffers: one call that also allocates the numpy buffer"}
    del host
    # the same dates as unencoded int32 values (ValueSegment<int32>: the 4-byte streaming instantiation)
    values = Rotating(storage.make_column(days, None, abi.ENC_UNENCODED))
    pred = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)
    out["int32_value_segments_lt_1995"] = measure(lambda: values.scan(pred), lambda m: rows * 4 + m * 8)
    del values
    # Sorted chunks (Chunk::individually_sorted_by; hyriseBenchmarkTPCH --clustering sorts lineitem by l_shipdate, tpch_benchmark.cpp:61-64):
    # the reference's SortedSegmentSearch, here prepare_jobs' binary searches -- the kernel writes the positions and reads no row.
    #   clustered table: the chunks follow each other in date order, all but the chunk that holds 1995-01-01 match entirely or not at all
    #   per-chunk sorted: every chunk spans all dates and is sorted on its own (every chunk emits a range)
    def flagged(values_of_chunks):
        host = storage.make_column(values_of_chunks, None, abi.ENC_DICTIONARY)
        for segment in host.segments:
            segment.sorted_by = abi.SORT_ASCENDING_NULLS_FIRST
        return DeviceColumn(host)
    pred = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)
    clustered_days = np.sort(days)
    clustered = flagged(clustered_days)
    chunk_rows = abi.CHUNK_DEFAULT_SIZE
    # (all-match chunks own no RowIDs -- chunk_state says ALL_MATCH, table_scan.cpp:201-205 -- and no-match chunks none: only the chunk that
    #  holds the literal writes positions)
    written = sum(int((clustered_days[b:b + chunk_rows] < tpch.DAY_1995_01_01).sum()) for b in range(0, rows, chunk_rows)
                  if clustered_days[b] < tpch.DAY_1995_01_01 <= clustered_days[min(rows, b + chunk_rows) - 1])
    out["sorted_clustered_table_lt_1995"] = dict(measure(lambda: step_fn(pred, clustered), lambda m: written * 8), row_ids_written=written,
                                                 note="every chunk but one takes an early-out (all rows / no row match: decided from its dictionary), the one that holds the "
                                                      "literal is searched (JOB_RANGE); nothing of the column is read: the time is the launch and 916 workgroups' bookkeeping")
    del clustered, clustered_days
    per_chunk = days.copy()
    for begin in range(0, rows, chunk_rows):
        per_chunk[begin:begin + chunk_rows].sort()
    sorted_chunks = flagged(per_chunk)
    out["sorted_chunks_lt_1995"] = dict(measure(lambda: step_fn(pred, sorted_chunks), lambda m: m * 8),
                                        note="every chunk: two 64-ary searches (three dependent loads each), then its range of positions is written; no row is read")
    del sorted_chunks, per_chunk
    # Compressed layouts scanned in place: l_shipdate's value ids need 12 bits (2 526 dates + the NULL id): a BitPackingVector of 12 bits per
    # row instead of FixedWidthInteger<2> (vector_compression/bitpacking), and RunLengthSegment<int32> over the clustered dates
    packed_host = storage.make_column(days, None, abi.ENC_DICTIONARY)
    packed_host = storage.HostColumn([storage.bit_pack_segment(segment) for segment in packed_host.segments], packed_host.data_type)
    packed_bits = int(packed_host.segments[0].bits)
    packed = Rotating(packed_host)
    out["bit_packed_value_ids_lt_1995"] = dict(measure(lambda: packed.scan(pred), lambda m: rows * packed_bits // 8 + m * 8), bits_per_value_id=packed_bits,
                                               note="the generic instantiation unpacks the value ids in registers; bytes counted: bits / 8 per row + 8 per match")
    del packed, packed_host
    clustered_days = np.sort(days)
    runs_host = storage.HostColumn([storage.encode_run_length(clustered_days[begin:begin + chunk_rows]) for begin in range(0, rows, chunk_rows)], abi.TYPE_INT)
    n_runs = sum(segment.aux_size for segment in runs_host.segments)
    runs = DeviceColumn(runs_host)
    out["run_length_clustered_dates_lt_1995"] = dict(measure(lambda: step_fn(pred, runs), lambda m: n_runs * 8 + m * 8), runs=n_runs,
                                                     note="RunLengthSegment<int32> read in place, run by run: one search of the end positions per wave (2048 rows), every run tested once; bytes counted: 8 per run + 8 per match "
                                                          "(sorted_chunks_lt_1995 writes the same positions without reading anything: the floor of this case)")
    del runs, runs_host, clustered_days
    # l_shipdate as Hyrise's schema has it: DictionarySegment<pmr_string> of ISO dates -- the same attribute vectors, the literal resolved
    # per chunk on the host (lower / upper bound in 916 string dictionaries: reported beside the scan as host_literal_resolution_ms)
    from hyrise_amd.operators import string_predicate
    strings_host, dictionaries = tpch.string_date_column(column.host)
    strings = Rotating(strings_host)
    string_pred = string_predicate(abi.PRED_LESS_THAN, dictionaries, "1995-01-01")
    out["string_dictionary_twin_lt_1995"] = measure(lambda: strings.scan(string_pred), lambda m: rows * width + m * 8)
    t0 = time.perf_counter()
    string_predicate(abi.PRED_LESS_THAN, dictionaries, "1995-01-01")
    out["string_dictionary_twin_lt_1995"]["host_literal_resolution_ms"] = (time.perf_counter() - t0) * 1e3
    del strings
    # the other streaming instantiations: u8 value ids (l_returnflag = 'R': a dictionary of three strings, scanned as value ids) and
    # FrameOfReference offsets (l_orderkey < literal: u16 offsets + one minimum per 2048-row block)
    rng = np.random.default_rng(44)
    flags = Rotating(storage.make_column(rng.integers(0, 3, rows).astype(np.int32), None, abi.ENC_DICTIONARY))
    pred = make_predicate(abi.PRED_EQUALS, abi.TYPE_INT, 2)
    out["u8_value_ids_returnflag_eq"] = measure(lambda: flags.scan(pred), lambda m: rows * 1 + m * 8)
    del flags
    order_keys = np.sort(rng.integers(1, 60_000_000, rows).astype(np.int32))
    keys = Rotating(storage.make_column(order_keys, None, abi.ENC_FRAME_OF_REFERENCE))
    pred = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, 20_000_000)
    out["frame_of_reference_orderkey_lt"] = dict(measure(lambda: keys.scan(pred), lambda m: rows * keys.host.segments[0].width + m * 8), offset_width=int(keys.host.segments[0].width))
    del keys, order_keys
    # ColumnVsColumn (Q4 / Q12): l_commitdate < l_receiptdate, both dictionary-encoded with u16 value ids
    rng = np.random.default_rng(43)
    orderdate = rng.integers(0, tpch.LAST_ORDERDATE + 1, rows, dtype=np.int32)
    commit = DeviceColumn(storage.make_column((orderdate + rng.integers(30, 91, rows, dtype=np.int32)).astype(np.int32), None, abi.ENC_DICTIONARY))
    receipt = DeviceColumn(storage.make_column((orderdate + rng.integers(2, 152, rows, dtype=np.int32)).astype(np.int32), None, abi.ENC_DICTIONARY))
    matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
    offsets = torch.zeros(commit.n_chunks + 1, dtype=torch.int64, device=dev)
    counts2 = torch.zeros(commit.n_chunks, dtype=torch.int32, device=dev)
    result = abi.ScanResult()
    result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS
    result.matches, result.capacity = matches.data_ptr(), rows
    result.offsets, result.counts = offsets.data_ptr(), counts2.data_ptr()

    def run_columns():
        abi.check(lib.hy_table_scan_columns(commit.handle, receipt.handle, abi.PRED_LESS_THAN, C.byref(result)))

    dt, km = timed_kernel(lib, torch, run_columns, steps, PROFILE_EVERY)
    m = int(counts2.sum().item())
    out["column_vs_column_commit_lt_receipt"] = {"rows_per_s": rows / dt, "ms_per_step": dt * 1e3, "matches": m, "kernel_ms": km,
                                                  "kernel_GBps": (rows * 4 + m * 8) / (km * 1e-3) / 1e9 if km else None}
    return out


def device_identity(torch, local_rank):
    """A 64-bit identity of the GPU this rank computes on: PCI domain / bus / device where the runtime tells, else the device's UUID bytes."""
    import hashlib
    props = torch.cuda.get_device_properties(local_rank)
    text = "|".join(str(getattr(props, name, "")) for name in ("pci_domain_id", "pci_bus_id", "pci_device_id", "uuid", "name")) + "|" + str(torch.cuda.current_device())
    return int.from_bytes(hashlib.sha256((os.uname().nodename + "|" + text).encode()).digest()[:7], "little")


def count_distinct(identities):
    return len(set(int(x) for x in identities))


def require_distinct_gpus(rccl_ranks, world, share_gpu):
    """`--gpus N` is a claim about hardware: N ranks on fewer than N GPUs is not an N-GPU measurement (HY_BENCH_SHARE_GPU: the debug mode that
    says so itself -- gloo, control flow only)."""
    if rccl_ranks < world and not share_gpu:
        raise SystemExit(f"--gpus {world}: the {world} ranks run on {rccl_ranks} distinct GPU(s) (HY_BENCH_SHARE_GPU=1 exercises the control flow on one GPU)")


def distinct_devices(torch, dist, local_rank, world, share_gpu):
    """all-gather of every rank's device identity over the job's backend (RCCL unless the ranks share a GPU) -> number of distinct GPUs."""
    mine = torch.tensor([device_identity(torch, local_rank)], dtype=torch.int64, device="cpu" if share_gpu else torch.device("cuda", local_rank))
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return count_distinct(int(g.item()) for g in gathered)


def cpp_operator_chain(placements):
    """tests/cpp/operator_chain --time: TableScan -> JoinHash [-> AggregateHash] and JoinHash alone through `_on_execute()` of the C++ adapter
    (hyrise_amd/host/hyrise_host.hpp) at SF10 size, with the intermediates as DevicePosLists in HBM and -- beside it -- with host-memory results
    (the boundary as rounds 1-5 used it).  The binary is a separate process: its own tables (same shapes), its own calibration of the library's
    result-buffer pool.  It is timed WITHOUT the oracle (that is --oracle, the tests' business)."""
    import subprocess
    binary = os.path.join(ROOT, "tests", "cpp", "operator_chain")
    if not os.path.exists(binary):
        return {"error": "tests/cpp/operator_chain missing: run __graft_entry__.build()"}
    try:
        proc = subprocess.run([binary, "--time", "7", "--calibrate", str(max(1, min(placements, 12)))], capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        return {"error": "tests/cpp/operator_chain timed out"}
    for text in reversed(proc.stdout.splitlines()):
        if text.startswith("{"):
            out = json.loads(text)["cpp_operator_chain_ms"]
            out["ok"] = proc.returncode == 0 and "OPERATOR CHAIN OK" in proc.stdout
            return out
    return {"error": (proc.stdout + proc.stderr)[-400:]}


def main():
    args = parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_cases = args.no_join = args.no_aggregate = args.no_multi = args.no_ssb = True
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
    if args.switch:
        abi.switches(dict(s.split("=", 1) for s in args.switch)).__enter__()   # (for the life of the process)
    stream = torch.cuda.current_stream()
    abi.check(lib.hy_set_stream(C.c_void_p(stream.cuda_stream)))

    # ---- data: this rank's SF10-shaped shard (seed differs per rank), encoded like Hyrise, uploaded once -------------
    from hyrise_amd import storage
    rows = args.rows or tpch.LINEITEM_ROWS_SF10
    single = rank == 0 and world == 1 and not args.rows
    if single and not (args.no_join and args.no_aggregate and args.no_cases):
        sf10_tables()   # (the other legs need every column: generate once, the headline's join keys are two of them)
    days, host_column = tpch.shipdate_column(rows, seed=42 + rank)
    uploads = {}
    # (a thread's first upload also allocates its pinned staging block, 64 MiB, about 0.1 s: taken here, not charged to a column)
    import numpy as _np
    storage.DeviceColumn(storage.make_column(_np.zeros(1 << 16, dtype=_np.int32), None, abi.ENC_UNENCODED)).close()
    columns = [timed_upload("l_shipdate (DictionarySegment<int32>, u16 value ids)", host_column, uploads) for _ in range(COLUMN_COPIES)]
    n_chunks = host_column.n_chunks
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)
    o_orderkey, l_orderkey = join_keys(rank, args.rows)
    n_orders, n_lineitems = len(o_orderkey), len(l_orderkey)
    orders_host = storage.make_column(o_orderkey, None, abi.ENC_UNENCODED)
    lineitem_host = storage.make_column(l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
    # like the scanned column, the join's inputs exist COLUMN_COPIES times and the steps take them in rotation (540 MB of keys)
    orders_copies = [timed_upload("o_orderkey (ValueSegment<int32>)", orders_host, uploads) for _ in range(COLUMN_COPIES)]
    lineitem_copies = [timed_upload("l_orderkey (FrameOfReference, u16 offsets)", lineitem_host, uploads) for _ in range(COLUMN_COPIES)]
    orders, lineitem = orders_copies[0], lineitem_copies[0]
    offset_width = int(lineitem_host.segments[0].width)

    dev = torch.device("cuda", local_rank)
    matches = torch.empty((rows, 2), dtype=torch.int32, device=dev)
    offsets = torch.zeros(n_chunks + 1, dtype=torch.int64, device=dev)
    result = abi.ScanResult()
    result.mem = abi.MEM_DEVICE
    result.matches, result.capacity = matches.data_ptr(), rows
    counts = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    result.flags = abi.SCAN_CHUNK_REGIONS  # chunk c's PosList at matches[offsets[c] : offsets[c] + counts[c]]
    result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()
    # HY_JOIN_ASYNC: no host round trip per join -- the step's kernels are queued back to back, the pair count is read once, after the timed region
    run_join, join_result, join_buffers = device_join(lib, torch, dev, orders_copies, lineitem_copies, n_lineitems, asynchronous=not os.environ.get("HY_BENCH_SYNC_JOIN"),
                                                      placements=args.placements)
    for _ in range(2 * COLUMN_COPIES):   # (setup, not warm-up: the first join over a resident build column looks at its keys in two passes and
        run_join()                       #  leaves their range behind as the column's hint; every later join fills its table in one checked pass)
    run_join.finish()
    turn = [0]

    def scan_step(pred=predicate, column=None):
        if column is None:
            column = columns[turn[0] % COLUMN_COPIES]
            turn[0] += 1
        abi.check(lib.hy_table_scan(column.handle, C.byref(pred), None, 0, C.byref(result)))

    def step():
        scan_step()
        run_join()

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
    run_join.finish()   # (the last join's pair count and fit flag: device memory until now)
    kinds = kernel_times(lib)
    abi.check(lib.hy_set_profiling(0))
    # The same timed region with the join's PosLists in the MEDIAN candidate of the calibration -- what a pool that is not calibrated gets on
    # average (`ms_per_step` is the calibrated placement, this is the honest second number)
    elapsed_median = None
    if run_join.use_median:
        run_join.use_median()
        for _ in range(max(2, args.warmup)):
            step()
        abi.check(lib.hy_set_profiling(0 if os.environ.get("HY_BENCH_NO_EVENTS") else PROFILE_EVERY))   # (the same event pairs on the stream as above)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed_median = time.perf_counter() - t0
        run_join.finish()
        abi.check(lib.hy_set_profiling(0))
        run_join.use_best()

    n_matches = int(counts.sum().item())
    expected = int((days < tpch.DAY_1995_01_01).sum())
    if n_matches != expected:
        raise SystemExit(f"rank {rank}: scan produced {n_matches} matches, numpy says {expected}")
    n_pairs = int(join_result.n_pairs)
    if n_pairs != n_lineitems:
        raise SystemExit(f"rank {rank}: join produced {n_pairs} pairs, every one of the {n_lineitems} lineitems has exactly one order")

    rccl_ranks = None
    if dist is not None:
        t = torch.tensor([elapsed, elapsed_median or 0.0], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if elapsed_median is not None:
            elapsed_median = float(t[1].item())
        # how many DISTINCT GPUs carry the ranks (all-gathered PCI bus ids): a run that claims N GPUs must have N
        rccl_ranks = distinct_devices(torch, dist, local_rank, world, share_gpu)
        require_distinct_gpus(rccl_ranks, world, share_gpu)

    ms_per_step = elapsed / args.steps * 1e3
    ms_per_step_median_placement = elapsed_median / args.steps * 1e3 if elapsed_median is not None else None
    step_rows = rows + n_orders + n_lineitems
    value = step_rows * world / (elapsed / args.steps)

    # algorithmic bytes per step: attribute vector + RowIDs written | build keys + probe keys + 16 B per pair
    width = host_column.segments[0].width
    scan_bytes = rows * width + n_matches * 8
    join_bytes = n_orders * 4 + n_lineitems * offset_width + n_pairs * 16
    step_kernels = {"scan_slices": roofline_object("scan_slices", scan_bytes, kinds["scan"][0], committed_traffic("scan_slices"))}
    step_kernels.update(join_kernels(kinds, n_orders, n_lineitems, n_pairs, offset_width))

    # the join WITHOUT the build column's key hint (what the first join over a freshly loaded column costs: two passes over the build keys)
    first_join_no_hint_ms = None
    if rank == 0:
        with abi.option(abi.OPT_JOIN_HINT, 0):
            times = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_join()
                run_join.finish()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
        first_join_no_hint_ms = sorted(times[1:])[len(times[1:]) // 2] * 1e3

    # ---- the operators alone, the other configs (N = 1) ---------------------------------------------------------------
    scan_info = None
    if single:
        dt_scan, scan_kernel_ms = timed_kernel(lib, torch, scan_step, args.steps, PROFILE_EVERY, kind="scan")
        scan_info = {"workload": "configs[1] alone: TableScan ColumnVsValue, SF10 lineitem l_shipdate < 1995-01-01, three column copies in rotation",
                     "rows_per_s": rows / dt_scan, "ms_per_scan": dt_scan * 1e3, "matches": n_matches, "algorithmic_bytes": scan_bytes,
                     "roofline": dict(roofline_object("whole operator (host-timed)", scan_bytes, dt_scan * 1e3),
                                      dominant_kernel=roofline_object("scan_slices", scan_bytes, scan_kernel_ms, committed_traffic("scan_slices")))}
    extra_cases = None
    if single and not args.no_cases:
        extra_cases = scan_cases(lib, torch, dev, args.steps, days, columns[0], scan_step, counts, rows, width)
    for pair in join_buffers[:2]:   # (back to the library's pool: the join leg and the C++ operator chain draw the calibrated pair from it)
        if pair is not None:
            pair.release()
    del join_buffers, orders_copies[1:], lineitem_copies[1:]
    join_info = (join_leg(lib, torch, dev, args.steps, not args.no_cases, not args.no_cpu_baseline, orders_host, lineitem_host, orders, lineitem)
                 if single and not args.no_join else None)
    aggregate_info = aggregate_leg(lib, torch, args.steps, not args.no_cases, not args.no_cpu_baseline) if single and not args.no_aggregate else None

    q6_info = q6_leg(torch, dev, args.steps) if single and not args.no_cases else None
    q1_info = q1_leg(lib, torch, dev, args.steps) if single and not args.no_cases else None

    cpp_chain = None
    if single and not args.no_cases:
        abi.check(lib.hy_result_pool_trim())   # (this process's pooled PosLists: the binary is another process on the same GPU)
        cpp_chain = cpp_operator_chain(args.placements)
    multi = None
    if world > 1 and not args.no_multi and not args.rows:
        from hyrise_amd import distributed
        multi = distributed.bench_legs(lib, torch, dist, dev, rank, world, share_gpu, steps=max(3, min(args.steps, 10)))
    ssb_info = None
    if not args.no_ssb and not args.rows:   # config 5: SSB SF30 star joins, lineorder chunk-sharded over the ranks
        del columns, matches, orders, lineitem
        from hyrise_amd import ssb
        ssb_info = ssb.bench(30.0, 3, world, rank, dist, share_gpu, local_rank)
        for query in ("q2.1", "q4.1"):   # the whole query against the HBM roofline: its algorithmic bytes over its host-timed duration
            ssb_info[query]["roofline"] = roofline_object("whole query as one hy_star_join_aggregate call (dimension tables, one probe pass over lineorder, the aggregate inside it; host-timed)",
                                                          ssb_info[query]["algorithmic_bytes"], ssb_info[query]["ms"])
        device_rows = {query: ssb_info[query].pop("_rows") for query in ("q2.1", "q4.1")}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            ssb_info["cpu_baseline"] = cpu_baseline_ssb()
            # parity at BASELINE's size: the oracle executor's SF30 group rows and joined-row counts against the HIP plan's -- integer sums, exact
            for query in ("q2.1", "q4.1"):
                want, joined = ssb_info["cpu_baseline"][query].pop("_rows"), ssb_info["cpu_baseline"][query].pop("_joined")
                if device_rows[query] != want or ssb_info[query]["joined_rows"] != joined:
                    raise SystemExit(f"SSB SF30 {query}: the HIP plan ({ssb_info[query]['joined_rows']} joined rows, {len(device_rows[query])} groups) differs from the "
                                     f"CPU oracle ({joined} joined rows, {len(want)} groups)")
                ssb_info[query]["oracle_parity"] = f"{len(want)} group rows and {joined} joined rows equal to the CPU oracle's at SF30"

    if rank == 0:
        step_roofline = roofline_object("TableScan + JoinHash step: every kernel and launch gap of one hy_table_scan + one hy_join_hash, host-timed over the timed region",
                                        scan_bytes + join_bytes, ms_per_step)
        step_roofline.update(dominant_kernel=step_kernels.get("pk_emit") or step_kernels["scan_slices"], kernels=step_kernels,
                             launches_timed={name: kinds[name][1] for name in ("scan", "join_probe", "join_count", "join_build")},
                             traffic_source=traffic_source(), algorithmic_bytes={"scan": scan_bytes, "join": join_bytes}, event_overhead_ms=event_overhead_ms(lib))
        step_roofline["traffic"] = committed_traffic("step")
        line = {
            "metric": "rows/sec TableScan+JoinHash, TPC-H SF10 lineitem (ColumnVsValue l_shipdate < 1995-01-01, then JoinHash orders x lineitem on the order key)",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "configs[1] + configs[2] per step: TableScan ColumnVsValue on SF10 lineitem l_shipdate (DictionarySegment<int32> + u16 attribute vectors, "
                                   f"916 chunks x 65535 rows, {COLUMN_COPIES} column copies in rotation) and JoinHash orders x lineitem on the order key (o_orderkey int32 values, "
                                   f"l_orderkey FrameOfReference u16; Inner; {COLUMN_COPIES} copies of both key columns in rotation; HY_JOIN_ASYNC: pair count read after the timed region); "
                                   "columns, PosLists and pair lists resident in HBM",
                       "rows_per_step_per_gpu": step_rows, "scan_rows": rows, "build_rows": n_orders, "probe_rows": n_lineitems, "chunks_per_gpu": n_chunks,
                       "scan_selectivity": n_matches / rows, "join_pairs": n_pairs, "parallelism": f"chunk-sharded x{world}, no collective"},
            "roofline": step_roofline,
            # H2D once per column (reported apart from the resident-column runs -- every other figure in this line starts behind it)
            "upload": uploads,
        }
        if scan_info:
            line["scan"] = scan_info
        if extra_cases:
            line["cases"] = extra_cases
        if join_info:
            line["join"] = join_info
        if aggregate_info:
            line["aggregate"] = aggregate_info
        if q6_info:
            line["q6"] = q6_info
        if q1_info:
            line["q1"] = q1_info
        if multi:
            line["multi_gpu"] = multi
            # ONE SF10 database split over the ranks -- the strong-scaling figures, next to the (weak-scaling) `value`: rows/s of the whole job
            line["strong_scaling"] = {name: {"rows_per_s": leg["rows_per_s"], "ms": leg["ms"]} for name, leg in multi.items() if isinstance(leg, dict) and "rows_per_s" in leg}
            line["strong_scaling"]["n_gpus"] = world
            line["strong_scaling"]["note"] = ("scan_strong: the SF10 l_shipdate scan, chunks sharded, no collective; aggregate_q1: per-rank partials + one RCCL all-reduce; "
                                              "join_broadcast_build: all-gather of the build column; join_repartition: all-to-all of (key, RowID) tuples by key % G and back")
        if ssb_info:
            line["ssb"] = dict(ssb_info, workload="configs[4]: SSB SF30 Q2.1 / Q4.1 star joins as ONE hy_star_join_aggregate call each (the plan dimension scans -> one JoinHash per "
                                                   "dimension -> Projection -> AggregateHash, run as one probe pass over lineorder with the aggregate inside it), synthetic tables per the SSB specification")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_step(host_column, predicate, rows, orders_host, lineitem_host)
        if run_join.placement:
            line["config"]["output_placement"] = run_join.placement
        if ms_per_step_median_placement is not None:
            line["ms_per_step_median_placement"] = ms_per_step_median_placement
        if rccl_ranks is not None:
            line["rccl_ranks"] = rccl_ranks
        if cpp_chain is not None:
            line["cpp_operator_chain_ms"] = cpp_chain
        if first_join_no_hint_ms is not None:
            line.setdefault("join", {})["first_join_no_hint_ms"] = first_join_no_hint_ms
        # the reference's benchmark runner writes its detailed results to a file (-o) and prints a summary
        # (src/benchmarklib/benchmark_runner.cpp:443-531): the full object goes to --details, the LAST stdout line is its summary
        write_details(line, args.details)
        print(json.dumps(compact_line(line, args.details), separators=(",", ":")))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
