"""bench.py's last stdout line is what the driver parses: it must stay a few KB and carry the contract's keys (round 4's 22 KB line was cut
off and could not be parsed).  The full object goes to a file, like the reference's benchmark runner does (benchmark_runner.cpp:443-531)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def full_line():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as fh:   # a full result object as bench.py builds it (round 4, one GPU)
        return json.load(fh)


def check(compact):
    text = json.dumps(compact, separators=(",", ":"))
    assert "\n" not in text and len(text) < 4096, len(text)
    parsed = json.loads(text)
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "dominant_kernel", "kernels"):
        assert key in parsed["roofline"], key
    assert parsed["roofline"]["bound"] == "hbm" and 0 < parsed["roofline"]["frac"] < 1
    assert abs(parsed["roofline"]["frac"] - parsed["roofline"]["achieved"] / parsed["roofline"]["peak"]) < 1e-3
    assert set(parsed["roofline"]["kernels"]) == {"scan_slices", "pk_emit", "pk_count", "rank_table_fill_waves"}
    assert "workload" in parsed["config"] and "hint" in parsed["config"]["workload"] and "model" not in parsed["config"]
    return parsed


def test_single_gpu_line_is_compact_and_complete():
    import bench
    line = full_line()
    line.setdefault("join", {})["first_join_no_hint_ms"] = 0.484
    parsed = check(bench.compact_line(line, os.path.join(ROOT, "bench_details.json")))
    for key in CONTRACT:
        assert key in parsed, key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in parsed["cpu_baseline"], key
    assert parsed["join"]["first_join_no_hint_ms"] == 0.484
    assert parsed["details"] == "bench_details.json"
    assert abs(parsed["value"] - line["value"]) / line["value"] < 1e-4 and abs(parsed["ms_per_step"] - line["ms_per_step"]) < 1e-5
    assert parsed["legs"]["ssb_sf30"]["q2.1"]["groups"] == 280


def test_multi_gpu_line_is_compact():
    import bench
    line = full_line()
    for key in ("scan", "cases", "join", "aggregate", "q6", "q1", "cpu_baseline"):   # (N > 1 runs none of the single-GPU legs)
        line.pop(key, None)
    line["n_gpus"] = 8
    legs = {name: {"rows_per_s": 1.5e11, "ms": 0.4, "note": "x" * 400, "bytes_exchanged": 123456789, "parity": True}
            for name in ("scan_strong", "aggregate_q1", "join_broadcast_build", "join_repartition")}
    line["multi_gpu"] = legs
    line["strong_scaling"] = dict({name: {"rows_per_s": leg["rows_per_s"], "ms": leg["ms"]} for name, leg in legs.items()}, n_gpus=8, note="y" * 600)
    parsed = check(bench.compact_line(line))
    assert parsed["strong_scaling"]["n_gpus"] == 8 and "note" not in parsed["strong_scaling"]
    assert parsed["multi_gpu"]["join_repartition"]["bytes_exchanged"] == 123456789 and "note" not in parsed["multi_gpu"]["join_repartition"]


def test_details_file_holds_the_full_object(tmp_path):
    import bench
    line = full_line()
    path = str(tmp_path / "details.json")
    bench.write_details(line, path)
    with open(path) as fh:
        assert json.load(fh) == line


def test_multi_gpu_runs_count_their_distinct_gpus():
    """`--gpus N` must be N GPUs: the ranks all-gather their device identities (`rccl_ranks` in the line) and the run refuses fewer than N
    unless HY_BENCH_SHARE_GPU says it is the one-GPU debug mode."""
    import pytest
    import bench
    assert bench.count_distinct([11, 12, 13, 14]) == 4 and bench.count_distinct([11, 11, 12, 12]) == 2
    bench.require_distinct_gpus(8, 8, False)
    bench.require_distinct_gpus(1, 2, True)
    with pytest.raises(SystemExit):
        bench.require_distinct_gpus(1, 2, False)
    line = full_line()
    for key in ("scan", "cases", "join", "aggregate", "q6", "q1", "cpu_baseline"):
        line.pop(key, None)
    line["n_gpus"], line["rccl_ranks"] = 4, 4
    assert check(bench.compact_line(line))["rccl_ranks"] == 4


def test_single_gpu_scale_line_has_the_bench_config():
    """The driver's SCALE run at N = 1 is the BENCH command: the line's `config` (and metric / unit / dtype / scaling) may not depend on the
    step counts or on anything but the workload, so the two lines describe the same configuration."""
    import bench
    a, b = full_line(), full_line()
    b["steps"], b["warmup"], b["ms_per_step"], b["value"] = 7, 1, a["ms_per_step"] * 1.01, a["value"] / 1.01
    ca, cb = bench.compact_line(a), bench.compact_line(b)
    for key in ("metric", "unit", "dtype", "scaling", "data", "higher_is_better", "config", "n_gpus"):
        assert ca[key] == cb[key], key
    defaults, explicit = bench.parse_args([]), bench.parse_args(["--gpus", "1"])
    assert vars(defaults) == vars(explicit) and defaults.gpus == 1


def test_line_carries_median_placement_and_the_cpp_operator_chain():
    import bench
    line = full_line()
    line["ms_per_step_median_placement"] = 0.4321
    line["config"]["output_placement"] = {"candidates": 12, "join_ms_per_candidate": [0.33] * 12, "chosen": 3, "by": "hy_result_pool_calibrate"}
    line["cpp_operator_chain_ms"] = {"scan_join_aggregate_device_resident": 2.5, "scan_join_device_resident": 1.8, "scan_join_host_result": 536.0,
                                     "join_orders_lineitem_device_resident": 0.64, "join_orders_lineitem_host_result": 374.0, "scan_device_resident": 0.43,
                                     "scan_host_result": 117.0, "rows": {"orders": 1}, "pool_candidates": 6, "pool_chosen": 0, "ok": True}
    parsed = check(bench.compact_line(line))
    assert parsed["ms_per_step_median_placement"] == 0.4321
    assert parsed["legs"]["cpp_operator_chain_ms"]["join_orders_lineitem_device_resident"] == 0.64 and "hy_result_pool_calibrate" in parsed["config"]["workload"]
