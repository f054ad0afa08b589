#!/bin/bash
# One gpurun call (round 5): the aggregate tests (aggregate_small_domain's shapes among them), then the kernel's time on config 4 (SF10 Q1 core)
# from a kernel trace, and the same GROUP BY with MIN / MAX and integer measures added.   usage: bash tools/run_small_domain_check.sh
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/small && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
cd "$R"
timeout 900 python -m pytest tests/test_aggregate_gpu.py tests/test_fused_small_gpu.py tests/test_aggregate_wide_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/small/tests.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sd_trace -- python "$R/tools/aggregate_bench.py" > "$R/gpurun_out/small/bench.txt" 2>&1
python "$R/tools/kernel_stats.py" /tmp/sd_trace 2>/dev/null | head -12 | tee "$R/gpurun_out/small/kernel_stats.txt"
timeout 300 python "$R/tools/aggregate_bench.py" 2>&1 | tail -4 | tee -a "$R/gpurun_out/small/bench.txt"
EXTREMES=1 timeout 300 python "$R/tools/aggregate_bench.py" 2>&1 | tail -4 | tee "$R/gpurun_out/small/bench_extremes.txt"
