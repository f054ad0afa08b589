#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's scan workload on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats  -> per-kernel durations (profiles/scan_kernel_stats.csv)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC has 4 slots: FETCH_SIZE 3 + WRITE_SIZE 2 do not fit
#      one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots") -> HBM bytes per scan_slices launch (profiles/scan_pmc.json)
# Never combines --pmc with sys/hip/hsa tracing.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o scan -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-join --no-aggregate --no-cases --no-ssb > $OUT/bench_trace.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o scan -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-join --no-aggregate --no-cases --no-ssb > $OUT/bench_fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o scan -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-join --no-aggregate --no-cases --no-ssb > $OUT/bench_write.log 2>&1 || true
find $OUT -name '*.csv' | head -20
python $R/tools/summarize_profile.py $OUT
