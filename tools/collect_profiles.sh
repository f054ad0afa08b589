#!/bin/bash
# Collects this round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root) over bench.py ITSELF: one
# --kernel-trace --stats run (the JSON line that process prints is kept beside its trace: HIP-event kernel_ms and the traced
# durations come from ONE process), then FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc runs (the TCC has four counter slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2; MI355X_MICROARCH.md "rocprofv3 PMC slots").  --pmc never shares a run with tracing.
# tools/summarize_bench_profile.py turns the CSVs into gpurun_out/round/r06_bench_*; copy those into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/round
rm -rf $OUT && mkdir -p $OUT
cd /tmp
cmd="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cases --no-ssb --no-multi --details /tmp/bench_details_profiled.json"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $cmd > $OUT/r06_bench_traced.json 2> $OUT/trace.log || tail -3 $OUT/trace.log
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $cmd > $OUT/fetch.json 2> $OUT/fetch.log || tail -3 $OUT/fetch.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $cmd > $OUT/write.json 2> $OUT/write.log || tail -3 $OUT/write.log
python $R/tools/summarize_bench_profile.py $OUT "$1"
# the raw rocprofv3 trees are large; only the summaries travel back
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/fetch.json $OUT/write.json
