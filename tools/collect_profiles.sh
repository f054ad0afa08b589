#!/bin/bash
# Collects this round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   scan       python bench.py (scan leg only)         -- dominant kernel scan_slices
#   join       python tools/join_bench.py              -- SF10 orders x lineitem, every kernel of one hy_join_hash
#   aggregate  python tools/aggregate_bench.py         -- TPC-H Q1 core as SURVEY.md 8(d) specifies it
#   fused      python tools/fused_bench.py             -- TPC-H Q6 / Q1: the operator chains beside hy_scan_project_aggregate (kernel trace only)
# For each: one --kernel-trace --stats run, then FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc runs (the TCC has four counter
# slots: FETCH_SIZE takes 3, WRITE_SIZE 2; MI355X_MICROARCH.md "rocprofv3 PMC slots").  --pmc never shares a run with tracing.
# tools/summarize_round.py turns the CSVs into gpurun_out/round/r02_*; copy those into profiles/.
# usage: tools/collect_profiles.sh [scan] [join] [aggregate] [fused]      (default: the first three)
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HY_TPCH_CACHE=/tmp/tpch_cache
OUT=$R/gpurun_out/round
rm -rf $OUT && mkdir -p $OUT
cd /tmp
legs=${@:-scan join aggregate}
for leg in $legs; do
  case $leg in
    scan) cmd="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-join --no-aggregate --no-cases --no-ssb --no-multi" ;;
    join) cmd="python $R/tools/join_bench.py" ;;
    aggregate) cmd="python $R/tools/aggregate_bench.py" ;;
    fused) cmd="python $R/tools/fused_bench.py" ;;
    *) echo "unknown leg $leg"; exit 1 ;;
  esac
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$leg/trace -o $leg -- $cmd > $OUT/$leg.trace.log 2>&1 || tail -3 $OUT/$leg.trace.log
  if [ $leg != fused ]; then
    timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$leg/fetch -o $leg -- $cmd > $OUT/$leg.fetch.log 2>&1 || tail -3 $OUT/$leg.fetch.log
    timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$leg/write -o $leg -- $cmd > $OUT/$leg.write.log 2>&1 || tail -3 $OUT/$leg.write.log
  fi
  grep -h "join ms\|aggregate ms\|\"metric\"\|fused\|chain" $OUT/$leg.trace.log | tail -4 | cut -c1-300
done
python $R/tools/summarize_round.py $OUT $legs
# the raw rocprofv3 trees are large; only the summaries travel back
for leg in $legs; do rm -rf $OUT/$leg; done
