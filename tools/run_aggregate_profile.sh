#!/bin/bash
# rocprofv3 kernel trace of the TPC-H Q1 core aggregate (tools/aggregate_bench.py); run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/aprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o agg -- python $R/tools/aggregate_bench.py > $OUT/log.txt 2>&1
grep "aggregate ms" $OUT/log.txt
python $R/tools/kernel_stats.py $OUT 8
