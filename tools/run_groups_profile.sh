#!/bin/bash
# rocprofv3 kernel trace of AggregateHash with many groups (tools/aggregate_groups_bench.py, NGROUPS=...); via gpurun.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/gprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o groups -- python $R/tools/aggregate_groups_bench.py > $OUT/log.txt 2>&1
grep "groups" $OUT/log.txt
python $R/tools/kernel_stats.py $OUT 14
rm -rf $OUT/*/  # raw traces stay on the box
