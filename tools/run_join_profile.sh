#!/bin/bash
# rocprofv3 kernel trace of the SF10 orders x lineitem join (tools/join_bench.py); run through gpurun from the repo root.
# usage: tools/run_join_profile.sh [output directory under gpurun_out, default jprof]
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/${1:-jprof}
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o join -- python $R/tools/join_bench.py > $OUT/log.txt 2>&1
grep "join ms" $OUT/log.txt
python $R/tools/kernel_stats.py $OUT 18
