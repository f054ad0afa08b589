#!/bin/bash
# rocprofv3 kernel trace of the SSB star joins (tools/ssb_bench.py); run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/sprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ssb -- python $R/tools/ssb_bench.py > $OUT/log.txt 2>&1
tail -5 $OUT/log.txt
python $R/tools/kernel_stats.py $OUT 24
rm -rf $OUT/*/
