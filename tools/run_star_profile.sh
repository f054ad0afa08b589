#!/bin/bash
# rocprofv3 kernel trace of hy_star_join_aggregate alone, per SSB query -> gpurun_out/$1/star_q*.txt
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$1
for q in 2.1 4.1; do
  OUT=$R/gpurun_out/$1/t$q
  rm -rf $OUT && mkdir -p $OUT
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/tools/ssb_star_time.py $q 10 > $OUT/log.txt 2>&1)
  (tail -1 $OUT/log.txt; python $R/tools/kernel_stats.py $OUT 22) > $R/gpurun_out/$1/star_q$q.txt 2>&1
  rm -rf $OUT
  cut -c1-140 $R/gpurun_out/$1/star_q$q.txt
done
