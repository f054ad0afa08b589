#!/bin/bash
# rocprofv3 kernel trace of the general join cases, one process per case -> gpurun_out/$1/join_case_*.txt
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$1
for c in shuffled_probe shuffled_build duplicate_build_x4; do
  OUT=$R/gpurun_out/$1/t_$c
  rm -rf $OUT && mkdir -p $OUT
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o j -- python $R/tools/join_cases.py $c --steps 5 > $OUT/log.txt 2>&1)
  (grep "ms/join" $OUT/log.txt; python $R/tools/kernel_stats.py $OUT 16) > $R/gpurun_out/$1/join_case_$c.txt 2>&1
  rm -rf $OUT
  cut -c1-140 $R/gpurun_out/$1/join_case_$c.txt
done
