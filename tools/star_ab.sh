#!/bin/bash
# star_finish variants (tools/build_variant.sh NAME "join.hip" -D...) against the default build: kernel trace per library -> gpurun_out/star_ab/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/star_ab
: > $R/gpurun_out/star_ab/summary.txt
for v in default "$@"; do
for q in 2.1 4.1; do
  OUT=$R/gpurun_out/star_ab/t; rm -rf $OUT; mkdir -p $OUT
  lib=$R/hyrise_amd/libhyrise_amd.so; [ "$v" != default ] && lib=$R/hyrise_amd/variants/lib_$v.so
  (cd /tmp && HY_LIBRARY=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/tools/ssb_star_time.py $q 10 > $OUT/log.txt 2>&1)
  echo "== $v Q$q" >> $R/gpurun_out/star_ab/summary.txt
  python $R/tools/kernel_stats.py $OUT 8 | grep -E 'star_finish\(|star_probe' | cut -c1-130 >> $R/gpurun_out/star_ab/summary.txt
  rm -rf $OUT
  echo "   $(HY_LIBRARY=$lib python $R/tools/ssb_star_time.py $q 20 2>&1 | tail -1 | cut -c1-50)" >> $R/gpurun_out/star_ab/summary.txt
done
done
cat $R/gpurun_out/star_ab/summary.txt
