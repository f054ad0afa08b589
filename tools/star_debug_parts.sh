#!/bin/bash
# star_finish with parts taken out (a -DHY_DEBUG_SWITCHES build; results are wrong then): kernel trace per HY_STAR_DEBUG value -> gpurun_out/star_dbg/summary.txt
# build first, here:  tools/build_variant.sh debug "join.hip" -DHY_DEBUG_SWITCHES
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HY_LIBRARY=$R/hyrise_amd/variants/lib_debug.so
mkdir -p $R/gpurun_out/star_dbg
: > $R/gpurun_out/star_dbg/summary.txt
for q in 2.1 4.1; do
for d in ${@:-0 1 2 4 6}; do
  OUT=$R/gpurun_out/star_dbg/t; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && HY_STAR_DEBUG=$d timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/tools/ssb_star_time.py $q 10 > $OUT/log.txt 2>&1)
  echo "== Q$q debug $d: $(tail -1 $OUT/log.txt | cut -c1-60)" >> $R/gpurun_out/star_dbg/summary.txt
  python $R/tools/kernel_stats.py $OUT 6 | grep -E 'star_finish|star_probe' | cut -c1-130 >> $R/gpurun_out/star_dbg/summary.txt
  rm -rf $OUT
done
done
cat $R/gpurun_out/star_dbg/summary.txt
