#!/bin/bash
# Parts of the small-domain kernels switched off (HY_AGG_SMALL_DEBUG on a -DHY_DEBUG_SWITCHES build: tools/build_variant.sh dbg "aggregate.hip aggregate_wide.hip" -DHY_DEBUG_SWITCHES): 1 no histograms, 2 no 2-byte column
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HY_LIBRARY=$R/hyrise_amd/variants/lib_dbg.so
OUT=$R/gpurun_out/agg_dbg; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for d in ${PARTS:-0 1 8 16 32 57}; do
  export HY_AGG_SMALL_DEBUG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$d -o agg -- python $R/tools/aggregate_bench.py > $OUT/d$d.log 2>&1
  f=$(find $OUT/d$d -name "*kernel_stats.csv" | head -1)
  echo "== debug $d" >> $OUT/summary.txt
  python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'sd_' in r['Name']: print('   %-30s avg %8.1f us' % (r['Name'].split('(')[0][-30:], float(r['AverageNs'])/1e3))" >> $OUT/summary.txt
  rm -rf $OUT/d$d
done
cat $OUT/summary.txt
