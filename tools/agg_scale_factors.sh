#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/agg_sf; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for sf in 2.79 5.59 10 20; do
  export SF=$sf
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o agg -- python $R/tools/aggregate_bench.py > $OUT/s.log 2>&1
  f=$(find $OUT/s -name "*kernel_stats.csv" | head -1)
  echo "== SF $sf" >> $OUT/summary.txt
  python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'sd_' in r['Name']: print('   %-30s avg %8.1f us' % (r['Name'].split('(')[0][-30:], float(r['AverageNs'])/1e3))" >> $OUT/summary.txt
  rm -rf $OUT/s
done
cat $OUT/summary.txt
