#!/bin/bash
# Kernel trace of tools/aggregate_bench.py for the in-tree library and every variant named: tools/agg_ab.sh [variant ...]  (through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/agg_ab
rm -rf $OUT && mkdir -p $OUT
cd /tmp
for v in current "$@"; do
  if [ $v = current ]; then unset HY_LIBRARY; else export HY_LIBRARY=$R/hyrise_amd/variants/lib_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o agg -- python $R/tools/aggregate_bench.py > $OUT/$v.log 2>&1
  echo "== $v" >> $OUT/summary.txt
  grep 'aggregate ms' $OUT/$v.log | tail -2 >> $OUT/summary.txt
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  python - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) >= 1.0: print("   %-60s calls %4s  avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/$v
done
cat $OUT/summary.txt
