#!/bin/bash
# HBM-side request counters of the join's probe kernels (separate --pmc passes, never with tracing); via gpurun.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/jpmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp
for set in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/$name -o join -- python $R/tools/join_bench.py > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log
done
python - <<PY
import csv, glob, collections
for path in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for (kern, ctr), (v, n) in sorted(acc.items()):
        if "probe" in kern or "materialize" in kern or "directory" in kern:
            print(f"{kern:40s} {ctr:28s} per launch {v / n:14.1f}  launches {n}")
PY
