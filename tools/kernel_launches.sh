#!/bin/bash
# Every launch of the named kernels of one tool run, in order, with its duration (rocprofv3 --kernel-trace):
#   tools/kernel_launches.sh <name> <kernel substring[,substring...]> <python script + args>   -> gpurun_out/launches_<name>.txt
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HY_TPCH_CACHE=/tmp/tpch_cache
NAME=$1; KERNELS=$2; shift 2
OUT=$R/gpurun_out/launches_$NAME
rm -rf $OUT && mkdir -p $OUT
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $R/"$@" > $OUT/log.txt 2>&1)
python - > $R/gpurun_out/launches_$NAME.txt <<PY
import csv, glob
wanted = "$KERNELS".split(",")
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hy::", "")[:48]
        if any(x in k for x in wanted): rows.append((int(r["Start_Timestamp"]), k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for t, k, us in sorted(rows): print("%-48s %9.1f us" % (k, us))
PY
tail -4 $OUT/log.txt; cat $R/gpurun_out/launches_$NAME.txt
rm -rf $OUT
