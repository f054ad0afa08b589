#!/bin/bash
# HBM traffic per launch of the kernels of one tool run: --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE runs (never with tracing)
#   tools/kernel_traffic.sh <name> <kernel substring[,substring...]> <python script + args>   -> gpurun_out/traffic_<name>.txt
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HY_TPCH_CACHE=/tmp/tpch_cache
NAME=$1; KERNELS=$2; shift 2
OUT=$R/gpurun_out/traffic_$NAME
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o run -- python $R/"$@" > $OUT/f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o run -- python $R/"$@" > $OUT/w.log 2>&1
python - > $R/gpurun_out/traffic_$NAME.txt <<PY
import csv, glob, collections
print("# per launch: FETCH_SIZE / WRITE_SIZE in KB as counted (gfx950: HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KB, MI355X_MICROARCH.md)")
wanted = "$KERNELS".split(",")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("f", "w"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hy::", "")[:48]
            if not any(x in k for x in wanted): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
import os
if os.environ.get("LIST"):   # every launch in dispatch order (a kernel whose launches differ: TPC-H Q1's four projections)
    rows = collections.defaultdict(dict)
    for d in ("f", "w"):
        for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hy::", "")[:48]
                if any(x in k for x in wanted): rows[(int(r["Dispatch_Id"]), k)][r["Counter_Name"]] = rows[(int(r["Dispatch_Id"]), k)].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    for (i, k), v in sorted(rows.items()):
        print("  launch %5d %-40s read %8.1f MB  written %8.1f MB" % (i, k, 2 * v.get("FETCH_SIZE", 0) * 1024 / 1e6, v.get("WRITE_SIZE", 0) * 1024 / 1e6))
for k, v in sorted(acc.items()):
    per = {c: x / max(1, n[(k, c)]) for c, x in v.items()}
    print("%-48s HBM %8.1f MB  (read %.1f, written %.1f)  launches %d" % (k, (2 * per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * 1024 / 1e6,
          2 * per.get("FETCH_SIZE", 0) * 1024 / 1e6, per.get("WRITE_SIZE", 0) * 1024 / 1e6, n[(k, "FETCH_SIZE")]))
PY
cat $R/gpurun_out/traffic_$NAME.txt
rm -rf $OUT
