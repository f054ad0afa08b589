#!/bin/bash
# SQ counters of the kernels of one tool run; through gpurun from the repo root: tools/sq_counters.sh <name> <kernel substring[,substring...]> <python script + args>
# --pmc only (never with tracing), two passes -> gpurun_out/sq_<name>.txt
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HY_TPCH_CACHE=/tmp/tpch_cache
NAME=$1; KERNELS=$2; shift 2
OUT=$R/gpurun_out/sq_$NAME
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -o run -- python $R/"$@" > $OUT/a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/b -o run -- python $R/"$@" > $OUT/b.log 2>&1
python - > $R/gpurun_out/sq_$NAME.txt <<PY
import csv, glob, collections
wanted = "$KERNELS".split(",")
for d in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("hy::", "")[:40]
            if not any(x in k for x in wanted): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in sorted(acc.items()):
            print(k)
            for c, x in sorted(v.items()): print("   %-24s %.4g per launch" % (c, x / max(1, n[(k, c)])))
PY
cat $R/gpurun_out/sq_$NAME.txt
rm -rf $OUT
