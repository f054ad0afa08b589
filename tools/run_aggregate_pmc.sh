#!/bin/bash
# SQ counters of the aggregate kernels (tools/aggregate_bench.py); run through gpurun from the repo root.  --pmc only.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/${1:-apmc}
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq -o agg -- python $R/tools/aggregate_bench.py > $OUT/log1.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -o agg -- python $R/tools/aggregate_bench.py > $OUT/log2.txt 2>&1
python - <<PY
import csv, glob, collections
for d in ("sq", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if "aggregate" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            print(k)
            for c, x in sorted(v.items()): print("   %-24s %.4g per launch" % (c, x / max(1, n[(k, c)])))
PY
