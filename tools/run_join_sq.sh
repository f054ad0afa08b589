#!/bin/bash
# SQ counters of the join's kernels (tools/join_bench.py); run through gpurun from the repo root.  --pmc only, two passes.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HY_TPCH_CACHE=/tmp/tpch_cache
OUT=$R/gpurun_out/jsq
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -o join -- python $R/tools/join_bench.py > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/b -o join -- python $R/tools/join_bench.py > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("hy::", "")[:28]
            if not any(x in k for x in ("rt_probe_emit", "rt_stream_count", "rank_table_fill_dense", "dense_key_stats")): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in sorted(acc.items()):
            print(k)
            for c, x in sorted(v.items()): print("   %-24s %.4g per launch" % (c, x / max(1, n[(k, c)])))
PY
rm -rf $OUT/a $OUT/b
