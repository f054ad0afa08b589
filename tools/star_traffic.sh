#!/bin/bash
# HBM traffic and SQ counters of the star plan's kernels per SSB query: --pmc passes only (never with tracing), FETCH_SIZE and WRITE_SIZE in
# SEPARATE runs (the TCC's counter slots), then the SQ instruction counters -> gpurun_out/star_traffic.txt.  Through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/star_traffic
rm -rf $OUT && mkdir -p $OUT
cd /tmp
for q in 2.1 4.1; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f$q -o run -- python $R/tools/ssb_star_time.py $q 5 > $OUT/f$q.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w$q -o run -- python $R/tools/ssb_star_time.py $q 5 > $OUT/w$q.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/s$q -o run -- python $R/tools/ssb_star_time.py $q 5 > $OUT/s$q.log 2>&1
done
python - > $R/gpurun_out/star_traffic.txt <<PY
import csv, glob, collections
print("# per launch: FETCH_SIZE / WRITE_SIZE in KB as counted (gfx950: HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KB, MI355X_MICROARCH.md), SQ instruction counts")
for q in ("2.1", "4.1"):
    print("== Q" + q)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for d in ("f", "w", "s"):
        for f in glob.glob("$OUT/%s%s/**/*counter_collection.csv" % (d, q), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hy::", "")[:40]
                if not k.startswith("star_"): continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in sorted(acc.items()):
        per = {c: x / max(1, n[(k, c)]) for c, x in v.items()}
        hbm = (2 * per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * 1024
        print("%-24s HBM %8.1f MB  " % (k, hbm / 1e6) + "  ".join("%s %.4g" % (c, per[c]) for c in sorted(per)))
PY
cat $R/gpurun_out/star_traffic.txt
rm -rf $OUT
