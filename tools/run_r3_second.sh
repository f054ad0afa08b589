set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_join_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/join_bench.py 10 2>&1 | tee gpurun_out/r3/join_bench2.txt | tail -12
HY_JOIN_TRACE=1 timeout 300 python tools/join_bench.py 3 2>&1 | grep "trace" | tee gpurun_out/r3/join_trace2.txt
HY_LIBRARY=$PWD/hyrise_amd/libhyrise_amd_tile4096.so timeout 300 python tools/join_bench.py 10 2>&1 | tee gpurun_out/r3/join_bench2_tile4096.txt | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3/jprof -o join -- python $GRAFT_REPO_ROOT/tools/join_bench.py 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/r3/jprof | head
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r3/jprof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['Percentage'])
PY
