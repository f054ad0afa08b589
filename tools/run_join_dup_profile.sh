R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$R/gpurun_out/dprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dup -- python $R/tools/join_dup_bench.py > $OUT/log.txt 2>&1
grep "join ms" $OUT/log.txt | tail -2
python $R/tools/kernel_stats.py $OUT 12
