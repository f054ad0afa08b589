mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_small_gpu.py tests/test_fused_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > gpurun_out/gputest_small.log 2>&1; echo "rc=$?" >> gpurun_out/gputest_small.log
tail -4 gpurun_out/gputest_small.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/q1prof
rm -rf $OUT && mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o q1 -- python $R/tools/q1_fused_time.py 10 > $OUT/log.txt 2>&1)
(grep "fused_\|groups" $OUT/log.txt; python $R/tools/kernel_stats.py $OUT 8) > gpurun_out/q1_fused_profile.txt 2>&1
rm -rf $OUT
tail -12 gpurun_out/q1_fused_profile.txt
