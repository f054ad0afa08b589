R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/q1chain; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q1c -o q -- python $R/tools/q1_chain_time.py 5 > $R/gpurun_out/q1chain/log.txt 2>&1
(grep "Q1 chain" $R/gpurun_out/q1chain/log.txt; python $R/tools/kernel_stats.py /tmp/q1c 25) | tee $R/gpurun_out/q1chain/stats.txt
