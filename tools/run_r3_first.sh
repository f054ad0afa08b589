set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_join_gpu.py -x -q 2>&1 | tail -15
timeout 300 python tools/join_bench.py 10 2>&1 | tee gpurun_out/r3/join_bench1.txt | tail -12
timeout 120 tools/hbm_write_bin 2>&1 | tee gpurun_out/r3/hbm_write.txt | tail -40
