mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_join_gpu.py -m gpu -q -x -k "primary_key" > gpurun_out/gputest_small.log 2>&1; tail -3 gpurun_out/gputest_small.log
timeout 600 python tools/ssb_bench.py --sf 30 2>&1 | grep -v amdgpu.ids | tail -2
