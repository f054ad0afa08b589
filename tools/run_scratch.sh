mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_small_gpu.py tests/test_fused_gpu.py -m gpu -x -q > gpurun_out/gputest_small.log 2>&1; echo "rc=$?" >> gpurun_out/gputest_small.log
tail -25 gpurun_out/gputest_small.log
timeout 600 python tools/q1_fused_time.py 10 2>&1 | tail -4
