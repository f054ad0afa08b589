mkdir -p gpurun_out
timeout 50 python -m pytest tests/test_fused_small_gpu.py tests/test_aggregate_gpu.py tests/test_join_gpu.py -m gpu -x -q -k "three_chunks or small or lds or partition" > gpurun_out/gputest_small.log 2>&1; echo "rc=$?" >> gpurun_out/gputest_small.log
tail -4 gpurun_out/gputest_small.log
