mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_aggregate_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "small_domain or aggregate or q1" 2>&1 | tail -3
timeout 300 python tools/agg_debug.py 2>&1 | grep -v amdgpu.ids > gpurun_out/agg_debug.log; cat gpurun_out/agg_debug.log
