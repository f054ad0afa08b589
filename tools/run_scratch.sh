mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_join_gpu.py -m gpu -q -x -k "primary_key or pk or hint" 2>&1 | tail -3
timeout 600 python tools/join_bench.py 10 2>&1 | grep -v amdgpu.ids | head -3
