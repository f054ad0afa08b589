mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_scan_gpu.py -m gpu -q -x -k "bit_packed" > gpurun_out/gputest_small.log 2>&1; tail -5 gpurun_out/gputest_small.log
