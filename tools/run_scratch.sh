mkdir -p gpurun_out
timeout 600 python tools/scan_ab.py 40 2>&1 | grep -v amdgpu.ids > gpurun_out/scan_ab.txt; cat gpurun_out/scan_ab.txt
timeout 600 python tools/join_bench.py 10 2>&1 | grep -v amdgpu.ids > gpurun_out/join_bench.txt; cat gpurun_out/join_bench.txt
