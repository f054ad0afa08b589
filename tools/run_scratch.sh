mkdir -p gpurun_out
timeout 600 python tools/join_placement.py 10 2>&1 | grep -v amdgpu.ids > gpurun_out/join_placement.txt; cat gpurun_out/join_placement.txt
