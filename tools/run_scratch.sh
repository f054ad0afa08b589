mkdir -p gpurun_out
timeout 600 python tools/join_placement.py 8 absolute 2>&1 | grep -v amdgpu.ids > gpurun_out/join_placement_c.txt; cut -c1-100 gpurun_out/join_placement_c.txt
