mkdir -p gpurun_out
timeout 600 python tools/scan_ab.py 40 3 2>&1 | grep -v amdgpu.ids > gpurun_out/scan_ab.txt; grep -v "u16 value ids" gpurun_out/scan_ab.txt
