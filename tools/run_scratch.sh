mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_scan_gpu.py tests/test_validate_gpu.py tests/test_fused_gpu.py -m gpu -q -x > gpurun_out/gputest_small.log 2>&1; tail -4 gpurun_out/gputest_small.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-join --no-aggregate --no-ssb --no-multi --no-cpu-baseline 2> gpurun_out/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('scan', d['scan']['ms_per_scan'])
for k,v in d['cases'].items(): print(k, round(v['ms_per_step'],4), v.get('kernel_ms'), v.get('kernel_GBps'))
"
