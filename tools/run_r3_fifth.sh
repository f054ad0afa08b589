cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/collect_profiles.sh "$1" 2>&1 | tail -20
python - <<'PY'
import json
d=json.loads(open('gpurun_out/round/r03_bench_traced.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])
for k,v in d['roofline']['kernels'].items(): print(' ', k, round(v['kernel_ms']*1e3,1), 'us', round(v['frac'],3))
print('join', d['join']['ms_per_join'], {k:round(v['kernel_ms']*1e3,1) for k,v in d['join']['roofline']['kernels'].items()})
print('agg', d['aggregate']['ms_per_aggregate'], d['aggregate']['roofline']['dominant_kernel']['kernel_ms'])
PY
