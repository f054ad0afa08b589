set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python tools/join_bench.py 10 2>&1 | tee gpurun_out/r3/join_bench5.txt | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench5.json 2> gpurun_out/r3/bench5.err; tail -3 gpurun_out/r3/bench5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench5.json').read().strip().splitlines()[-1])
print(d['metric']); print('value', d['value'], 'ms_per_step', d['ms_per_step'])
r=d['roofline']; print('step frac', r['frac'], 'achieved', r['achieved'])
for k,v in r['kernels'].items(): print(' ', k, round(v['kernel_ms']*1e3,1), 'us', round(v['frac'],3))
print('scan', d['scan']['ms_per_scan'], d['scan']['roofline']['dominant_kernel']['kernel_ms'])
j=d['join']; print('join', j['ms_per_join'], {k:round(v['kernel_ms']*1e3,1) for k,v in j['roofline']['kernels'].items()})
print('semi', {k:v['ms_per_join'] for k,v in j['semi'].items()})
print('cases', {k:v['ms_per_join'] for k,v in j.get('cases',{}).items()})
print('agg', d['aggregate']['ms_per_aggregate'])
print('q6', d['q6']['ms_per_query'], d['q6']['fused']['ms_per_query'], 'q1', d['q1']['chain_ms_per_query'], d['q1']['fused']['ms_per_query'])
print('ssb', d['ssb']['q2.1']['ms'], d['ssb']['q4.1']['ms'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
