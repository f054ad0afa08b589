mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_aggregate_gpu.py tests/test_ssb_gpu.py -m gpu -x -q > gpurun_out/gputest_small.log 2>&1; echo "rc=$?" >> gpurun_out/gputest_small.log
tail -6 gpurun_out/gputest_small.log
timeout 500 python tools/ssb_spill_ab.py 30 2>&1 | tail -9
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-join --no-ssb > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_quick.json"))
a = d.get("aggregate") or d.get("config", {}).get("aggregate")
def find(x, key):
    if isinstance(x, dict):
        if key in x: return x[key]
        for v in x.values():
            r = find(v, key)
            if r is not None: return r
print("groups_100000", find(d, "groups_100000"))
print("q1 fused", find(d, "q1"))
PY
