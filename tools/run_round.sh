#!/bin/bash
# One GPU-box session: the -m gpu suite, the bench line, then bench.py profiled as one process (tools/collect_profiles.sh).
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/gputest.log
tail -4 gpurun_out/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
bash tools/collect_profiles.sh "$1"
ls gpurun_out/round
