#!/bin/bash
# rocprofv3 kernel trace of the headline step only (bench.py --headline-only): per-kernel average durations -> gpurun_out/$1/headline_stats.txt
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1/trace
rm -rf $OUT && mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o h -- python $R/bench.py --headline-only --steps 40 --warmup 5 --details /tmp/d.json > $OUT/line.json 2> $OUT/log.txt)
python $R/tools/kernel_stats.py $OUT 14 > $R/gpurun_out/$1/headline_stats.txt 2>&1
tail -1 $OUT/line.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'median placement', d.get('ms_per_step_median_placement'))" >> $R/gpurun_out/$1/headline_stats.txt 2>&1
rm -rf $OUT
cat $R/gpurun_out/$1/headline_stats.txt
