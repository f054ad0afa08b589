#!/bin/bash
# One GPU-box session: the -m gpu suite, the bench line, then bench.py profiled as one process (tools/collect_profiles.sh), the kernel
# traces of the SSB star joins (tools/run_ssb_profile.sh) and of TPC-H Q1 through hy_scan_project_aggregate (tools/q1_fused_time.py).
# Every step under its own timeout.  usage: bash tools/run_round.sh <commit>
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then   # (SKIP_TESTS=1: the suite has passed at this commit in an earlier session)
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/gputest.log
  tail -4 gpurun_out/gputest.log
fi
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
bash tools/collect_profiles.sh "$1"
ls gpurun_out/round
bash tools/run_ssb_profile.sh > gpurun_out/ssb_profile.txt 2>&1
bash tools/run_star_profile.sh round > /dev/null 2>&1   # hy_star_join_aggregate alone, per query: gpurun_out/round/star_q*.txt
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/q1prof
rm -rf $OUT && mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o q1 -- python $R/tools/q1_fused_time.py 10 > $OUT/log.txt 2>&1)
(tail -4 $OUT/log.txt; python $R/tools/kernel_stats.py $OUT 8) > gpurun_out/q1_fused_profile.txt 2>&1
rm -rf $OUT
tail -12 gpurun_out/q1_fused_profile.txt
