#!/bin/bash
# One GPU-box session of round 5: parity of the touched paths first, then the headline step under A/B switches / variant builds.
# usage (through gpurun, from the repo root): bash tools/r5_ab.sh OUTDIR "variant:switches ..." ; each entry = LIB[:SWITCH=V[,SWITCH=V]][@ENV=V]
OUT=gpurun_out/$1; shift
mkdir -p $OUT
n=0
[ -z "$SKIP_TESTS" ] && timeout 600 python -m pytest tests/test_join_gpu.py tests/test_scan_gpu.py tests/test_aggregate_wide_gpu.py tests/test_full_size_gpu.py tests/test_aggregate_gpu.py -m gpu -x -q > $OUT/gputest.log 2>&1; echo "rc=$?" >> $OUT/gputest.log
tail -5 $OUT/gputest.log
for entry in "$@"; do
  lib=${entry%%[:@]*}
  rest=${entry#$lib}
  switches=""; envs=""
  if [[ "$rest" == :* ]]; then s=${rest#:}; s=${s%%@*}; for x in ${s//,/ }; do switches="$switches --switch $x"; done; fi
  if [[ "$rest" == *@* ]]; then envs=${rest##*@}; fi
  libpath=hyrise_amd/libhyrise_amd.so
  [ "$lib" != "main" ] && libpath=hyrise_amd/variants/lib_$lib.so
  line=$(env HY_LIBRARY=$PWD/$libpath $envs timeout 300 python bench.py --headline-only --steps 40 --warmup 5 --details /tmp/d.json $switches 2>$OUT/err_$lib.log | tail -1)
  echo "$line" > $OUT/line_$n.json; n=$((n+1))
  echo "$entry $(echo "$line" | python tools/ab_line.py 2>&1)" | tee -a $OUT/ab.txt
done
