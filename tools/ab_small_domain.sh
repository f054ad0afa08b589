#!/bin/bash
# A/B of aggregate_small_domain builds in one gpurun session (profiles/r05_small_domain_ab.txt): variants built with tools/build_variant.sh
# from other checkouts (lib_sd_orig.so = round 4's kernel, lib_sd_packed.so = its bookkeeping packed) against the in-tree library.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out/small
for round in 1 2; do
for v in sd_orig sd_packed current; do
  if [ $v = current ]; then unset HY_LIBRARY; else export HY_LIBRARY=$PWD/hyrise_amd/variants/lib_$v.so; fi
  [ $v != current ] && [ ! -f "$HY_LIBRARY" ] && continue
  echo "== $v" | tee -a gpurun_out/small/ab.txt
  timeout 200 python tools/aggregate_bench.py 2>&1 | tail -2 | tee -a gpurun_out/small/ab.txt
done
done
