#!/bin/bash
# Another build of the library for A/B timing in one GPU session: tools/build_variant.sh NAME [-DFLAG ...]
#   -> hyrise_amd/variants/lib_NAME.so (git-ignored; travels with gpurun), loaded with HY_LIBRARY=$PWD/hyrise_amd/variants/lib_NAME.so
# e.g. the three builds behind DESIGN.md 4.7's last figures:
#   tools/build_variant.sh scalar -DHY_FS_SCALAR_STACK -DHY_FS_CACHED_IDS; tools/build_variant.sh packed -DHY_FS_CACHED_IDS; tools/build_variant.sh nt
#   for v in scalar packed nt scalar packed; do HY_LIBRARY=$PWD/hyrise_amd/variants/lib_$v.so python tools/q1_fused_time.py 20; done
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$R/hyrise_amd/variants" /tmp/hy_variant_objs
objs=()
for f in runtime scan join aggregate aggregate_wide projection exchange boundary comm plan; do
  o=/tmp/hy_variant_objs/${f}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wall -Wno-unused-function "$@" "$R/hyrise_amd/csrc/$f.hip" -o "$o" &
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/hyrise_amd/variants/lib_$name.so" "${objs[@]}" -ldl
ls -la "$R/hyrise_amd/variants/lib_$name.so"
