#!/bin/bash
# Another build of the library for A/B timing in one GPU session: tools/build_variant.sh NAME "file.hip [file.hip ...]" [-DFLAG ...]
#   -> hyrise_amd/variants/lib_NAME.so (git-ignored; travels with gpurun), loaded with HY_LIBRARY=$PWD/hyrise_amd/variants/lib_NAME.so
# Only the named translation units are compiled with the flags; the others come from build/obj (__graft_entry__.build()'s objects).
# e.g.  tools/build_variant.sh wide2 "aggregate.hip aggregate_wide.hip" -DHY_SD_WIDE_PARTS=2
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; files=$2; shift 2
mkdir -p "$R/hyrise_amd/variants" /tmp/hy_variant_objs
objs=()
for f in runtime scan join aggregate aggregate_wide aggregate_widest projection exchange boundary comm plan result_pool; do
  if [[ " $files " == *" $f.hip "* ]]; then
    o=/tmp/hy_variant_objs/${f}_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wall -Wno-unused-function "$@" "$R/hyrise_amd/csrc/$f.hip" -o "$o" &
  else
    o=$R/build/obj/$f.o
  fi
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/hyrise_amd/variants/lib_$name.so" "${objs[@]}" -ldl
ls -la "$R/hyrise_amd/variants/lib_$name.so"
