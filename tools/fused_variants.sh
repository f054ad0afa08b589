#!/bin/bash
# A/B builds of the fused kernel's tuning constants (csrc/aggregate.hip: HY_FUSED_ROWS, HY_FUSED_WAVES, HY_FUSED_CELLS, HY_FUSED_LDS_SLOTS).
#   tools/fused_variants.sh build "ROWS=4 WAVES=3" "CELLS=32" ...    here (no GPU): hyrise_amd/libhyrise_amd_<tag>.so per variant
#   tools/fused_variants.sh run                                      on the GPU box (through gpurun): tools/fused_bench.py per variant + the default
# The variant libraries are git-ignored (*.so) and travel with the gpurun snapshot like the default one; delete them afterwards.
R=$(cd "$(dirname "$0")/.." && pwd)
S="runtime scan join aggregate projection exchange boundary"
case $1 in
  build)
    shift
    for variant in "$@"; do
      tag=$(echo "$variant" | tr ' =' '__' | tr 'A-Z' 'a-z')
      flags=""
      for kv in $variant; do flags="$flags -DHY_FUSED_${kv%%=*}=${kv##*=}"; done
      sources=""
      for s in $S; do sources="$sources $R/hyrise_amd/csrc/$s.hip"; done
      echo "building $tag ($flags)"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function $flags -o $R/hyrise_amd/libhyrise_amd_$tag.so $sources || exit 1
    done ;;
  run)
    echo "== default"; python $R/tools/fused_bench.py 2>/dev/null | grep "fused\|chain"
    for lib in $R/hyrise_amd/libhyrise_amd_*.so; do
      [ -e "$lib" ] || continue
      echo "== $(basename $lib)"; HY_LIBRARY=$lib python $R/tools/fused_bench.py 2>/dev/null | grep "fused\|chain"
    done ;;
  *) sed -n 2,6p "$0" ;;
esac
