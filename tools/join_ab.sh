#!/bin/bash
# tools/join_headline.py for the in-tree library and the variants named, alternating, in one GPU session: tools/join_ab.sh variant [variant ...]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for round in 1 2 3; do
  for v in current "$@"; do
    if [ $v = current ]; then unset HY_LIBRARY; else export HY_LIBRARY=$R/hyrise_amd/variants/lib_$v.so; fi
    printf "%-10s " $v; python tools/join_headline.py ${JOINS:-12} 2>&1 | tail -1
  done
done
