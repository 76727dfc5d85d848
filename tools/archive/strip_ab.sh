#!/bin/bash
# strip_sim for several prebuilt libraries (build/libsvgf_<name>.so), interleaved: tools/archive/strip_ab.sh "A B" [rounds] [strip_sim args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
NAMES="$1"; ROUNDS=${2:-2}; shift; shift
for round in $(seq 1 $ROUNDS); do for v in $NAMES; do
  echo -n "$v: "
  SVGF_LIBRARY=$R/build/libsvgf_$v.so python3 tools/strip_sim.py "$@" 2>&1 | grep -E "ms/frame" | sed 's/.*driver [a-z]*: //'
done; done
