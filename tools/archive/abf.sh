#!/bin/bash
# Twins of the library that differ in the fused kernel's knobs, interleaved on ONE device: tools/archive/abf.sh "F0 F1 ..." [rounds] [workloads]
# (build/libsvgf_<name>.so, built beforehand with svgf_amd.build.build_library(extra_flags=..., out=...))
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
NAMES="$1"; ROUNDS=${2:-2}; WLS=${3:-4k}
for round in $(seq 1 $ROUNDS); do for v in $NAMES; do
  AB_TAG=$v AB_MODES=${AB_MODES:-fused} SVGF_LIBRARY=$R/build/libsvgf_$v.so python3 tools/archive/ab_fuse.py 1 $WLS 2>/dev/null
done; done
