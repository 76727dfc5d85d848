#!/bin/bash
# Ablation / tuning of the LDS a-trous kernel on the GPU box (diagnostic -DSVGF_DIAG build).
#   MODES="0 1 2"  0 = kernel, 1 = streaming only, 2 = arithmetic only;  KRS="1 2" outputs per thread
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')"
for tx in ${KRS:-1}; do
for mode in ${MODES:-0 1 2}; do
  echo "== SVGF_ATROUS_KR=$tx SVGF_ATROUS_MODE=$mode"
  SVGF_LIBRARY=$R/build/libsvgf_diag.so SVGF_ATROUS_KR=$tx SVGF_ATROUS_MODE=$mode python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extra "$@" 2>${DIAGLOG:-/dev/null} | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print({k:v['ms'] for k,v in d['stages'].items()}, d['ms_per_step'])"
done; done
