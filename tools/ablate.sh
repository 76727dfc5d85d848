#!/bin/bash
# Ablation of the LDS a-trous kernel on the GPU box: full / streaming-only / arithmetic-only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')"
for mode in 0 1 2; do
  echo "== SVGF_ATROUS_MODE=$mode"
  SVGF_LIBRARY=$R/build/libsvgf_diag.so SVGF_ATROUS_MODE=$mode python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extra "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print({k:v['ms'] for k,v in d['stages'].items()}, d['ms_per_step'])"
done
