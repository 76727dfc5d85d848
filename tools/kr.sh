#!/bin/bash
# KR (outputs per thread) A/B, diag build, interleaved on one device (KR = 2 runs the 256-column variant)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')" 2>/dev/null
for round in 1 2 3; do for kr in 1 2; do
  echo -n "KR $kr: "
  env SVGF_LIBRARY=$R/build/libsvgf_diag.so SVGF_ATROUS_KR=$kr python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extra $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
