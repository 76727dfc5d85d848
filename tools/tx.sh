#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')" 2>/dev/null
for wl in 1080p 4k; do for tx in 256 128; do
  echo -n "$wl TX $tx: "
  SVGF_LIBRARY=$R/build/libsvgf_diag.so SVGF_ATROUS_TX=$tx python3 bench.py --workload $wl --steps 30 --warmup 3 --no-cpu --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
