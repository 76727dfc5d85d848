#!/bin/bash
# Diagnostic: how many workgroup slots the a-trous band sizing should aim for, per frame size.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')" 2>/dev/null
for wl in ${WL:-4k}; do for slots in ${SLOTS:-512 1024 1536 2048 4096}; do
  echo -n "$wl slots $slots: "
  SVGF_LIBRARY=$R/build/libsvgf_diag.so SVGF_ATROUS_SLOTS=$slots python3 bench.py --workload $wl --steps 30 --warmup 3 --no-cpu --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
