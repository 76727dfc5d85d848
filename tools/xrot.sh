#!/bin/bash
# sweep (tile order, groups per XCD) of the a-trous kernel on one device: CFGS="order xm;..." tools/xrot.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')" 2>/dev/null
IFS=';' read -ra CF <<< "${CFGS:-0 1;1 1;1 2;1 4;1 8;0 16}"
for round in 1 2; do for cfg in "${CF[@]}"; do set -- $cfg
  echo -n "xorder $1 xm $2: "
  env SVGF_LIBRARY=$R/build/libsvgf_diag.so SVGF_ATROUS_XORDER=$1 SVGF_ATROUS_XM=$2 python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extra $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
