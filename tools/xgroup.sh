#!/bin/bash
# sweep of the XCD grouping of the a-trous tile order (diag build; env VAR = SVGF_ATROUS_XM or SVGF_ATROUS_XGROUP), interleaved on one device
# usage: [BENCH_ARGS='--storage f16'] tools/xgroup.sh <workload> <VAR> v1 v2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')" 2>/dev/null
WL=$1; VAR=$2; shift; shift
for round in 1 2; do for xg in "$@"; do
  echo -n "$WL $VAR $xg: "
  env SVGF_LIBRARY=$R/build/libsvgf_diag.so $VAR=$xg python3 bench.py --workload $WL --steps 40 --warmup 5 --no-cpu --no-extra $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
