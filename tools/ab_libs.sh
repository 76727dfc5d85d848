#!/bin/bash
# A/B two PREBUILT libraries (build/libsvgf_A.so, build/libsvgf_B.so; e.g. two source versions built before gpurun)
# in ONE session on ONE device, interleaved rounds: tools/ab_libs.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2 3; do for v in A B; do
  echo -n "$v: "
  SVGF_LIBRARY=$R/build/libsvgf_$v.so python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extra "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
