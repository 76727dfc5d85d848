#!/bin/bash
# A/B two builds of the library in ONE session on ONE device (interleaved rounds): tools/ab.sh "<flagsA>" "<flagsB>" [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p build
FA="$1"; FB="$2"; shift; shift
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags='$FA'.split(), out='$R/build/libsvgf_A.so')
b.build_library(extra_flags='$FB'.split(), out='$R/build/libsvgf_B.so')" 2>/dev/null
for round in 1 2 3; do for v in A B; do
  echo -n "$v: "
  SVGF_LIBRARY=$R/build/libsvgf_$v.so python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extra "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
