#!/bin/bash
# A/B/... any number of PREBUILT libraries (build/libsvgf_<name>.so) in ONE session on ONE device, interleaved rounds:
#   tools/abn.sh "A B C" [rounds] [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
NAMES="$1"; ROUNDS=${2:-2}; shift; shift
for round in $(seq 1 $ROUNDS); do for v in $NAMES; do
  echo -n "$v: "
  SVGF_LIBRARY=$R/build/libsvgf_$v.so python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extra "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print([round(v['ms'],4) for k,v in d['stages'].items()], d['ms_per_step'])"
done; done
