#!/bin/bash
# the young-pixel moments launch under the bench's pan for several prebuilt twins: tools/archive/pan_moments_ab.sh "A B" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in $(seq 1 ${2:-2}); do for v in $1; do echo -n "$v: "
  SVGF_LIBRARY=$R/build/libsvgf_$v.so python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extra --motion pan 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d['stages']['temporal+moments']; print('temporal', s['temporal_ms'], 'moments', s['moments_ms'], 'frame', d['ms_per_step'], 'young', d['young_fraction'])"
done; done
