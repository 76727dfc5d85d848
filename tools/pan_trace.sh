#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in P2 B P2 B; do
  export SVGF_LIBRARY=$R/build/libsvgf_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pan_$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extra --motion ${MOTION:-pan} --prime-ms 100 --prime-frames 100 > /dev/null 2>&1
  echo "== $v"; python3 - <<P
import csv,glob
f=glob.glob("$R/gpurun_out/pan_$v/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'moments' in n or 'temporal' in n: print(n[:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
P
done
