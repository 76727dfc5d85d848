#!/bin/bash
# rocprofv3 kernel durations of the temporal and moments launches for prebuilt twins (build/libsvgf_<name>.so), bench pan by default:
#   tools/archive/pan_trace.sh "A B" [static|pan]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  export SVGF_LIBRARY=$R/build/libsvgf_$v.so
  rm -rf $R/gpurun_out/pan_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pan_$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extra --motion ${2:-pan} --prime-ms 100 --prime-frames 100 > /dev/null 2>&1
  echo "== $v"; python3 - <<P
import csv,glob
f=glob.glob("$R/gpurun_out/pan_$v/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'moments' in n or 'temporal' in n: print(n[:72], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
P
done
