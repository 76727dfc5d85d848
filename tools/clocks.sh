#!/bin/bash
# Shader clock and socket power while the bench runs (the à-trous launches are 70 % of a frame): tools/clocks.sh <tag> [env assignments...]
#   tools/clocks.sh product
#   tools/clocks.sh stream SVGF_LIBRARY=$PWD/build/libsvgf_DG.so SVGF_ATROUS_MODE=1     (diagnostic build: streaming only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r2
TAG=$1; shift
( env "$@" python3 bench.py --steps 30000 --warmup 10 --no-cpu --no-extra --motion static > gpurun_out/r2/clk_bench_$TAG.log 2>&1 & )
sleep 12
for i in $(seq 1 8); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz)/sclk \1 MHz/' -e 's/.*Power (W): \(.*\)/power \1 W/' | paste - -
  sleep 1
done > gpurun_out/r2/clocks_$TAG.log
sleep 14
python3 - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r2/clk_bench_$TAG.log") if l.startswith("{")][-1])
    print("$TAG: ms/frame", d["ms_per_step"], "stages", [round(v["ms"], 4) for v in d["stages"].values()])
except Exception as e:
    print("$TAG: no bench line", e)
PY
cat gpurun_out/r2/clocks_$TAG.log
