#!/bin/bash
# After `gpurun -- tools/final_round.sh <tag>`: copy what was measured into profiles/ (run here, in the container).
TAG=${1:-r02}
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$TAG
python3 tools/make_traffic.py gpurun_out/prof_${TAG}_4k_f32 3840x2160_f32 ${TAG#r0} > /dev/null
python3 tools/make_traffic.py gpurun_out/prof_${TAG}_4k_f16 3840x2160_f16 ${TAG#r0} > /dev/null
python3 tools/make_traffic.py gpurun_out/prof_${TAG}_1080p 1920x1080_f32 ${TAG#r0} > /dev/null
for t in 4k_f32 4k_f16 1080p_f32; do cp $O/rocprofv3_summary_$t.txt profiles/${TAG}_rocprofv3_summary_$t.txt; done
cp "$(ls gpurun_out/prof_${TAG}_4k_f32/trace/runc/*kernel_stats.csv | head -1)" profiles/${TAG}_kernel_stats_4k_f32.csv
grep -h "^{" gpurun_out/prof_${TAG}_4k_f32/bench_trace.log > profiles/${TAG}_bench_under_rocprofv3_4k_f32.json
for f in bench_4k_f32 bench_4k_f16 bench_1080p_f32 bench_8k_f32 bench_8k_f32_stripdriver_1gpu; do cp $O/$f.json profiles/${TAG}_$f.json; done
cp $O/strip_sim.txt profiles/${TAG}_strip_sim_8k_over_8.txt
cp $O/stamps.txt profiles/${TAG}_stamps_4k_f32.txt
cp gpurun_out/parity_report.json profiles/${TAG}_parity_report.json
cat $O/pytest_gpu.log
python3 - <<PY
import json
for f in ("bench_4k_f32","bench_4k_f16","bench_1080p_f32","bench_8k_f32","bench_8k_f32_stripdriver_1gpu"):
    d=json.loads(open("$O/"+f+".json").read().strip().split("\n")[-1]); r=d.get("roofline") or {}
    print(f, d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), d["pass_roofline"].get("frac_of_8TBps"))
d=json.loads([l for l in open("gpurun_out/prof_${TAG}_4k_f32/bench_trace.log") if l.startswith("{")][-1]); print("traced run", d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
grep atrous_lds profiles/${TAG}_rocprofv3_summary_4k_f32.txt | head -5 | cut -c1-130
