#!/bin/bash
# After `gpurun -- tools/final_round.sh <tag>`: copy what was measured into profiles/ (run here, in the container), then
# `python3 tools/profiles_readme.py <tag>` regenerates that round's section of profiles/README.md FROM the copied files.
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$TAG
RN=$(echo $TAG | sed 's/^r0*//')
cp $O/hbm_traffic.json profiles/hbm_traffic.json          # made on the box from that call's PMC passes (tools/final_round.sh), before its bench lines
for t in 4k_f32 4k_f16 1080p_f32; do cp $O/rocprofv3_summary_$t.txt profiles/${TAG}_rocprofv3_summary_$t.txt; done
# the raw per-kernel stats of THE SAME trace run the 4K fp32 summary was condensed from
cp "$(ls gpurun_out/prof_${TAG}_4k_f32/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_kernel_stats_4k_f32.csv
grep -h "^{" gpurun_out/prof_${TAG}_4k_f32/bench_trace.log > profiles/${TAG}_bench_under_rocprofv3_4k_f32.json
for f in bench_4k_f32 bench_4k_f16 bench_1080p_f32 bench_8k_f32 bench_4k_f32_pair_launch bench_4k_f32_two_in_flight bench_8k_f32_stripdriver_1gpu; do cp $O/$f.json profiles/${TAG}_$f.json; done
cp $O/strip_sim.txt profiles/${TAG}_strip_sim_8k_over_8.txt
[ -f $O/cold_frames.txt ] && cp $O/cold_frames.txt profiles/${TAG}_cold_frames.txt
cp $O/pytest_gpu.log profiles/${TAG}_pytest_gpu.txt
[ -f $O/fuzz_parity.txt ] && { echo "# tests/fuzz_parity.py on MI355X in the round's final call (tools/final_round.sh): summary lines"; grep -E "^FAIL|^fuzz_parity:" $O/fuzz_parity.txt; } >> profiles/${TAG}_fuzz_parity.txt
cp gpurun_out/parity_report.json profiles/${TAG}_parity_report.json
python3 tools/profiles_readme.py $TAG
