#!/bin/bash
# Everything the round's profiles/ files come from, in one gpurun call on one box: tools/final_round.sh <tag>
# Order: tests, the rocprofv3 passes (trace; PMC in separate runs), profiles/hbm_traffic.json from those PMC passes (stamped with the hash
# of the kernel sources), THEN the bench lines — so that the lines carry roofline.traffic measured on this box, in this call, on these sources.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/$TAG; mkdir -p $O
RN=$(echo $TAG | sed 's/^r0*//')
python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5 > $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
tools/prof.sh ${TAG}_4k_f32 > $O/prof_4k_f32.log 2>&1; python3 tools/summarize_prof.py gpurun_out/prof_${TAG}_4k_f32 | grep -v "at::native\|rocclr" > $O/rocprofv3_summary_4k_f32.txt
tools/prof.sh ${TAG}_4k_f16 --storage f16 > $O/prof_4k_f16.log 2>&1; python3 tools/summarize_prof.py gpurun_out/prof_${TAG}_4k_f16 | grep -v "at::native\|rocclr" > $O/rocprofv3_summary_4k_f16.txt
tools/prof.sh ${TAG}_1080p --workload 1080p > $O/prof_1080p.log 2>&1; python3 tools/summarize_prof.py gpurun_out/prof_${TAG}_1080p | grep -v "at::native\|rocclr" > $O/rocprofv3_summary_1080p_f32.txt
python3 tools/make_traffic.py gpurun_out/prof_${TAG}_4k_f32 3840x2160_f32 $RN > /dev/null
python3 tools/make_traffic.py gpurun_out/prof_${TAG}_4k_f16 3840x2160_f16 $RN > /dev/null
python3 tools/make_traffic.py gpurun_out/prof_${TAG}_1080p 1920x1080_f32 $RN > /dev/null
cp profiles/hbm_traffic.json $O/hbm_traffic.json
python3 bench.py 2>$O/bench_4k_f32.err | grep "^{" > $O/bench_4k_f32.json
python3 bench.py --storage f16 --no-cpu --no-extra 2>/dev/null | grep "^{" > $O/bench_4k_f16.json
python3 bench.py --workload 1080p --no-cpu --no-extra 2>/dev/null | grep "^{" > $O/bench_1080p_f32.json
python3 bench.py --workload 8k --no-cpu --no-extra --steps 20 2>/dev/null | grep "^{" > $O/bench_8k_f32.json
python3 bench.py --fuse --no-cpu --no-extra 2>/dev/null | grep "^{" > $O/bench_4k_f32_pair_launch.json
python3 bench.py --frames-in-flight 2 --no-cpu --no-extra 2>/dev/null | grep "^{" > $O/bench_4k_f32_two_in_flight.json
python3 bench.py --strips --workload 8k --steps 20 --warmup 3 2>/dev/null | grep "^{" > $O/bench_8k_f32_stripdriver_1gpu.json
python3 tools/strip_sim.py --rounds 3 2>&1 | grep -E "ms/frame|whole|wire|MB per" > $O/strip_sim.txt
python3 tools/cold_frames.py 2>&1 | grep "frame" > $O/cold_frames.txt
# the seeded sweep of the parity / bit-identity claims, LAST (whatever it finds, the measurements above stand): ${FUZZ_MINUTES:-10} minutes
python3 -m tests.fuzz_parity --minutes ${FUZZ_MINUTES:-10} --seed ${FUZZ_SEED:-6100000} --out $O/fuzz_parity_full.txt > $O/fuzz_parity.txt 2>&1
# the raw per-dispatch tables stay on the box (gpurun merges at most 64 MiB back): the summaries, hbm_traffic.json and kernel_stats.csv were made from them above
find gpurun_out/prof_${TAG}_* -name "*kernel_trace.csv" -delete; find gpurun_out/prof_${TAG}_* -name "*counter_collection.csv" -delete
cat $O/pytest_gpu.log; cut -c1-160 $O/bench_4k_f32.json; cat $O/strip_sim.txt; tail -2 $O/fuzz_parity.txt
