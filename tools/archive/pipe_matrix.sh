#!/bin/bash
# one frame at a time against two frames in flight (svgf_set_frames_in_flight) on the bench's other workloads, interleaved on one box
one() { python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extra "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
for f in 1 2; do one --workload 1080p --frames-in-flight $f; one --storage f16 --frames-in-flight $f; one --workload 8k --frames-in-flight $f;  one --motion pan --frames-in-flight $f; done; done
