#!/bin/bash
# does the length of the untimed load in front of the windows change the windows?  bench.py with --prime-ms/--prime-frames 0/0, 100/0, 400/600, 1500/2000, interleaved
one() { python3 bench.py --no-cpu --no-extra --prime-ms $1 --prime-frames $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('prime $1 ms / $2 frames:', d['ms_per_step'], d['windows_ms'], d['ms_per_step_without_stage_events'], d['roofline']['frac'])"; }
for r in 1 2; do one 0 0; one 100 0; one 400 600; one 1500 2000; done
