#!/bin/bash
# the whole GPU suite, then the default bench line with the fields a round's verdict looks at (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1800 python3 -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/check_pytest.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/check_pytest.txt | tail -15
( time timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err ) 2>&1 | grep real
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/check_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["config"]["prev_guide"][:40], d["roofline"]["frac"], d["pass_roofline"]["frac_of_8TBps"])
a = d["also"]
print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in a.items()})
print(a.get("3840x2160_f16"))
print(a.get("7680x4320"))
PY
