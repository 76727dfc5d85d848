"""A strip frame as the device ran it: rocprofv3 --kernel-trace of tools/strip_sim.py, one frame printed as a timeline (kernel, start and end in
microseconds from the frame's temporal launch, queue) and the means over the traced frames: per kernel, and per GAP between consecutive filter
kernels — where an exchange that is not hidden shows up.
    on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_strip -- python3 $R/tools/strip_sim.py --plans per-iteration --no-whole --rounds 1 --steps 60 --warm-frames 200 --warm-ms 100
    anywhere:        python tools/strip_trace.py gpurun_out/prof_strip [frames]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    short = name.split("(")[0].split("<")[0].split("::")[-1]
    if "atrous_lds_kernel" in name:
        s = name.split("atrous_lds_kernel<")[1].split(">")[0].split(",")
        short = f"atrous S={s[1].strip()}"
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?"), name))
rows.sort()
# frames: from one temporal_kernel to the next
starts = [i for i, r in enumerate(rows) if r[2].startswith("temporal_kernel")]
starts = starts[-(nframes + 1):]
frames = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
print(f"{f}: {len(frames)} frames")
fr = frames[len(frames) // 2]
t0 = fr[0][0]
print("one frame (us from the temporal launch's start):")
for s, e, short, q, _ in fr:
    print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})  q{q:>3}  {short}")
dur, gap = defaultdict(list), defaultdict(list)
period = []
for fr in frames:
    filt = [r for r in fr if r[2].startswith(("temporal", "moments", "atrous"))]
    for r in fr:
        dur[r[2]].append((r[1] - r[0]) / 1e3)
    for a, b in zip(filt[:-1], filt[1:]):
        gap[f"{a[2]} -> {b[2]}"].append((b[0] - a[1]) / 1e3)
for a, b in zip(frames[:-1], frames[1:]):
    period.append((b[0][0] - a[0][0]) / 1e3)
    last = [r for r in a if r[2].startswith("atrous")][-1]
    gap["last a-trous -> next temporal"].append((b[0][0] - last[1]) / 1e3)
print(f"frame period: {sum(period) / len(period):.1f} us")
print("mean duration (us) x launches per frame:")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {sum(v) / len(v):7.1f} x {len(v) / len(frames):4.1f}  = {sum(v) / len(frames):7.1f}  {k}")
print("mean gap between consecutive filter kernels (us):")
for k, v in gap.items():
    print(f"  {sum(v) / len(v):7.1f}  {k}")
