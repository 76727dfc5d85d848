"""Diagnostic: what persistent NaN pixels cost.  A static 4K sequence whose radiance holds NaN at N random pixels in EVERY frame (the NaNs
settle in the history, as in the reference): steady-state stage times against the clean sequence."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F
W, H = 3840, 2160
dev = torch.device("cuda:0")
gb, rads = bench.make_inputs(W, H, "f32", dev, nframes=2)
gb2 = F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone())
gen = torch.Generator(device="cpu").manual_seed(7)
for n_nan in (0, 100, 5000, 60000, 200000):
    rr = [r.clone() for r in rads]
    if n_nan:
        idx = torch.randint(0, W * H, (n_nan,), generator=gen).to(dev)
        for r in rr:
            r.view(-1, 4)[idx, 1] = float("nan")
    d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
    for k in range(60):
        d.Render(rr[k % 2], gb if k % 2 == 0 else gb2, (gb2 if k % 2 == 0 else gb) if k else None)
    torch.cuda.synchronize()
    d.timing_enable(1)
    for k in range(60, 160):
        out = d.Render(rr[k % 2], gb if k % 2 == 0 else gb2, gb2 if k % 2 == 0 else gb)
    torch.cuda.synchronize()
    ms, n = d.timing_read()
    nans = int(torch.isnan(out).any(-1).sum().item())
    print(f"{n_nan:7d} NaN radiance texels per frame -> {nans:8d} NaN output pixels; frame {sum(ms) / n:.4f} ms: temporal {ms[0] / n:.4f}, moments {ms[1] / n:.4f}, a-trous " + " ".join(f"{m / n:.4f}" for m in ms[2:7]))
    d.close()
