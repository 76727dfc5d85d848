"""Diagnostic: the cold path sustained — reset, three frames (every pixel young: moments_lds_kernel), again and again — so that the moments launch is
timed at the device's sustained clocks (tools/cold_frames.py syncs after every frame).  Prints the mean stage times of the cold frames."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F
W, H = 3840, 2160
storage = argv[1] if len(argv) > 1 else "f32"
dev = torch.device("cuda:0")
gb, rads = bench.make_inputs(W, H, storage, dev, nframes=2)
gb2 = F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone())
d = F.Denoiser(W, H, F.Params(storage=storage, steps=5))
for rep in range(40):                      # warm
    d.reset_history()
    for k in range(3):
        d.Render(rads[k % 2], gb if k % 2 == 0 else gb2, (gb2 if k % 2 == 0 else gb) if k else None)
torch.cuda.synchronize()
d.timing_enable(1)
for rep in range(60):
    d.reset_history()
    for k in range(3):
        d.Render(rads[k % 2], gb if k % 2 == 0 else gb2, (gb2 if k % 2 == 0 else gb) if k else None)
torch.cuda.synchronize()
ms, n = d.timing_read()
print(f"{storage}: {n} cold frames back to back: mean frame {sum(ms) / n:.4f} ms; temporal {ms[0] / n:.4f}, moments {ms[1] / n:.4f}, a-trous " + " ".join(f"{m / n:.4f}" for m in ms[2:7]))
