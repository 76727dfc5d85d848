"""Diagnostic: per-frame time of the first frames of a sequence (cold history) vs steady state."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv = ["bench.py"]
import bench
from svgf_amd import filter as F
W, H = 3840, 2160
dev = torch.device("cuda:0")
for storage in ("f32", "f16"):
    gb, rads = bench.make_inputs(W, H, storage, dev, nframes=2)
    d = F.Denoiser(W, H, F.Params(storage=storage, steps=5))
    d.Render(rads[0], gb, gb); d.reset_history(); torch.cuda.synchronize()
    d.timing_enable(True)
    for k in range(8):
        d.Render(rads[k % 2], gb, gb if k else None)
        torch.cuda.synchronize()
        ms, fr = d.timing_read()
        print(storage, "frame", k, "total %.3f ms" % sum(ms), " ".join("%.3f" % m for m in ms))
