"""Diagnostic: what the per-stage HIP events cost inside the timed frame loop (4K fp32)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv = ["bench.py"]
import bench
from svgf_amd import filter as F
W, H = 3840, 2160
dev = torch.device("cuda:0")
gb, rads = bench.make_inputs(W, H, "f32", dev)
d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
for k in range(20):
    d.Render(rads[k % 4], gb, gb)
for rnd in range(3):
    for timing in (False, True):
        d.timing_enable(timing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(100):
            d.Render(rads[k % 4], gb, gb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100 * 1e3
        if timing:
            d.timing_read()
        print(f"stage events {'on ' if timing else 'off'}: {dt:.4f} ms/frame")
