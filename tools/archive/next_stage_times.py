"""Diagnostic: device time of the stages either side of the path: svgf_taa and svgf_pack_gbuffer at 4K."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from svgf_amd import filter as F
W, H = 3840, 2160
dev = torch.device("cuda:0")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for storage in ("f32", "f16"):
    d = F.Denoiser(W, H, F.Params(storage=storage))
    a, b, c = d.new_colour(), d.new_colour(), d.new_colour()
    a.uniform_(0, 1); b.uniform_(0, 1)
    ms = timed(lambda: d.TAA(a, b, c))
    T = 4 if storage == "f32" else 2
    print(f"svgf_taa 4K {storage}: {ms:.4f} ms = {W * H * 12 * T / ms / 1e6:.0f} GB/s of 3 planes x {4 * T} B/px")
d = F.Denoiser(W, H, F.Params(storage="f32"))
pos = torch.rand((H, W, 4), device=dev); nrm = torch.rand((H, W, 4), device=dev); bary = torch.rand((H, W, 4), device=dev)
eye = [1.0 if i % 5 == 0 else 0.0 for i in range(16)]
ms = timed(lambda: d.PackGBuffer(pos, nrm, bary, eye, eye, (0.0, 0.0, 5.0)))
print(f"svgf_pack_gbuffer 4K (incl. 3 torch.empty): {ms:.4f} ms = {W * H * (48 + 32) / ms / 1e6:.0f} GB/s of 48 B/px in + 32 B/px out")
