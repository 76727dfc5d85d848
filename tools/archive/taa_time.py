"""Diagnostic: svgf_taa at 4K / 1080p, ms per launch (HIP events around 200 launches), both storages."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from svgf_amd import filter as F
for W, H in ((3840, 2160), (1920, 1080)):
    for storage in ("f32", "f16"):
        d = F.Denoiser(W, H, F.Params(storage=storage, steps=5))
        dt = d.colour_dtype()
        filt = torch.rand((H, W, 4), device="cuda", dtype=torch.float32).to(dt)
        hist = torch.rand((H, W, 4), device="cuda", dtype=torch.float32).to(dt)
        out = torch.empty_like(filt)
        for _ in range(50):
            d.TAA(filt, hist, out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            d.TAA(filt, hist, out)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 200
        bpp = 48 if storage == "f32" else 24
        print(f"{W}x{H} {storage}: {ms:.4f} ms per launch = {W * H * bpp / ms / 1e9:.2f} TB/s of the {bpp} B/px the stage touches", flush=True)
        d.close()
