"""Diagnostic: what an idle gap on the host side costs the frames that follow it.  50-frame windows of the 4K fp32 frame, each preceded by
`gap` ms of sleep after a device synchronisation (the device idles): ms per frame of the window, and of its first and last ten frames."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F

W, H = (1920, 1080) if "1080p" in argv else (3840, 2160)
dev = torch.device("cuda:0")
scene = bench.Scene(W, H, dev, pool=2)
pool = bench.FramePool(scene, "f32", "static")
d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
d.set_prev_guide(True)
n = 0
for _ in range(60):
    d.Render(*pool.frame(n)); n += 1
def run(k):
    global n
    t0 = time.perf_counter()
    for _ in range(k):
        d.Render(*pool.frame(n)); n += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k
for gap in (0, 0, 2, 5, 10, 20, 40, 100, 0, 0):
    torch.cuda.synchronize()
    time.sleep(gap * 1e-3)
    a = run(10); b = run(30); c = run(10)
    print(f"idle {gap:4d} ms -> first 10 frames {a:.4f}, next 30 {b:.4f}, last 10 {c:.4f} ms per frame")
