import os, sys
sys.path.insert(0, "/root/repo")
import torch
sys.argv = ["bench.py"]
import bench
from svgf_amd import filter as F
W, H = 3840, 2160
dev = torch.device("cuda:0")
scene = bench.Scene(W, H, dev)
pool = bench.FramePool(scene, "f32", "pan")
d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
d.set_prev_guide(True)
for k in range(40):
    d.Render(*pool.frame(k)); torch.cuda.synchronize()
    if k % 2 == 0 or k < 8:
        hist = d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())
        print(k, d.adaptive_moments_state(), d.adaptive_moments_sample(), int((hist < 4).sum().item()))
