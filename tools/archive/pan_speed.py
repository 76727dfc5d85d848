"""Diagnostic: the young-pixel machinery under faster pans than the bench's (-2.5, +1.5) px per frame: stage times, the frame driver's sample and its choice.
    python3 tools/archive/pan_speed.py [scale ...]      (multiples of the bench pan; default 1 2 4 8)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F
W, H = 3840, 2160
dev = torch.device("cuda:0")
for scale in [float(a) for a in argv[1:]] or [1.0, 2.0, 4.0, 8.0]:
    mv = (bench.PAN_MV[0] * scale, bench.PAN_MV[1] * scale)
    scene = bench.Scene(W, H, dev, mv=mv)
    pool = bench.FramePool(scene, "f32", "pan")
    for adaptive in (True, False):
        d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
        d.set_prev_guide(True)
        d.set_adaptive_moments(adaptive)
        for k in range(60):
            d.Render(*pool.frame(k))
        torch.cuda.synchronize()
        d.timing_enable(1)
        for k in range(60, 116):
            d.Render(*pool.frame(k))
        torch.cuda.synchronize()
        ms, n = d.timing_read()
        print(f"pan {mv}: adaptive {int(adaptive)} (streaming kernel now: {int(d.adaptive_moments_state())}, sample {d.adaptive_moments_sample()}): frame {sum(ms) / n:.4f} ms, "
              f"temporal {ms[0] / n:.4f}, moments {ms[1] / n:.4f}", flush=True)
        d.close()
    del pool, scene
