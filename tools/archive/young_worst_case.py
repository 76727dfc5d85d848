"""Diagnostic: the worst case of the young-pixel machinery — a frame in which EVERY wave holds some young pixels (but not 64): the two G-buffers the
frames alternate between differ in the normals of every `period`-th column, so those columns fail the reprojection test in every frame, stay at
history 1, and every wave of the temporal launch appends to the young list (one atomic per wave on one counter) while the moments launch walks a list
of W x H / period pixels.  Prints the stage times next to the static scene's.
    python3 tools/archive/young_worst_case.py [period ...]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F
W, H = 3840, 2160
dev = torch.device("cuda:0")
scene = bench.Scene(W, H, dev, pool=2)
ADAPTIVE = "--no-adaptive" not in argv
periods = [int(a) for a in argv[1:] if not a.startswith("--")] or [0, 64, 8, 2]
for period in periods:
    pool = bench.FramePool(scene, "f32", "static")
    if period:
        n = pool.gb[1].normal                      # uint16 [H, W, 4]: half bits {nx, ny, nz, matID}
        n.view(torch.int16)[:, ::period, 0:3] ^= -32768   # flip the sign of the normal in every `period`-th column of ONE of the two G-buffers
    d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
    d.set_adaptive_moments(ADAPTIVE)
    for k in range(60):
        d.Render(*pool.frame(k))
    torch.cuda.synchronize()
    d.timing_enable(1)
    for k in range(60, 100):
        d.Render(*pool.frame(k))
    torch.cuda.synchronize()
    ms, nfr = d.timing_read()
    hist = d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())
    young = float((hist < 4).float().mean().item())
    print(f"adaptive {int(ADAPTIVE)} (streaming kernel: {int(d.adaptive_moments_state())}); every {period or 'no':>3} column mismatching: young fraction {young:.4f}; frame {sum(ms) / nfr:.4f} ms: temporal {ms[0] / nfr:.4f}, moments {ms[1] / nfr:.4f}, "
          "a-trous " + " ".join(f"{m / nfr:.4f}" for m in ms[2:]), flush=True)
    d.close()
