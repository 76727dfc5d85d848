"""Diagnostic: what pipelining consecutive frames could buy.  Context 0 on stream 0 runs the temporal + moments launches of a frame
over and over, context 1 on stream 1 iterations 1-4 of the wavelet (stage calls on caller-owned planes); each alone, then both at
once.  together ~ max(alone) would mean the HBM-bound launch hides beside the arithmetic-bound ones; together ~ sum means nothing."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F

W, H = 3840, 2160
dev = torch.device("cuda:0")
scene = bench.Scene(W, H, dev, pool=2)
pool = bench.FramePool(scene, "f32", "static")
prio = int(argv[1]) if len(argv) > 1 else 0
streams = [torch.cuda.Stream(device=dev, priority=prio), torch.cuda.Stream(device=dev)]
ds = [F.Denoiser(W, H, F.Params(storage="f32", steps=5), stream=s.cuda_stream) for s in streams]
rad, gc, gp = pool.frame(0)
d0, d1 = ds
col = [d0.new_colour() for _ in range(4)]
mom = [d0.new_moments() for _ in range(2)]
his = [d0.new_history() for _ in range(2)]
his[0].fill_(8)
fb = [d1.new_colour() for _ in range(3)]
fb[0].copy_(rad if rad.dtype == fb[0].dtype else rad.to(fb[0].dtype)); fb[0][..., 3] = 0.01

def temporal(k):
    d0.TemporalMoments(col[0], rad, col[1], col[2], gc, gp, his[k & 1], his[(k & 1) ^ 1], mom[(k & 1) ^ 1], mom[k & 1])
def wavelet(k):
    pp = 0
    for i in range(1, 5):
        d1.FilterKernel(fb[pp], fb[1 - pp], fb[2], gc, 1 << i, i); pp ^= 1

def run(fns, n=60):
    for k in range(20):
        for f in fns: f(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        for f in fns: f(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a, b = run([temporal]), run([wavelet])
c = run([temporal, wavelet])
print(f"temporal+moments alone {a:.4f} ms; iterations 1-4 alone {b:.4f} ms; sum {a + b:.4f}; both streams at once {c:.4f} ms per pair (priority of the temporal stream {prio})")
