"""Diagnostic: what concurrency between independent frames can buy.  Two contexts on two HIP streams filter two independent frame
sequences at once; the aggregate rate against one context alone bounds what software pipelining of consecutive frames (temporal
launch of frame f+1 beside iterations 1-4 of frame f) could give."""
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
pools = [bench.FramePool(scene, "f32", "static") for _ in range(2)]
prio = int(argv[1]) if len(argv) > 1 else 0
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=prio)]
ds = [F.Denoiser(W, H, F.Params(storage="f32", steps=5), stream=s.cuda_stream) for s in streams]
def run(active, n=60):
    for k in range(50):
        for i in active: ds[i].Render(*pools[i].frame(k))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        for i in active: ds[i].Render(*pools[i].frame(k))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * len(active)) * 1e3
print(f"one context: {run([0]):.4f} ms/frame; two contexts on two streams: {run([0, 1]):.4f} ms/frame (aggregate); one again: {run([0]):.4f}")
