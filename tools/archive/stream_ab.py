"""Diagnostic: does the stream a context enqueues on matter?  The same frames through two contexts of one process — one on the legacy default
stream (what torch.cuda.current_stream() is unless the host says otherwise, and what the reference uses: App.cu never creates a stream), one on
a stream created by the host — in alternating timed windows (sync, K frames, sync), per frame size.
    python3 tools/archive/stream_ab.py [f32|f16]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F
storage = argv[1] if len(argv) > 1 else "f32"
dev = torch.device("cuda:0")


def run(W, H, K):
    scene = bench.Scene(W, H, dev, pool=2)
    pool = bench.FramePool(scene, storage, "static")
    s = torch.cuda.Stream()
    hi = torch.cuda.Stream(priority=-1)
    ctx = {"default": F.Denoiser(W, H, F.Params(storage=storage, steps=5)),
           "created": F.Denoiser(W, H, F.Params(storage=storage, steps=5), stream=s.cuda_stream),
           "created-hi": F.Denoiser(W, H, F.Params(storage=storage, steps=5), stream=hi.cuda_stream)}
    for d in ctx.values():
        d.set_prev_guide(True)
        for n in range(40):
            d.Render(*pool.frame(n))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.4:
        for d in ctx.values():
            for _ in range(10):
                d.Render(*pool.frame(n)); n += 1
        torch.cuda.synchronize()
    res = {k: [] for k in ctx}
    for rep in range(7):
        for name, d in ctx.items():
            for _ in range(20):
                d.Render(*pool.frame(n)); n += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                d.Render(*pool.frame(n)); n += 1
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) * 1e3 / K)
    print(f"{W}x{H} {storage}: " + "   ".join(f"{k} {sorted(v)[len(v) // 2]:.4f} (min {min(v):.4f})" for k, v in res.items()), flush=True)
    for d in ctx.values():
        d.close()


for (W, H, K) in ((1920, 1080, 200), (3840, 2160, 100), (1280, 720, 300)):
    run(W, H, K)
