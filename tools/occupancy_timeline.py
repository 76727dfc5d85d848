"""Diagnostic: how many workgroups of ONE à-trous launch are alive over the launch's duration (the tail of a launch), from the per-wave
entry / exit times (s_memrealtime, 100 MHz) the -DSVGF_STAMPS build logs.   SVGF_STAMPS_PREBUILT=1 python3 tools/occupancy_timeline.py"""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from svgf_amd import build as b
lib_path = os.path.join(R, "build", "libsvgf_stamps.so")
if not (os.environ.get("SVGF_STAMPS_PREBUILT") and os.path.exists(lib_path)):
    b.build_library(extra_flags=["-DSVGF_DIAG", "-DSVGF_STAMPS"], out=lib_path)
os.environ["SVGF_LIBRARY"] = lib_path
import torch
from svgf_amd import filter as F
sys.argv = ["bench.py"]
import bench
lib = F.load_library()
W, H = 3840, 2160
dev = torch.device("cuda:0")
gb, rads = bench.make_inputs(W, H, "f32", dev, nframes=2)
d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
for k in range(10):
    d.Render(rads[k % 2], gb, gb)
src, dst = d.new_colour(), d.new_colour()
src.copy_(d.Render(rads[0], gb, gb))
out16 = (C.c_ulonglong * 16)()
NW = (1 << 18) * 16
buf = np.zeros(NW, np.uint64)
for step in (1, 4, 16):
    for _ in range(3):
        d.FilterKernel(src, dst, None, gb, step, 1)
    torch.cuda.synchronize()
    lib.svgf_diag_stamps(out16, 1)
    d.FilterKernel(src, dst, None, gb, step, 1)
    torch.cuda.synchronize()
    lib.svgf_diag_stamp_log(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_ulonglong(NW))
    log = buf.reshape(-1, 16)
    used = log[:, 8] > 0
    t0, t1 = log[used, 13].astype(np.int64), log[used, 14].astype(np.int64)
    base = t0.min()
    t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01          # microseconds
    total = t1.max()
    print(f"step {step}: {used.sum()} waves, launch {total:.1f} us from the first wave's entry to the last wave's exit; wave lifetime mean {np.mean(t1 - t0):.1f} us")
    edges = np.linspace(0, total, 21)
    alive = [np.sum((np.minimum(t1, e1) - np.maximum(t0, e0)).clip(0)) / (e1 - e0) for e0, e1 in zip(edges[:-1], edges[1:])]
    print("   waves alive per 5 % slice of the launch (of", 256 * 4 * 5, "slots at S <= 8):", " ".join(f"{a:.0f}" for a in alive))
