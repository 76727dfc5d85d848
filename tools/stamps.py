"""Diagnostic: per-phase cycle shares of the LDS a-trous kernel from in-kernel s_memtime stamps (SVGF_DIAG build)."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from svgf_amd import build as b
lib_path = os.path.join(R, "build", "libsvgf_stamps.so")
os.makedirs(os.path.dirname(lib_path), exist_ok=True)
if not (os.environ.get("SVGF_STAMPS_PREBUILT") and os.path.exists(lib_path)):      # (prebuilt here, it travels to the GPU box with the tree)
    b.build_library(extra_flags=["-DSVGF_DIAG", "-DSVGF_STAMPS"] + os.environ.get("SVGF_STAMPS_FLAGS", "").split(), out=lib_path)
os.environ["SVGF_LIBRARY"] = lib_path
import torch
from svgf_amd import filter as F
sys.argv = ["bench.py"]
import bench
lib = F.load_library()
W, H = 3840, 2160
dev = torch.device("cuda:0")
gb, rads = bench.make_inputs(W, H, "f32", dev, nframes=2)
EXTRA = os.environ.get("SVGF_STAMPS_FLAGS", "").split()
d = F.Denoiser(W, H, F.Params(storage="f32", steps=5))
for k in range(10):
    d.Render(rads[k % 2], gb, gb)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib.svgf_diag_stamps(out, 1)
src, dst = d.new_colour(), d.new_colour()
src.copy_(d.Render(rads[0], gb, gb))
names = ["fetch issue", "setup + tap loop", "epilogue + stores", "barrier 1", "wait rows + commit", "barrier 2"]
for step in (1, 4, 16):
    lib.svgf_diag_stamps(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        d.FilterKernel(src, dst, None, gb, step, 1)
    e1.record()
    torch.cuda.synchronize()
    print(f"step {step}: {e0.elapsed_time(e1) / 5:.4f} ms per launch (this instrumented build)")
    lib.svgf_diag_stamps(out, 1)
    if "-DSVGF_WAVE_SPECIALISED=1" in EXTRA:
        cw, lw = max(out[8], 1), max(out[15], 1)
        ctot = sum(out[i] for i in range(4))
        if out[9]:
            print(f"   workgroups resident on the CU when one starts: {out[9] / max(out[8] / 4, 1):.2f} on average")
        print(f"step {step} (wave-specialised): compute waves {out[8]}, steps per wave {out[10] / cw:.1f}, loop ticks per wave {ctot / cw:.0f}, prologue {100.0 * out[6] / max(out[7], 1):.1f} % of the lifetime {out[7] / cw:.0f}")
        for i, nm in enumerate(["setup + taps 0-14", "wait for the refill", "taps 15-24 + epilogue", "stores"]):
            print(f"   {nm:24s} {100.0 * out[i] / max(ctot, 1):5.1f} %   {out[i] / cw:9.0f} ticks/wave")
        ltot = out[4] + out[5] + out[13]
        print(f"   loader waves {out[15]}, lifetime {out[14] / lw:.0f} ticks, loop {ltot / lw:.0f}")
        for v, nm in ((out[4], "wait for compute waves"), (out[5], "wait rows + convert + write"), (out[13], "signal + next requests")):
            print(f"   {nm:28s} {100.0 * v / max(ltot, 1):5.1f} %   {v / lw:9.0f} ticks/wave")
        continue
    tot = sum(out[i] for i in range(6))
    waves = out[8]
    print(f"step {step}: waves {waves}, ticks per wave {tot / max(waves,1):.0f} (s_memtime ticks)")
    print(f"   wave-steps {out[10]}, uniform-normal fast path {100.0 * out[11] / max(out[10], 1):.1f} %, all-sky skipped {100.0 * out[12] / max(out[10], 1):.1f} %")
    for i, n in enumerate(names):
        print(f"   {n:22s} {100.0 * out[i] / tot:5.1f} %   {out[i] / max(waves,1):9.0f} ticks/wave")
    if out[9]:
        print(f"   workgroups resident on the CU when one starts: {out[9] / max(waves / 4, 1):.2f} on average")
    print(f"   prologue {out[6] / max(waves,1):9.0f} ticks/wave = {100.0 * out[6] / max(out[7],1):5.1f} % of the wave lifetime {out[7] / max(waves,1):9.0f}; steps per wave {out[10] / max(waves,1):.1f}")
