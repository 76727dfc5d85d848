#!/usr/bin/env python3
"""bench.py — Mpixels/s of the full SVGF pass (temporal + moments + 5 à-trous iterations) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one frame through the hot path (svgf_denoise_frame, or the strip driver for N > 1) with all inputs
resident in HBM.  N = 1: 3840x2160 fp32 (BASELINE.json configs[2], the configuration the metric's roofline target is
quoted on).  N > 1: one 7680x4320 fp32 frame cut into N row strips with halo exchange over RCCL (configs[3]); strong
scaling; without WORLD_SIZE in the environment the N rank processes are started from here.  One JSON line on stdout.

Inputs.  The current and the previous G-buffer are DISTINCT device planes, ping-ponged frame by frame as the reference
binds two framebuffers (src/App.cu:471-474) — also with a static camera, where their contents are equal.  Two motions
are timed: "static" (the headline, SURVEY.md 8d steady state) and "pan", a camera pan of mv = (-2.5, +1.5) pixels per
frame (prev - cur, GBuffer.frag:67-69) over a pool of consecutive frames walked forth and back.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# Algorithmic bytes per pixel: every plane a stage touches counted once.
ALG_BYTES = {
    "f32": dict(temporal=130, moments=33, atrous_iter=56, atrous_feedback=16, full5=459),
    "f16": dict(temporal=98, moments=17, atrous_iter=40, atrous_feedback=8, full5=323),
}
# Bytes the fused frame driver really moves per pixel in steady state, every plane it touches counted once (no cache
# credit): the temporal launch reads radiance, both G-buffers (3 planes each), previous colour / moments / history and
# writes history, moments and the filter buffer (the moments stage's copy, Filter.cuh:521, is folded into it and the
# temporal colour itself is stored only where it is read again); it also writes the 16-byte guide texel
# {depth, ddepth, normal, instance ID} the iterations read instead of the 16 + 8 byte motion / normal texels — and, one frame
# later, the reprojection test reads instead of the 32 bytes of the previous G-buffer's three planes (the frames the bench hands
# over ping-pong between two G-buffers, so the previous one is always the one the guide was made from; tests/test_bench_inputs.py).
# An iteration reads colour and the guide and writes colour (+ the feedback colour in iteration 0).
MOVED_BYTES = {
    "f32": dict(temporal_moments=16 + 32 + 16 + 16 + 8 + 1 + 1 + 8 + 16 + 16, atrous_iter=16 + 16 + 16, atrous_feedback=16),
    "f16": dict(temporal_moments=8 + 32 + 16 + 8 + 4 + 1 + 1 + 4 + 8 + 16, atrous_iter=8 + 16 + 8, atrous_feedback=8),
}
HBM_PEAK_GBPS = 8000.0        # MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
WORKLOADS = {"1080p": (1920, 1080), "4k": (3840, 2160), "8k": (7680, 4320)}
PRIME_FRAMES = 40             # history must reach steady state (h >= 4) before anything is timed (§8d); 40 rather than 8 frames also
                              # bring the device to its sustained clocks: with 8, a 5-step timed region read 0.80 instead of 0.73 ms
PAN_MV = (-2.5, 1.5)          # the pan that exercises the truncation of the reprojected coordinate
PAN_POOL = 8                  # consecutive frames of the pan held in HBM (walked 0..7..0..)
STRIP_PAN_MV = (1.5, -3.5)    # N > 1: a pan that reaches 4 rows per frame, so that moments and history rows really travel in the state exchange
METRIC = "Mpixels/s (and ms/frame) for full SVGF temporal+5 a-trous pass at 1080p/4K"
SCENE_NOTE = {
    "planar": "svgf_amd/synth.py scene 'planar' (SURVEY.md 8d): tilted ground plane, three quads, one sphere, 8 % sky - piecewise CONSTANT normals except on the sphere",
    "curved": "svgf_amd/synth.py scene 'curved': rolling terrain, a large sphere, an upright cylinder, 8 % sky - smooth-shaded, normalize(FragNormal) per texel (GBuffer.frag:65): "
              "99.7 % of the surface texels differ from their left neighbour's normal bits; analytic depth / ddepth; static camera, same 1-spp noise model",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=list(WORKLOADS), default=None)
    ap.add_argument("--storage", choices=["f32", "f16"], default="f32")
    ap.add_argument("--iters", type=int, default=5, help="à-trous iterations (BASELINE: 5)")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--halo-plan", default="auto")
    ap.add_argument("--motion", choices=["static", "pan", "both"], default="both", help="N = 1: which camera motions are timed (value = static)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (pan, 1080p, fp16, cold frames)")
    ap.add_argument("--strips", action="store_true", help="run the strip driver even at N=1 (exercises the N>1 code path)")
    ap.add_argument("--leg-timeout", type=float, default=240.0, help="N > 1: seconds one leg (a plan, the pan, the one-GPU frame) may take before the job ends with what is measured")
    ap.add_argument("--no-one-gpu", action="store_true", help="N > 1: skip the whole frame on rank 0's GPU alone (one_gpu_ms / speedup_vs_one_gpu)")
    ap.add_argument("--fuse", action="store_true", help="iterations 0 and 1 as one launch (svgf_atrous_pair): bit-identical, measured slower (DESIGN.md 3.3c)")
    ap.add_argument("--frames-in-flight", type=int, choices=[1, 2], default=1,
                    help="2: svgf_set_frames_in_flight(2) - iterations 1.. of a frame on a side stream beside the next frame's temporal launch (bit-identical results; "
                         "a frame's result is ordered on the stream one call later)")
    ap.add_argument("--prev-guide", action="store_true", help="svgf_set_prev_guide(1) for the headline run (an opt-in of the ABI: the host vouches that the previous "
                                                             "G-buffer's planes are last frame's current ones, untouched; the reprojection test then reads the kept 16-B guide "
                                                             "plane instead of three planes, 32 B/px).  Default: off, the ABI's default; the opt-in is reported as also.prev_guide_on")
    ap.add_argument("--prime-ms", type=float, default=400.0, help="untimed load before the first timed frame: at least this long ...")
    ap.add_argument("--prime-frames", type=int, default=600, help="... and at least this many frames, --warmup included (DESIGN.md 6: the post-idle clock ramp and the "
                                                                   "one-off stall of a process's first ~4 000 stream operations belong in front of the timed region; 0 / 0 for smoke runs)")
    ap.add_argument("--windows", type=int, default=5, help="the --steps-frame timed window is repeated this many times; ms_per_step is the median window")
    return ap.parse_args()


def alg_bytes_full(storage, iters):
    b = ALG_BYTES[storage]
    return b["temporal"] + b["moments"] + iters * b["atrous_iter"] + (b["atrous_feedback"] if iters > 0 else 0)


def moved_temporal(storage, prev_guide):
    """MOVED_BYTES counts 16 B/px at the reprojected address: the kept guide plane (svgf_set_prev_guide).  The ABI's default reads the previous
    G-buffer's three planes there: 32 B/px."""
    return MOVED_BYTES[storage]["temporal_moments"] + (0 if prev_guide else 16)


def moved_bytes_full(storage, iters, fused=False, prev_guide=False):
    b = MOVED_BYTES[storage]
    if fused and iters >= 2:        # the pair launch reads colour + guide once and writes the feedback plane and iteration 1's result
        return moved_temporal(storage, prev_guide) + (iters - 2) * b["atrous_iter"] + 4 * (16 if storage == "f32" else 8)
    return moved_temporal(storage, prev_guide) + iters * b["atrous_iter"] + (b["atrous_feedback"] if iters > 0 else 0)


# ------------------------------------------------------------------ inputs ---------------------
def make_inputs(W, H, storage, device, nframes=4, row_begin=0, row_end=None):
    """Static camera, rows [row_begin,row_end): ONE G-buffer and a pool of `nframes` independent 1-spp radiance frames in
    HBM (the diagnostic tools and the strip runs use this; the strip runs upload the G-buffer twice)."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import synth
    npdt = np.float32 if storage == "f32" else np.float16
    sc = synth.make_scene(W, H, 0, row_begin=row_begin, row_end=row_end)
    gb = F.GBuffer(*(torch.from_numpy(sc[k]).to(device) for k in ("motion", "normal", "uv")))
    rads = [torch.from_numpy(synth.make_radiance(sc["base"], W, k, row_begin=row_begin).astype(npdt)).to(device)
            for k in range(nframes)]
    return gb, rads


class Scene:
    """Device-resident synthetic inputs of one frame size for both motions.  Two canvases a little larger than the frame are
    generated once on the host (svgf_amd/synth.py; even and odd frames of the pan sit half a pixel apart) and every frame
    of the pan is a window of one of them, copied into tight planes of its own on the device."""

    def __init__(self, W, H, device, pool=PAN_POOL, mv=PAN_MV, nmasks=4, scene="planar"):
        import torch
        from svgf_amd import synth
        assert pool >= 2 and pool % 2 == 0 and all(float(2 * m).is_integer() for m in mv)
        self.W, self.H, self.device, self.pool, self.mv, self.kind = W, H, device, pool, mv, scene
        k = pool // 2 - 1
        sx, sy = int(round(2 * mv[0])), int(round(2 * mv[1]))              # shift of the window per two frames
        x0, x1 = min(0, sx * k), W + max(0, sx * k)
        y0, y1 = min(0, sy * k), H + max(0, sy * k)
        self.origin, self.shift = (x0, y0), (sx, sy)
        self.canvas = []
        for parity in (0, 1):
            sc = synth.make_scene(W, H, parity, mv=mv, row_begin=y0, row_end=y1, col_begin=x0, col_end=x1, scene=scene)
            self.canvas.append({n: torch.from_numpy(sc[n]).to(device) for n in ("motion", "normal", "uv", "base")})
        ys, xs = np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64)
        self.hit = [torch.from_numpy(synth.uniform01(synth.SEED, f + 1, ys, xs, 0) < np.float32(0.25)).to(device) for f in range(nmasks)]

    def window(self, f):
        """Planes of frame f of the pan (f = 0 is also the static frame): dict of tight device tensors."""
        c = self.canvas[f & 1]
        k = f // 2
        xa, ya = k * self.shift[0] - self.origin[0], k * self.shift[1] - self.origin[1]
        return {n: t[ya:ya + self.H, xa:xa + self.W].contiguous() for n, t in c.items()}

    def radiance(self, base, mask, storage):
        """1-spp style radiance {r,g,b,1}: a path either finds the light (p = 1/4, carrying 4x the radiance) or returns black."""
        import torch
        out = torch.ones((self.H, self.W, 4), dtype=torch.float32, device=self.device)
        out[..., :3] = torch.where(self.hit[mask][..., None], base * 4.0, torch.zeros_like(base)).clamp_(0.0, 1.0)
        return out if storage == "f32" else out.to(torch.float16)


class UpsampledScene:
    """A static 2W x 2H scene made on the device from a W x H Scene: the same geometry sampled at twice the rate (every G-buffer texel becomes a 2 x 2
    block, the depth derivative per pixel halves), fresh 1-spp noise per pixel.  For `also.7680x4320` of the default run: synth.make_scene takes
    ~30 s of host time per 8K canvas, this takes none.  (`bench.py --workload 8k` times the 8K scene of svgf_amd/synth.py itself.)"""

    def __init__(self, scene, nmasks=2):
        import torch
        self.W, self.H, self.device = 2 * scene.W, 2 * scene.H, scene.device
        w = scene.window(0)
        up = lambda t: t.repeat_interleave(2, 0).repeat_interleave(2, 1).contiguous()      # noqa: E731
        up16 = lambda t: up(t.view(torch.int16)).view(torch.uint16)                       # noqa: E731 (no uint16 gather in torch)
        mo = up(w["motion"])
        mo[..., :2] = 0.0
        mo[..., 3] *= 0.5
        self._w = {"motion": mo, "normal": up16(w["normal"]), "uv": up16(w["uv"]), "base": up(w["base"])}
        g = torch.Generator(device=self.device)
        g.manual_seed(0x53564746)
        self.hit = [torch.rand((self.H, self.W), generator=g, device=self.device) < 0.25 for _ in range(nmasks)]

    def window(self, f):
        return {n: t.clone() if n == "motion" else t for n, t in self._w.items()}

    radiance = Scene.radiance


class FramePool:
    """frame(n) -> (radiance, cur G-buffer, prev G-buffer) of the n-th frame of a sequence; everything lives in HBM."""

    def __init__(self, scene: Scene, storage, motion):
        from svgf_amd import filter as F
        self.motion = motion
        if motion == "static":
            w = scene.window(0)
            w["motion"][..., :2] = 0.0
            # the same contents in two sets of planes: frame n reads set n & 1 as current and the other one as previous
            self.gb = [F.GBuffer(w["motion"].clone(), w["normal"].clone(), w["uv"].clone()) for _ in range(2)]
            self.rad = [scene.radiance(w["base"], m, storage) for m in range(len(scene.hit))]
        else:
            self.P = scene.pool
            self.fwd, self.bwd, self.rad = [], [], []
            for f in range(self.P):
                w = scene.window(f)
                back = w["motion"].clone()
                back[..., :2] *= -1.0                     # walking the pool backwards, frame f's predecessor is frame f+1
                self.fwd.append(F.GBuffer(w["motion"], w["normal"], w["uv"]))
                self.bwd.append(F.GBuffer(back, w["normal"], w["uv"]))
                self.rad.append(scene.radiance(w["base"], f % len(scene.hit), storage))
            self.prev = None

    def frame(self, n):
        if self.motion == "static":
            return self.rad[n % len(self.rad)], self.gb[n & 1], self.gb[(n & 1) ^ 1]
        period = 2 * (self.P - 1)
        m = n % period
        idx, forward = (m, True) if m < self.P else (period - m, False)
        if m == 0 and n > 0:
            forward = False
        cur = (self.fwd if forward else self.bwd)[idx]
        prev = self.prev if n > 0 and self.prev is not None else cur
        self.prev = cur
        return self.rad[idx], cur, prev


# ------------------------------------------------------------------ single GPU -----------------
def run_single(pool: FramePool, W, H, storage, iters, variant, steps, warmup, device, cold_frames=0, fuse=False, windows=5, in_flight=1, prime=(400.0, 600), prev_guide=False, adaptive=True,
               path_stats=False):
    """-> dict(ms_per_step = median over `windows` timed windows of `steps` frames each (sync, K frames, sync), windows_ms, stage_ms[list],
    ms_no_events: one more window without the per-stage HIP events, ...)."""
    import torch
    from svgf_amd import filter as F
    d = F.Denoiser(W, H, F.Params(storage=storage, steps=iters, variant=variant), device=device.index or 0)
    d.set_iteration_fusion(fuse)
    d.set_frames_in_flight(in_flight)
    d.set_adaptive_moments(adaptive)
    d.set_prev_guide(prev_guide)   # default off = the ABI's default.  (On: the pools do hand over last frame's current G-buffer, untouched, as `prev` — tests/test_bench_inputs.py — the precondition of svgf_set_prev_guide)
    n = 0
    for _ in range(PRIME_FRAMES + warmup):
        d.Render(*pool.frame(n))
        n += 1

    def rewarm(ms=60.0, frames=0):
        """untimed frames right before a timed window whenever the host has just kept the device idle (timing_read, a garbage collection):
        after >= 2 ms of idleness the part needs tens of milliseconds to be back at its clocks (tools/archive/idle_gap.py: the next 10-40 frames run
        5-20 % slower).  The first call also runs `frames` frames: a process sees ONE stall of 20-65 ms when it has enqueued its first
        ~4 000 stream operations (tools/strip_sim.py --per-frame) - it belongs in front of the timed windows, not in one of them."""
        nonlocal n
        t0, k = time.perf_counter(), 0
        while (time.perf_counter() - t0) * 1e3 < ms or k < frames:
            for _ in range(10):
                d.Render(*pool.frame(n))
                n += 1
            k += 10
            torch.cuda.synchronize(device)

    def window():
        nonlocal n
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            d.Render(*pool.frame(n))
            n += 1
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) * 1e3 / steps
    import gc
    gc.collect()
    gc.disable()               # no generation-2 collection of the interpreter (20-30 ms with torch loaded) inside a window: it would starve the launch queue
    rewarm(prime[0], max(0, prime[1] - PRIME_FRAMES - warmup))
    d.timing_enable(4)         # HIP events between the stages of every 4th timed frame, on the stream the kernels are launched on
    win = [window() for _ in range(max(1, windows))]
    stage_ms, frames = d.timing_read()
    d.timing_enable(False)
    rewarm()
    no_events = window()       # the same window without any stage event: what the events cost the timed frames
    gc.enable()
    out = d.Render(*pool.frame(n))
    d.flush()                  # (two frames in flight: the result is ordered on the stream by the next call, or by this)
    assert bool(torch.isfinite(out.float()).all()), "non-finite output"
    d.set_frames_in_flight(1)
    hist = d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())
    young = float((hist < 4).float().mean().item())
    # which tap path the a-trous waves took (svgf_path_stats_enable, include/svgf_ext.h): counted on 8 more frames, after everything timed
    share = None
    if path_stats and variant != "direct":
        d.path_stats_enable(True)
        for _ in range(8):
            d.Render(*pool.frame(n))
            n += 1
        st = d.path_stats_read()
        d.path_stats_enable(False)
        share = {str(1 << i): (round(st[1 << i][1] / st[1 << i][0], 4) if st[1 << i][0] else None) for i in range(min(iters, F.PATH_STAT_STEPS))}
        tot = [sum(st[1 << i][k] for i in range(min(iters, F.PATH_STAT_STEPS))) for k in (0, 1)]
        share["all"] = round(tot[1] / tot[0], 4) if tot[0] else None
    cold = []
    if cold_frames:                          # §8d: cold frames (history < 4: the 7x7 moments estimate runs everywhere) reported apart
        d.reset_history()
        torch.cuda.synchronize(device)
        d.timing_enable(True)
        for k in range(cold_frames):
            rad, cur, prev = pool.frame(k)
            d.Render(rad, cur, prev if k else None)
            ms_k, _ = d.timing_read()        # synchronises
            cold.append(round(sum(ms_k), 4))
        d.timing_enable(False)
    d.close()
    srt = sorted(win)
    return dict(ms_per_step=srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2]), windows_ms=win, ms_no_events=no_events,
                stage_ms=[m / max(frames, 1) for m in stage_ms], frames=frames, cold_ms=cold, young_fraction=young, fused=bool(fuse and iters >= 2 and variant != "direct"),
                in_flight=in_flight, uniform_path_share=share)


def run_interactive(pool: FramePool, W, H, storage, iters, variant, device, producer_target_ms=4.0, frames=48):
    """The regime the reference runs in: ONE denoise per displayed frame between other work of the same GPU — application::Render path-traces
    the frame (App.cu:550) and then calls the three filter methods on the same stream (App.cu:552-556).  Neither the sustained headline
    (back-to-back frames) nor an idle device between frames is that.  Two figures, both the sum of the library's per-stage HIP events:
      interleaved: producer, denoise, producer, denoise ... on one stream, where the producer is a memory-bound stand-in for the path tracer
                   (in-place passes over a 1 GiB buffer, their number calibrated to ~producer_target_ms);
      isolated:    denoise, host sleep of 5 ms (the device idles and leaves its sustained clocks), denoise ...
    -> dict(interleaved_ms, producer_ms, isolated_ms, ...)."""
    import torch
    from svgf_amd import filter as F
    d = F.Denoiser(W, H, F.Params(storage=storage, steps=iters, variant=variant), device=device.index or 0)
    d.set_prev_guide(True)
    n = 0
    for _ in range(PRIME_FRAMES):
        d.Render(*pool.frame(n)); n += 1
    buf = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=device).fill_(1.0)

    def producer(passes):
        for _ in range(passes):
            buf.mul_(1.0000001)                      # 2 GiB of HBM traffic per pass
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    producer(4); torch.cuda.synchronize(device)
    e0.record(); producer(8); e1.record(); torch.cuda.synchronize(device)
    per_pass = e0.elapsed_time(e1) / 8
    passes = max(1, int(round(producer_target_ms / per_pass)))
    # warm: the same alternation, untimed
    for _ in range(16):
        producer(passes); d.Render(*pool.frame(n)); n += 1
    torch.cuda.synchronize(device)
    d.timing_enable(1)
    pe = []
    for k in range(frames):
        if k % 8 == 0:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); producer(passes); b.record(); pe.append((a, b))
        else:
            producer(passes)
        d.Render(*pool.frame(n)); n += 1
    torch.cuda.synchronize(device)
    ms, fr = d.timing_read()
    interleaved = sum(ms) / max(fr, 1)
    stages_i = [m / max(fr, 1) for m in ms]
    producer_ms = sum(x.elapsed_time(y) for x, y in pe) / len(pe)
    iso = []
    for k in range(24):
        d.Render(*pool.frame(n)); n += 1
        torch.cuda.synchronize(device)
        ms_k, _ = d.timing_read()
        if k >= 4:
            iso.append(sum(ms_k))
        time.sleep(0.005)
    d.timing_enable(False)
    d.close()
    del buf
    iso.sort()
    return {"interleaved_ms": round(interleaved, 4), "producer_ms": round(producer_ms, 3), "producer": f"{passes} in-place passes over a 1 GiB fp32 buffer on the same stream "
            f"({per_pass:.3f} ms each: memory-bound, as the path tracer of App.cu:550 is)", "interleaved_stage_ms": [round(x, 4) for x in stages_i],
            "isolated_ms": round(iso[len(iso) // 2], 4), "isolated_ms_min": round(iso[0], 4), "isolated_ms_max": round(iso[-1], 4),
            "isolated": "one frame at a time, 5 ms of host sleep between frames (the device idles): median / min / max of 20 frames",
            "note": "sum of the stage times between the library's HIP events (the denoise alone, whatever surrounds it); the headline ms_per_step is sustained "
                    "back-to-back throughput, which an interactive renderer does not see"}


def run_graph_replay(pool: FramePool, W, H, storage, iters, variant, steps, device, prime_ms=300.0, windows=3):
    """svgf_denoise_frame under stream capture (include/svgf.h "Stream capture"): the static pool's period of four frames recorded ONCE
    into a hipGraph and replayed, against the same frames enqueued call by call on the same stream — with one frame in flight and with two
    (where the graph's cross-stream edges replace the event waits of the calls).  -> dict of ms per frame (median of `windows` windows)."""
    import torch
    from svgf_amd import filter as F
    assert pool.motion == "static"
    s = torch.cuda.Stream(device)
    steps = max(4, (steps + 3) // 4 * 4)
    out = {}
    for in_flight in (1, 2):
        d = F.Denoiser(W, H, F.Params(storage=storage, steps=iters, variant=variant), device=device.index or 0, stream=s.cuda_stream)
        d.set_prev_guide(True)
        d.set_frames_in_flight(in_flight)

        def four():
            for n in range(4):
                d.Render(*pool.frame(n))
            d.flush()
        with torch.cuda.stream(s):
            for _ in range(PRIME_FRAMES // 4):
                four()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                four()

            def timed(fn):
                t0 = time.perf_counter()
                while (time.perf_counter() - t0) * 1e3 < prime_ms:       # the device at its sustained clocks (run_single.rewarm)
                    for _ in range(8):
                        fn()
                    s.synchronize()
                w = []
                for _ in range(windows):
                    s.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps // 4):
                        fn()
                    s.synchronize()
                    w.append((time.perf_counter() - t0) * 1e3 / steps)
                return round(sorted(w)[len(w) // 2], 4)
            out[f"calls_{in_flight}_in_flight_ms"] = timed(four)
            out[f"graph_{in_flight}_in_flight_ms"] = timed(g.replay)
        del g
        d.close()
    out["note"] = ("four frames (the static pool's period) captured once on the context's stream and replayed; `calls`: the same four frames enqueued call by call; "
                   "results bit-identical (tests/test_gpu_graph.py); svgf_flush ends each group of four")
    return out


def timing_fields(r):
    """What the judge asked to see next to ms_per_step: the spread of the windows, the sum of the stage times (events make the frames that
    carry them slower, so it exceeds ms_per_step), and what the events cost."""
    w = r["windows_ms"]
    return {"ms_per_step_min": round(min(w), 4), "ms_per_step_max": round(max(w), 4), "windows": len(w), "windows_ms": [round(x, 4) for x in w],
            "stage_sum_ms": round(sum(r["stage_ms"]), 4), "ms_per_step_without_stage_events": round(r["ms_no_events"], 4),
            "event_overhead_ms_per_step": round(r["ms_per_step"] - r["ms_no_events"], 4),
            "timing_note": "ms_per_step = median of `windows` windows of --steps frames each (sync, K frames, sync), every 4th frame carrying 2 + iterations + 1 "
                           "HIP events (the stage times); stage_sum_ms is the mean of those frames' stage times; one more window without events gives the overhead"}


KERNEL_SOURCES = ("svgf_kernels.hip", "svgf_api.hip", "svgf_kernels.h", "svgf_ctx.h", "svgf_device.h", "svgf_atrous_taps.h", "svgf_atrous_lds.h", "svgf_atrous_fused.h", "svgf_moments_lds.h")


def kernel_source_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "svgf_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(W, H, storage, kernel):
    """-> (HBM bytes per launch, source) from the rocprofv3 PMC passes of tools/prof.sh as recorded in profiles/hbm_traffic.json — used only
    if that file was produced from THESE kernel sources (it records their hash); otherwise (None, None).  The value is NOT measured in
    this run: the source string says where it comes from."""
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(tfile))
        sha = kernel_source_sha()
        if rec.get("kernel_source_sha16") != sha:
            return None, None
        v = rec.get(f"{W}x{H}_{storage}", {}).get(kernel)
        return (v, f"profiles/hbm_traffic.json@{sha}") if v is not None else (None, None)
    except Exception:  # noqa: BLE001
        return None, None


def roofline_block(W, H, storage, iters, stage_ms, variant="auto", fused=False, prev_guide=False):
    """Roofline of the dominant kernel (the LDS-streaming a-trous kernel: `iters` launches per frame, or iters - 2 next to the pair launch)
    + per-stage table."""
    b, mv = ALG_BYTES[storage], MOVED_BYTES[storage]
    P = W * H
    first = 2 if fused else 0                       # with the fusion on, timing slot 2 holds the pair and slot 3 the gap between two events
    at_ms = stage_ms[2 + first:2 + iters]
    if not at_ms or sum(at_ms) <= 0:
        return None, {}
    n_l = len(at_ms)
    bytes_per_launch = (n_l * b["atrous_iter"] + (0 if fused else b["atrous_feedback"])) * P / n_l
    avg_ms = sum(at_ms) / n_l
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    kname = "atrous_direct_kernel" if variant == "direct" else "atrous_lds_kernel"
    traffic, src = measured_traffic(W, H, storage, "atrous_bytes_per_launch") if variant != "direct" and not fused else (None, None)
    roof = {"bound": "hbm", "kernel": kname, "launches_per_step": n_l, "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(avg_ms, 5),
            "traffic": traffic, "traffic_source": src}
    if not fused:
        # BASELINE.md section 2's own line for the iterations alone ("a-trous x5 only": 296 B/px at 0.60 x 8 TB/s = 0.511 ms at 3840x2160 fp32)
        roof.update(atrous_only_target(W, H, storage, iters, sum(at_ms)))
    # the launch's second bound: its vector ALUs (the kernel is co-limited: DESIGN.md 3.3).  Like `traffic`, from the PMC passes of the same sources.
    busy, bsrc = measured_traffic(W, H, storage, "atrous_valu_busy") if variant != "direct" and not fused else (None, None)
    if busy is not None:
        ipp, _ = measured_traffic(W, H, storage, "atrous_insts_valu_per_px")
        roof["secondary"] = {"bound": "valu", "valu_busy": busy, "insts_valu_per_px": ipp, "source": bsrc,
                             "what": "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) per launch; insts_valu_per_px = SQ_INSTS_VALU / (pixels / 64)"}

    def rate(px_bytes, ms):
        return round(px_bytes * P / (ms * 1e-3) / 1e9, 1) if ms > 0 else None
    tm = stage_ms[0] + stage_ms[1]
    tt, _ = measured_traffic(W, H, storage, "temporal_bytes_per_launch")
    stages = {"temporal+moments": {"ms": round(tm, 5), "temporal_ms": round(stage_ms[0], 5), "moments_ms": round(stage_ms[1], 5),
                                   "algorithmic_B_per_px": b["temporal"] + b["moments"], "moved_B_per_px": moved_temporal(storage, prev_guide),
                                   "algorithmic_equivalent_GBps": rate(b["temporal"] + b["moments"], tm), "moved_GBps": rate(moved_temporal(storage, prev_guide), tm),
                                   "measured_hbm_bytes": tt}}
    if fused:
        px, mpx = 2 * b["atrous_iter"] + b["atrous_feedback"], 4 * (16 if storage == "f32" else 8)      # colour + guide in, feedback + result out
        ms01 = stage_ms[2] + stage_ms[3]
        stages["atrous_steps1+2_one_launch"] = {"ms": round(ms01, 5), "kernel": "atrous_fused12_kernel", "algorithmic_B_per_px": px, "moved_B_per_px": mpx,
                                                "algorithmic_equivalent_GBps": rate(px, ms01), "moved_GBps": rate(mpx, ms01)}
    for i in range(first, iters):
        px = b["atrous_iter"] + (b["atrous_feedback"] if i == 0 else 0)
        mpx = mv["atrous_iter"] + (mv["atrous_feedback"] if i == 0 else 0)
        stages[f"atrous_step{1 << i}"] = {"ms": round(stage_ms[2 + i], 5), "algorithmic_B_per_px": px, "moved_B_per_px": mpx,
                                          "algorithmic_equivalent_GBps": rate(px, stage_ms[2 + i]), "moved_GBps": rate(mpx, stage_ms[2 + i])}
    return roof, stages


def atrous_only_target(W, H, storage, iters, atrous_ms):
    """The sum of the a-trous launches of a frame against 60 % of the HBM roofline on their algorithmic bytes ('a-trous x5 only')."""
    b = ALG_BYTES[storage]
    target = (iters * b["atrous_iter"] + b["atrous_feedback"]) * W * H / (0.60 * HBM_PEAK_GBPS * 1e9) * 1e3
    return {f"atrous_x{iters}_ms": round(atrous_ms, 4), f"atrous_x{iters}_target_ms": round(target, 3), f"atrous_x{iters}_target_met": bool(atrous_ms <= round(target, 3))}


def pass_block(W, H, storage, iters, ms, fused=False, prev_guide=False):
    alg, mov = alg_bytes_full(storage, iters), moved_bytes_full(storage, iters, fused, prev_guide)
    g = lambda bpp: bpp * W * H / (ms * 1e-3) / 1e9   # noqa: E731
    return {"algorithmic_bytes_per_px": alg, "algorithmic_equivalent_GBps": round(g(alg), 1), "frac_of_8TBps": round(g(alg) / HBM_PEAK_GBPS, 4),
            "frac_of_6.29TBps_copy": round(g(alg) / 6290.0, 4),
            "moved_bytes_per_px": mov, "moved_GBps": round(g(mov), 1), "moved_frac_of_8TBps": round(g(mov) / HBM_PEAK_GBPS, 4),
            "note": "frac_of_8TBps rates the SURVEY 8(d) algorithmic bytes (the contract's figure, 459 B/px fp32) against the frame time; the fused "
                    "driver moves fewer bytes (moved_bytes_per_px: no separate moments copy, temporal colour stored only where it is read again), "
                    "so moved_frac_of_8TBps is the share of the HBM peak the frame really uses"}


def usable_cpus():
    """Threads this process may really run on: the affinity mask and the cgroup's CPU quota, not the box's os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(storage, iters):
    """The scalar C++ oracle (kind 'port': the reference has no CPU path) timed on the host cores, on a bounded
    sample of the same workload: 1920x1080 synthetic frames (BASELINE configs[1], a quarter of the 4K frame), steady
    state, on every thread the process may use (usable_cpus: the GPU boxes show 256 CPUs and grant 16) — 960x540 where that is
    fewer than 32, to stay within ~10 s."""
    from oracle import oracle as orc
    from svgf_amd import synth
    cores = usable_cpus()
    W, H = (1920, 1080) if cores >= 32 else (960, 540)
    fr = [synth.make_frame(W, H, k) for k in range(2)]
    gb = {k: fr[0][k] for k in ("motion", "normal", "uv")}
    pipe = orc.Pipeline(W, H, storage, steps=iters, nthreads=cores)
    for k in range(4):                       # reach h >= 4
        pipe.frame(fr[k % 2]["radiance"], gb, gb)
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 8.0 and n < 40):
        pipe.frame(fr[n % 2]["radiance"], gb, gb)
        n += 1
    dt = time.perf_counter() - t0
    # (i) of SURVEY 8(d): one thread, 320x180, one warm + one timed frame
    w1, h1 = 320, 180
    f1 = synth.make_frame(w1, h1, 0)
    g1 = {k: f1[k] for k in ("motion", "normal", "uv")}
    one = orc.Pipeline(w1, h1, storage, steps=iters, nthreads=1)
    for k in range(5):
        one.frame(f1["radiance"], g1, g1)
    t1 = time.perf_counter()
    one.frame(f1["radiance"], g1, g1)
    one_mpx = w1 * h1 / (time.perf_counter() - t1) / 1e6
    # BASELINE.json configs[0]: 256x256 synthetic G-buffer + noisy radiance, ONE a-trous iteration through the scalar C++ loop, one thread
    w0 = h0 = 256
    f0 = synth.make_frame(w0, h0, 0)
    g0 = {k: f0[k] for k in ("motion", "normal", "uv")}
    npdt = np.float32 if storage == "f32" else np.float16
    src0, dst0, fb0 = f0["radiance"].astype(npdt), np.zeros((h0, w0, 4), npdt), np.zeros((h0, w0, 4), npdt)
    orc.atrous(w0, h0, storage, src0, dst0, fb0, g0, step=1, phi_colour=10.0, phi_normal=128.0, iteration=0)
    reps, t2 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t2 < 1.0 and reps < 50):
        orc.atrous(w0, h0, storage, src0, dst0, fb0, g0, step=1, phi_colour=10.0, phi_normal=128.0, iteration=0)
        reps += 1
    cfg0_ms = (time.perf_counter() - t2) * 1e3 / reps
    return {"value": round(W * H * n / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "single_thread_value": round(one_mpx, 4),
            "config0_256x256_one_atrous_iteration": {"ms": round(cfg0_ms, 3), "Mpixels/s": round(w0 * h0 / (cfg0_ms * 1e-3) / 1e6, 3), "cores": 1,
                                                     "what": "BASELINE.json configs[0]: 256x256 synthetic G-buffer + noisy radiance, single a-trous iteration (step 1) via the scalar C++ loop"},
            "sample": f"{n} steady-state frames of {W}x{H} {storage} ({'1/4' if W == 1920 else '1/16'} of the 4K workload), temporal+moments+{iters} a-trous, "
                      f"oracle/svgf_oracle.cpp -O2 row-parallel on {cores} threads (os.cpu_count() = {os.cpu_count()}), {dt:.1f} s; 1920x1080 (configs[1]) is the sample when >= 32 threads are granted"}


def emit(line):
    """The JSON line must be the last thing on stdout: RCCL printf()s a version banner into C stdio's buffer, which
    would otherwise be flushed after Python's own output at exit."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    print(json.dumps(line), flush=True)


STRIP_SPEEDUP_TARGET = 6.0     # BASELINE.json north_star: ">= 6x strip-parallel scaling at 8 GPUs on 8K frames"


def strips_line(res, args, W, H, storage, iters, world, incomplete=None):
    """The JSON line from what bench_strips has measured so far (the headline plan at least)."""
    N, ab = world, ALG_BYTES[storage]
    ms = res["ms_per_step"]    # already the MAX over ranks
    # roofline of the dominant kernel on rank 0's strip: its a-trous launches (halo rows included) between HIP events
    n_l, ms_l, by_l = res["atrous_timing"](ab["atrous_iter"], ab["atrous_feedback"])
    roof = None
    if n_l and ms_l > 0:
        ach = by_l / (ms_l * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "atrous_lds_kernel", "scope": "rank 0, its strip incl. redundantly computed halo rows",
                "launches_timed": n_l, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                "algorithmic_bytes_per_launch": int(by_l / n_l), "avg_launch_ms": round(ms_l / n_l, 5), "traffic": None, "traffic_source": None}
    mpx = lambda t: round(W * H / (t * 1e-3) / 1e6, 1)      # noqa: E731
    full_gbps = alg_bytes_full(storage, iters) * W * H / (ms * 1e-3) / 1e9
    one = res["one_gpu_ms"]
    target = STRIP_SPEEDUP_TARGET if world == 8 else None      # the north_star's figure is for 8 GPUs; other world sizes carry none

    def plan_entry(t, rows_held, host_ms):
        sp = round(one / t, 3) if one else None
        return {"ms_per_step": round(t, 4), "Mpixels/s": mpx(t), "rows_held_per_rank": rows_held, "host_enqueue_ms_per_frame": host_ms,
                "speedup_vs_one_gpu": sp, "frac_of_aggregate_8TBps": round(alg_bytes_full(storage, iters) * W * H / (t * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world), 4),
                "target_speedup": target, "target_met": (bool(sp >= target) if (sp is not None and target) else None)}
    plans = {res["plan"]: plan_entry(ms, res["rows_held"], res["host_ms"])}
    for name, r in res["other_plans"].items():
        plans[name] = plan_entry(r["ms_per_step"], r["rows_held"], r["host_ms"])
    for name, r in [(res["plan"], res)] + list(res["other_plans"].items()):
        plans[name]["edge_first"] = bool(r.get("edge_first", False))
        if "ms_per_step_three_launches" in r:
            plans[name]["ms_per_step_three_launches"] = round(r["ms_per_step_three_launches"], 4)
    exchanging = {k: v for k, v in plans.items() if k != "ghost"}
    fastest = min(plans, key=lambda k: plans[k]["ms_per_step"])
    pan = res["pan"]
    if pan:
        pan = dict(pan, ms_per_step=round(pan["ms_per_step"], 4), **{"Mpixels/s": mpx(pan["ms_per_step"])})
    line = {
        "metric": METRIC, "value": mpx(ms),
        "unit": "Mpixels/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if storage == "f32" else "f16",
        "data": "synthetic",
        "config": {"workload": f"{W}x{H} {storage} storage in {world} row strips ({res['rows_per_rank']} rows per GPU), halo exchange over "
                               f"RCCL send/recv ({res['driver']} driver), plan {res['plan']}, temporal + moments + {iters} a-trous iterations, steady state, "
                               f"static camera (motion reach {res['motion_reach']} rows, from the inputs), current and previous G-buffer in distinct planes",
                   "width": W, "height": H, "storage": storage, "atrous_iterations": iters, "halo_plan": res["plan"], "driver": res["driver"],
                   "world_size": world,
                   "edge_first": bool(res.get("edge_first", False)),
                   "schedule": "edge rows first (svgf_strips_set_edge_first(1), an opt-in): used because this run reproduced the one-GPU frame bit for bit with it (`verified`)"
                               if res.get("edge_first") else "the library's default: every exchange ordered behind an event, three launches per exchanging iteration"},
        "verified": res.get("verified"),
        "one_gpu_ms": round(one, 4) if one else None,
        "one_gpu_note": f"the same {W}x{H} frame through svgf_denoise_frame on rank 0's GPU alone, timed in this run before the strips" if one else None,
        "speedup_vs_one_gpu": plans[res["plan"]]["speedup_vs_one_gpu"], "target_speedup": target, "target_met": plans[res["plan"]]["target_met"],
        "halo_plans": plans,
        "fastest_plan": fastest, "fastest_plan_that_exchanges_between_iterations": min(exchanging, key=lambda k: exchanging[k]["ms_per_step"]) if exchanging else None,
        "halo_plans_note": "value / ms_per_step are the plan `auto` resolves to (config.halo_plan): grouped — iterations {0,1,2} | {3,4}, 1 state + 1 filter-row exchange per "
                           "frame — where its halo fits the strips, else per-iteration = BASELINE.json configs[3]'s 'RCCL halo exchange per a-trous iter' (1 + 4 exchanges); "
                           "ghost: the state exchange only, every iteration recomputed on ghost rows (no exchange between iterations: listed, never the headline); "
                           "target_speedup: the north_star's >= 6x at 8 GPUs (null at other world sizes); "
                           "edge_first / ms_per_step_three_launches: which schedule ms_per_step was timed under (config.schedule, `verified`), and the same plan under "
                           "the library's default schedule where the opt-in was used",
        "pan": pan,
        "pan_note": "a camera pan whose state exchange carries moments and history rows as well as colour (motion reach >= 3); value / ms_per_step are the static camera",
        "rccl_ranks": res["rccl_ranks"],
        "roofline": roof,
        "pass_roofline": {"algorithmic_bytes_per_px": alg_bytes_full(storage, iters), "achieved_GBps": round(full_gbps, 1),
                          "frac_of_aggregate_8TBps": round(full_gbps / (HBM_PEAK_GBPS * world), 4)},
        "host_enqueue_ms_per_frame": res.get("host_ms"),
    }
    if incomplete:
        line["incomplete"] = incomplete
    return line


# ------------------------------------------------------------------ launcher for N > 1 ---------
def free_port():
    """A rendezvous port nobody listens on, BELOW the kernel's ephemeral range (32768-60999 on Linux): a port handed out by bind(0) comes from
    that range, where the local end of any outgoing connection of any process may take it between this test and the rendezvous (seen as one
    failed N > 1 test in ~25 runs of the suite)."""
    import random
    import socket
    rng = random.Random(os.getpid() ^ int(time.time() * 1e3))
    for _ in range(200):
        port = rng.randrange(20000, 32000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
                return port
            except OSError:
                continue
    with socket.socket() as s:                   # (nothing free there: whatever the kernel offers)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters, before anything in
    this process has touched the GPU), hand rank 0's JSON line through, fail if any rank fails."""
    import torch
    have = torch.cuda.device_count()          # does not initialise the GPU
    share = os.environ.get("SVGF_BENCH_SHARE_DEVICES") == "1"      # testing only: N ranks on fewer devices (process group: gloo, see main)
    if have < args.gpus and not (share and have >= 1):
        print(f"bench.py: --gpus {args.gpus} but only {have} device(s) are visible", file=sys.stderr)
        return 2
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % have if share else r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # wait for all of them; a rank that fails takes the others down (they would wait for it in a collective for ever)
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        time.sleep(0.2)
    if failed:
        time.sleep(2.0)
        for p in procs:
            if p.poll() is None:
                p.kill()              # exactly the processes started above
    out0 = procs[0].communicate()[0].decode()
    codes = [p.wait() for p in procs]
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if lines and codes and all(c == 5 for c in codes):      # a leg hung after the headline: the line of what was measured, and a failing exit code
        print(lines[-1], flush=True)
        print(f"bench.py: rank exit codes {codes} (a leg did not finish: see `incomplete`)", file=sys.stderr)
        return 5
    if any(codes) or not lines:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def main():
    args = parse()
    N = args.gpus
    if N > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1 or args.strips:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:           # single process (--strips): any free port
            os.environ["MASTER_PORT"] = str(free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # SVGF_BENCH_SHARE_DEVICES=1 (tests: rank processes sharing a device): torch's process group over gloo.  RCCL itself refuses two
        # ranks on one device, so such a job ends in bench_strips with a non-zero exit code — there is no other driver to fall back to
        if os.environ.get("SVGF_BENCH_SHARE_DEVICES") == "1":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        assert world == N, f"--gpus {N} but WORLD_SIZE={world}"

    storage, iters = args.storage, args.iters
    if N == 1 and not args.strips:
        wl = args.workload or "4k"
        W, H = WORKLOADS[wl]
        scene = Scene(W, H, device)
        motions = ["static", "pan"] if args.motion == "both" and not args.no_extra else [args.motion if args.motion != "both" else "static"]
        res = {}
        fuse = bool(args.fuse)
        for m in motions:
            res[m] = run_single(FramePool(scene, storage, m), W, H, storage, iters, args.variant, args.steps, args.warmup, device,
                                cold_frames=5 if (m == "static" and not args.no_extra) else 0, fuse=fuse, windows=args.windows, in_flight=args.frames_in_flight, prime=(args.prime_ms, args.prime_frames),
                                prev_guide=args.prev_guide, path_stats=(m == motions[0] and not fuse))
        head = motions[0]
        r = res[head]
        ms = r["ms_per_step"]
        value = W * H / (ms * 1e-3) / 1e6
        roof, stages = roofline_block(W, H, storage, iters, r["stage_ms"], args.variant, r["fused"], args.prev_guide)
        line = {
            "metric": METRIC, "value": round(value, 1),
            "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if storage == "f32" else "f16",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} {storage} storage, temporal + moments + {iters} a-trous iterations (steps 1..{1 << max(iters - 1, 0)}), "
                                   f"steady state (history >= 4), {'static camera' if head == 'static' else 'camera pan ' + str(PAN_MV)}, current and previous "
                                   f"G-buffer in distinct planes (ping-ponged), 1-spp noise, seed 0x5356474600000001",
                       "width": W, "height": H, "storage": storage, "atrous_iterations": iters, "variant": args.variant, "motion": head,
                       "iterations_0_and_1_in_one_launch": r["fused"], "frames_in_flight": r["in_flight"],
                       "prev_guide": "off (ABI default): the reprojection test reads the previous G-buffer's three planes; also.prev_guide_on is the same run with the opt-in" if not args.prev_guide else
                                     "on (svgf_set_prev_guide, an opt-in of the ABI: the host vouches that the previous G-buffer's planes are last frame's current ones, "
                                     "not rewritten in between - as in the reference, App.cu:374)"},
            **timing_fields(r),
            "roofline": roof,
            "pass_roofline": pass_block(W, H, storage, iters, ms, r["fused"], args.prev_guide),
            "stages": stages,
            "stages_note": "stage times from HIP events recorded by the library on its launch stream, on every 4th timed frame; svgf_denoise_frame folds "
                           "the steady-state moments copy into the temporal launch and the moments slot only re-filters the young pixels the temporal launch listed, "
                           "so the two stages are rated together: algorithmic_* uses the SURVEY 8(d) bytes, moved_* the bytes the fused "
                           "launches really touch; measured_hbm_bytes / roofline.traffic are NOT measured in this run: they come from rocprofv3 PMC passes of the "
                           "same sources (roofline.traffic_source) or are null",
            "young_fraction": round(r["young_fraction"], 5),
            "scene": SCENE_NOTE["planar"],
            "uniform_normal_path_share": r["uniform_path_share"],
            "uniform_normal_path_share_note": "share of the a-trous wave-steps (64 pixels of a row that hold a surface pixel) served by the uniform-normal tap path (8 instead of 13 vector "
                                              "instructions per tap, bit-identical), per step and over all launches, counted on the device over 8 frames after the timed windows "
                                              "(svgf_path_stats_enable); also.curved_scene is the same measurement on geometry with per-texel normals",
        }
        if "pan" in res and head != "pan":
            p = res["pan"]
            _, pst = roofline_block(W, H, storage, iters, p["stage_ms"], args.variant, p["fused"])
            line["pan"] = {"mv": list(PAN_MV), "pool_frames": PAN_POOL, "ms_per_step": round(p["ms_per_step"], 4), "ms_per_step_min": round(min(p["windows_ms"]), 4),
                           "ms_per_step_max": round(max(p["windows_ms"]), 4),
                           "Mpixels/s": round(W * H / (p["ms_per_step"] * 1e-3) / 1e6, 1),
                           "frac_of_8TBps": pass_block(W, H, storage, iters, p["ms_per_step"])["frac_of_8TBps"],
                           "young_fraction": round(p["young_fraction"], 5),
                           "temporal_ms": pst["temporal+moments"]["temporal_ms"], "moments_ms": pst["temporal+moments"]["moments_ms"],
                           "stage_ms": {k: v["ms"] for k, v in pst.items()}}
        if r["cold_ms"]:
            cold_b = alg_bytes_full(storage, iters) + (32 if storage == "f32" else 28)      # moments 65 / 45 B/px cold instead of 33 / 17
            worst = max(r["cold_ms"][:3])
            line["cold_frames_ms"] = {"after_reset": r["cold_ms"], "algorithmic_bytes_per_px_cold": cold_b,
                                      "frac_of_8TBps_cold_worst": round(cold_b * W * H / (worst * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                      "note": "frames 0.. after svgf_reset_history, sum of stage events, one frame at a time (the device idles between "
                                              "them); history < 4 on frames 0-2: the 7x7 moments estimate runs on every pixel (49 taps: arithmetic-, not HBM-bound)"}
        if not args.no_extra and wl != "1080p":
            W2, H2 = WORKLOADS["1080p"]
            sc2 = Scene(W2, H2, device, pool=2)
            r2 = run_single(FramePool(sc2, storage, "static"), W2, H2, storage, iters, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=args.windows, prime=(args.prime_ms, args.prime_frames))
            roof2, _ = roofline_block(W2, H2, storage, iters, r2["stage_ms"], args.variant, r2["fused"])
            line["also"] = {"1920x1080": {"ms_per_step": round(r2["ms_per_step"], 4), "ms_per_step_min": round(min(r2["windows_ms"]), 4),
                                          "ms_per_step_without_stage_events": round(r2["ms_no_events"], 4),
                                          "Mpixels/s": round(W2 * H2 / (r2["ms_per_step"] * 1e-3) / 1e6, 1),
                                          "frac_of_8TBps": pass_block(W2, H2, storage, iters, r2["ms_per_step"])["frac_of_8TBps"],
                                          "atrous_avg_launch_ms": roof2["avg_launch_ms"] if roof2 else None, "atrous_roofline_frac": roof2["frac"] if roof2 else None}}
            if args.frames_in_flight == 1 and not fuse and "pan" in motions:
                # the bench pan at 1080p: the same few pixels per frame enter a frame a quarter the size (4 500 waves hold young pixels, of 32 400)
                sc2p = Scene(W2, H2, device)
                r2p = run_single(FramePool(sc2p, storage, "pan"), W2, H2, storage, iters, args.variant, max(args.steps, 20), args.warmup, device, windows=3,
                                 prime=(min(args.prime_ms, 150.0), min(args.prime_frames, 200)))
                _, st2p = roofline_block(W2, H2, storage, iters, r2p["stage_ms"], args.variant, r2p["fused"])
                line["also"]["1920x1080"]["pan"] = {"mv": list(PAN_MV), "ms_per_step": round(r2p["ms_per_step"], 4), "Mpixels/s": round(W2 * H2 / (r2p["ms_per_step"] * 1e-3) / 1e6, 1),
                                                    "temporal_ms": st2p["temporal+moments"]["temporal_ms"] if st2p else None,
                                                    "moments_ms": st2p["temporal+moments"]["moments_ms"] if st2p else None}
                del sc2p
            if args.frames_in_flight == 1 and not fuse:
                line["also"]["1920x1080"]["hip_graph"] = run_graph_replay(FramePool(sc2, storage, "static"), W2, H2, storage, iters, args.variant, max(args.steps, 40), device,
                                                                          prime_ms=min(300.0, args.prime_ms))
            del sc2
            if storage == "f32" and wl == "4k":      # BASELINE configs[4]: the reference-native fp16 storage on the same frame
                r3 = run_single(FramePool(scene, "f16", "static"), W, H, "f16", iters, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=args.windows, prime=(args.prime_ms, args.prime_frames))
                roof3, _ = roofline_block(W, H, "f16", iters, r3["stage_ms"], args.variant, r3["fused"])
                r3g = run_single(FramePool(scene, "f16", "static"), W, H, "f16", iters, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=3,
                                 prime=(min(args.prime_ms, 150.0), min(args.prime_frames, 200)), prev_guide=True)
                line["also"]["3840x2160_f16"] = {"ms_per_step": round(r3["ms_per_step"], 4), "ms_per_step_prev_guide_on": round(r3g["ms_per_step"], 4),
                                                 "Mpixels/s": round(W * H / (r3["ms_per_step"] * 1e-3) / 1e6, 1),
                                                 "frac_of_8TBps": pass_block(W, H, "f16", iters, r3["ms_per_step"])["frac_of_8TBps"],
                                                 "atrous_avg_launch_ms": roof3["avg_launch_ms"] if roof3 else None,
                                                 "atrous_roofline_frac": roof3["frac"] if roof3 else None,
                                                 "roofline_secondary": roof3.get("secondary") if roof3 else None,
                                                 # BASELINE.md's own 60 % line for configs[4] (323 B/px at 0.60 x 8 TB/s): where the number is, so is the miss
                                                 "target_ms": 0.558, "target_met": bool(r3["ms_per_step"] <= 0.558), "target_met_prev_guide_on": bool(r3g["ms_per_step"] <= 0.558), "bound": "valu",
                                                 "why": "fp16 storage halves the colour bytes and none of the arithmetic: the a-trous launches run the same ~290 vector "
                                                        "instructions per pixel on 40 instead of 56 B/px and are bound by vector issue (valu_busy, DESIGN.md 3.3), the temporal "
                                                        "launch already moves its bytes at the part's copy rate"}
        if not args.no_extra and wl == "4k" and storage == "f32":
            # BASELINE configs[3]'s frame on ONE GPU: the denominator of every strip-scaling figure
            sc8 = UpsampledScene(scene)
            r8 = run_single(FramePool(sc8, storage, "static"), sc8.W, sc8.H, storage, iters, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=3,
                            prime=(min(args.prime_ms, 300.0), min(args.prime_frames, 200)), prev_guide=args.prev_guide)
            roof8, _ = roofline_block(sc8.W, sc8.H, storage, iters, r8["stage_ms"], args.variant, r8["fused"], args.prev_guide)
            line.setdefault("also", {})["7680x4320"] = {
                "ms_per_step": round(r8["ms_per_step"], 4), "ms_per_step_min": round(min(r8["windows_ms"]), 4), "ms_per_step_max": round(max(r8["windows_ms"]), 4),
                "Mpixels/s": round(sc8.W * sc8.H / (r8["ms_per_step"] * 1e-3) / 1e6, 1),
                "pass_roofline": {k: v for k, v in pass_block(sc8.W, sc8.H, storage, iters, r8["ms_per_step"], r8["fused"], args.prev_guide).items() if k != "note"},
                "atrous_avg_launch_ms": roof8["avg_launch_ms"] if roof8 else None, "atrous_roofline_frac": roof8["frac"] if roof8 else None,
                "scene": "the 4K scene's G-buffer at twice the sampling rate (2 x 2 texel blocks, ddepth halved), fresh 1-spp noise per 8K pixel, made on the device; "
                         "`bench.py --workload 8k` times svgf_amd/synth.py's own 8K scene (profiles/r05_bench_8k_f32.json)",
                "note": "7680x4320 through svgf_denoise_frame on one GPU: what the strip-parallel speed-up at N GPUs is relative to (configs[3])"}
            del sc8
        if not args.no_extra and not args.prev_guide and wl == "4k":
            # the opt-in: svgf_set_prev_guide on (the host vouches for the previous G-buffer's planes; the pools do leave them alone)
            r6 = run_single(FramePool(scene, storage, "static"), W, H, storage, iters, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=args.windows, prime=(args.prime_ms, args.prime_frames), prev_guide=True)
            roof6, st6 = roofline_block(W, H, storage, iters, r6["stage_ms"], args.variant, r6["fused"], True)
            line.setdefault("also", {})["prev_guide_on"] = {
                "ms_per_step": round(r6["ms_per_step"], 4), "ms_per_step_min": round(min(r6["windows_ms"]), 4), "ms_per_step_max": round(max(r6["windows_ms"]), 4),
                "Mpixels/s": round(W * H / (r6["ms_per_step"] * 1e-3) / 1e6, 1), "frac_of_8TBps": pass_block(W, H, storage, iters, r6["ms_per_step"], False, True)["frac_of_8TBps"],
                "temporal_ms": st6["temporal+moments"]["temporal_ms"] if st6 else None,
                **({k: v for k, v in roof6.items() if k.startswith("atrous_x")} if roof6 else {}),
                "atrous_avg_launch_ms": roof6["avg_launch_ms"] if roof6 else None, "atrous_roofline_frac": roof6["frac"] if roof6 else None,
                "note": "svgf_set_prev_guide(ctx, 1), an opt-in: the reprojection test reads the guide plane the previous frame kept (16 B/px) instead of motion / normal / uv of "
                        "the previous G-buffer (32 B/px); rounds 2-4 quoted this configuration as the headline"}
        if not args.no_extra and wl == "4k" and iters == 5:
            # the GUI's range is 0-10 iterations (GUI.cpp:988), step = 1 << i (App.cu:502): steps 32 and 64 run through the LDS kernel too
            r7 = run_single(FramePool(scene, storage, "static"), W, H, storage, 7, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=3, prime=(args.prime_ms, args.prime_frames),
                            prev_guide=args.prev_guide)
            line.setdefault("also", {})["seven_iterations"] = {"ms_per_step": round(r7["ms_per_step"], 4), "Mpixels/s": round(W * H / (r7["ms_per_step"] * 1e-3) / 1e6, 1),
                                                               "atrous_launch_ms_by_step": {str(1 << i): round(r7["stage_ms"][2 + i], 5) for i in range(7)},
                                                               "note": "temporal + moments + 7 a-trous iterations (steps 1..64), all LDS-streaming launches"}
        if not args.no_extra and wl == "4k" and args.frames_in_flight == 1 and storage == "f32":
            # Heavy disocclusion: every 8th column of ONE of the two G-buffers the frames alternate between has its normals flipped, so those columns fail
            # the reprojection test in every frame (12 % of the surface pixels young, some in EVERY wave of the temporal launch): what thin geometry under
            # motion or a fast camera does to the young-pixel machinery (tools/archive/young_worst_case.py)
            import torch
            cp = FramePool(scene, storage, "static")
            cp.gb[1].normal.view(torch.int16)[:, ::8, 0:3] ^= -32768
            crowd = {}
            for name, ad in (("adaptive", True), ("young_pixel_launch_only", False)):
                rc = run_single(cp, W, H, storage, iters, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=3, prime=(min(args.prime_ms, 150.0), min(args.prime_frames, 200)), adaptive=ad)
                _, stc = roofline_block(W, H, storage, iters, rc["stage_ms"], args.variant, rc["fused"])
                crowd[name] = {"ms_per_step": round(rc["ms_per_step"], 4), "temporal_ms": stc["temporal+moments"]["temporal_ms"] if stc else None,
                               "moments_ms": stc["temporal+moments"]["moments_ms"] if stc else None, "young_fraction": round(rc["young_fraction"], 4)}
            crowd["note"] = ("every 8th column disoccluded in every frame; adaptive (default): the frame driver serves such frames with the LDS-streaming moments kernel "
                             "(svgf_set_adaptive_moments: same bits); young_pixel_launch_only: svgf_set_adaptive_moments(0); round 3's sources: 2.6 ms")
            line.setdefault("also", {})["crowded_frames"] = crowd
            del cp
        if not args.no_extra and wl == "4k" and args.frames_in_flight == 1:
            line.setdefault("also", {})["interactive"] = run_interactive(FramePool(scene, storage, "static"), W, H, storage, iters, args.variant, device)
        if not args.no_extra and args.variant == "auto" and wl == "4k":
            # the synthetic scene is piecewise planar: 68-87 % of the a-trous waves take the uniform-normal fast path (8 instead of 13
            # vector instructions per tap, same results).  What geometry without planar regions would cost: the fast path switched off.
            r4 = run_single(FramePool(scene, storage, "static"), W, H, storage, iters, "lds-general", max(args.steps, 20), args.warmup, device, fuse=fuse, windows=args.windows, prime=(args.prime_ms, args.prime_frames))
            roof4, _ = roofline_block(W, H, storage, iters, r4["stage_ms"], "auto", r4["fused"])
            line["also"]["no_uniform_normal_fast_path"] = {"ms_per_step": round(r4["ms_per_step"], 4), "Mpixels/s": round(W * H / (r4["ms_per_step"] * 1e-3) / 1e6, 1),
                                                           "frac_of_8TBps": pass_block(W, H, storage, iters, r4["ms_per_step"])["frac_of_8TBps"],
                                                           "atrous_avg_launch_ms": roof4["avg_launch_ms"] if roof4 else None,
                                                           "atrous_roofline_frac": roof4["frac"] if roof4 else None}
        if not args.no_extra and wl == "4k":
            # Geometry whose normal differs from texel to texel (GBuffer.frag:65 writes normalize(FragNormal): any smooth-shaded mesh): rolling terrain, a
            # sphere, a cylinder (svgf_amd/synth.py, scene "curved").  The headline's 60 % claim is scoped to the piecewise-planar scene SURVEY 8(d) prescribes.
            scc = Scene(W, H, device, pool=2, scene="curved")
            r9 = run_single(FramePool(scc, storage, "static"), W, H, storage, iters, args.variant, max(args.steps, 20), args.warmup, device, fuse=fuse, windows=args.windows,
                            prime=(args.prime_ms, args.prime_frames), prev_guide=args.prev_guide, path_stats=not fuse)
            roof9, _ = roofline_block(W, H, storage, iters, r9["stage_ms"], args.variant, r9["fused"], args.prev_guide)
            pb9 = pass_block(W, H, storage, iters, r9["ms_per_step"], r9["fused"], args.prev_guide)
            line.setdefault("also", {})["curved_scene"] = {
                "ms_per_step": round(r9["ms_per_step"], 4), "ms_per_step_min": round(min(r9["windows_ms"]), 4), "ms_per_step_max": round(max(r9["windows_ms"]), 4),
                "Mpixels/s": round(W * H / (r9["ms_per_step"] * 1e-3) / 1e6, 1), "frac_of_8TBps": pb9["frac_of_8TBps"],
                "target_frac": 0.60, "target_met": bool(pb9["frac_of_8TBps"] >= 0.60),
                "atrous_avg_launch_ms": roof9["avg_launch_ms"] if roof9 else None, "atrous_roofline_frac": roof9["frac"] if roof9 else None,
                **({k: v for k, v in roof9.items() if k.startswith("atrous_x")} if roof9 else {}),
                "uniform_normal_path_share": r9["uniform_path_share"], "young_fraction": round(r9["young_fraction"], 5),
                "scene": SCENE_NOTE["curved"]}
            del scc
        if not args.no_extra and args.frames_in_flight == 1 and wl == "4k":
            # throughput mode: iterations 1.. of frame f on a side stream beside the temporal launch of frame f + 1 (same results)
            r5 = run_single(FramePool(scene, storage, "static"), W, H, storage, iters, args.variant, args.steps, args.warmup, device, fuse=fuse, windows=args.windows, in_flight=2, prime=(args.prime_ms, args.prime_frames))
            roof5, st5 = roofline_block(W, H, storage, iters, r5["stage_ms"], args.variant, r5["fused"])
            line.setdefault("also", {})["two_frames_in_flight"] = {
                "ms_per_step": round(r5["ms_per_step"], 4), "ms_per_step_min": round(min(r5["windows_ms"]), 4), "ms_per_step_max": round(max(r5["windows_ms"]), 4),
                "Mpixels/s": round(W * H / (r5["ms_per_step"] * 1e-3) / 1e6, 1), **{k: v for k, v in pass_block(W, H, storage, iters, r5["ms_per_step"], r5["fused"]).items()
                                                                                  if k in ("frac_of_8TBps", "moved_frac_of_8TBps")},
                "atrous_avg_launch_ms": roof5["avg_launch_ms"] if roof5 else None, "atrous_roofline_frac": roof5["frac"] if roof5 else None,
                "temporal_ms": st5["temporal+moments"]["temporal_ms"] if st5 else None,
                "note": "svgf_set_frames_in_flight(2): launch durations here overlap (a-trous launches of frame f beside the temporal launch of frame f + 1), "
                        "so they are longer than in the headline run while the frame is shorter"}
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(storage, iters)
        if world > 1 or args.strips:
            import torch.distributed as dist
            dist.destroy_process_group()
        emit(line)
        return

    # N > 1 (or --strips): one 8K frame in N row strips with halo exchange
    import torch.distributed as dist
    from svgf_amd import strips
    wl = args.workload or "8k"
    W, H = WORKLOADS[wl]

    # A leg that never returns (an exchange that does not complete on this node) must not take the measured legs with it: every rank runs the
    # same watchdog; when a leg overruns its allowance, rank 0 prints the line of what IS measured (value = the headline plan, the legs after it
    # absent and `incomplete` naming the one that hung) and every rank leaves with exit code 5 — a hang is never a success.  Before the headline
    # there is nothing to print: exit code 4.  (A process that has touched the GPU is never re-executed: it prints and leaves.)
    import threading
    import time as _time
    watch = {"phase": "start", "deadline": _time.monotonic() + args.leg_timeout, "line": None, "done": False}

    def on_phase(name):
        watch["phase"], watch["deadline"] = name, _time.monotonic() + args.leg_timeout
        if name.startswith(os.environ.get("SVGF_BENCH_HANG_AT") or "\0"):       # tests: a leg that never returns
            _time.sleep(1e6)

    def on_head(res):
        watch["so_far"] = res

    def watchdog():
        while not watch["done"]:
            _time.sleep(0.5)
            if _time.monotonic() > watch["deadline"] and not watch["done"]:
                why = f"leg '{watch['phase']}' did not finish within {args.leg_timeout:.0f} s on rank {rank}"
                print(f"bench.py: {why}: leaving", file=sys.stderr, flush=True)
                if watch.get("so_far") is None:
                    os._exit(4)
                if rank == 0:
                    emit(strips_line(watch["so_far"], args, W, H, storage, iters, world, incomplete=why + "; the legs after it are absent"))
                os._exit(5)            # a leg hung: never exit code 0, whatever was measured before it is in the line
    threading.Thread(target=watchdog, daemon=True).start()

    try:
        res = strips.bench_strips(W, H, storage, iters, args.variant, args.steps, args.warmup, device, plan=args.halo_plan,
                                  make_inputs=make_inputs, prime_frames=PRIME_FRAMES,
                                  plans=() if args.no_extra else ("per-iteration", "grouped", "ghost"), pan_mv=None if args.no_extra else STRIP_PAN_MV,
                                  one_gpu_reference=not args.no_one_gpu, busy=(args.prime_ms, args.prime_frames), on_phase=on_phase, on_head=on_head)
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: rank {rank}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        os._exit(3)            # a rank that cannot run the measurement asked for ends the job (the launcher reports the exit codes)
    on_phase("closing")
    line = strips_line(res, args, W, H, storage, iters, world) if rank == 0 else None
    dist.barrier()
    dist.destroy_process_group()
    watch["done"] = True
    if line is not None:
        emit(line)


if __name__ == "__main__":
    main()
