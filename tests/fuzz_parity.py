"""Seeded random sweep of the parity and bit-identity claims (test infrastructure: it calls the oracle; nothing in the product imports it).

    python -m tests.fuzz_parity --minutes 10 --seed 1000 [--kinds stage,strips,driver,rows,pair,post,stage0,negzero,strips2,driver2,wide,widestrips,edge,edgedriver,edgestrips,graph,fullsize] [--out gpurun_out/fuzz.txt]

Each trial draws a frame size (down to 1 x 1, up past the 128-pixel tile and the 64-lane wave in both directions), a storage format, the
tunables over the GUI's ranges (GUI.cpp:988-993), a camera motion, optionally NaN / inf radiance texels and poisoned G-buffer texels
(tests/gbuffer_poison.py), and checks one of

  stage    temporal (bitwise), moments and one a-trous iteration (stage tolerances, tests/gpu_helpers.py:TOL) against the ORACLE from
           identical inputs, under a random kernel variant;
  strips   the C++ strip driver with real peer addressing (mailbox transport, random world size, halo plan, motion reach, schedule) against the
           single-context FRAME DRIVER, bit for bit, over a few frames;
  driver   the frame driver under a random setting (two frames in flight, general tap path, young-pixel launch only, svgf_set_prev_guide, or the
           stage calls on caller-owned planes instead) against the plain frame driver, bit for bit, and its history against the oracle's
           free-running one;
  rows     a stage call restricted to a row range (svgf_set_rows): the rows inside with the whole-frame call's bits, nothing else written;
  pair     iterations 0 and 1 in one launch: the frame driver with the fusion against without (finite input, bit for bit); poisoned input:
           svgf_atrous_pair's feedback plane against the oracle, its result against the device's own two launches (NaN masks, a bound);
  strips2  the strip driver over RCCL's own kernels (loop-back communicator) or the mailbox, one or two frames in flight, svgf_set_prev_guide, the
           fusion of iterations 0 + 1 (finite input) against the plain frame driver, bit for bit;
  driver2  the frame driver across svgf_reset_history / svgf_resize / svgf_set_params between frames: plain, under a random setting, and a fresh
           context from the last restart on — bit for bit;
  wide, widestrips   `stage` and `strips` on frames 1 024 - 8 200 columns wide (many column tiles: the XCD-aware tile order; few rows);
  edge     `stage` with the tunables at and beyond the ends of their ranges (PhiColour / PhiNormal 0, NaN / inf / negative thresholds, a history base of
           0, 256, 1 000, -5) and steps the LDS kernel does not serve (3, 5, 7, 100, 128, 256, 512);
  edgedriver, edgestrips   `driver` and `strips` with those tunables, a moments radius of 0-3 and 0-10 (0-7) iterations;
  graph    the frame driver recorded into a HIP graph and replayed against the directly enqueued frames, bit for bit;
  fullsize   1920x1080 and 3840x2160: strip driver and frame driver under a setting against the plain frame driver, bit for bit, poisoned frames;
  stage0   `stage` with -0.0, denormals and the storage type's extremes in the colour and moments planes;
  negzero  `stage0` with rectangles of -0.0 (inside, across the border, over the whole frame) and a tenth of all texels -0.0 in one channel;
  post     the stages after the path: TAA + sRGB against the oracle and tiled against per-pixel, albedo (de)modulation bit-exact.

A trial is a pure function of its seed: `run_trial(kind, seed)` re-runs one (tests/test_gpu_fuzz.py pins the seeds that ever failed, and a few
that never did).  Exit code 1 if any trial failed; the summary lists the seeds."""
from __future__ import annotations

import argparse
import sys
import time
import traceback

import numpy as np

from svgf_amd import synth
from tests.gbuffer_poison import poison_gbuffer
from tests.helpers import CDT, gbuf

KINDS = ("stage", "strips", "driver", "rows", "pair", "post", "stage0", "negzero", "strips2", "driver2", "wide", "widestrips", "edge", "edgedriver", "edgestrips", "graph", "fullsize")


def _size(rng):
    """Frame sizes: mostly ragged mid-sized, sometimes tiny, sometimes just around the tile / wave widths."""
    c = rng.integers(0, 10)
    if c == 0:
        return int(rng.integers(1, 9)), int(rng.integers(1, 9))
    if c == 1:
        return int(rng.choice([63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513])), int(rng.integers(1, 80))
    if c == 2:
        return int(rng.integers(1, 40)), int(rng.integers(60, 700))
    return int(rng.integers(9, 700)), int(rng.integers(9, 260))


def _tunables(rng):
    return dict(phi_colour=float(np.exp(rng.uniform(np.log(0.05), np.log(100.0)))), phi_normal=float(rng.uniform(0.5, 256.0)),
                depth_threshold=float(rng.uniform(0.0, 3.0)), normal_threshold=float(rng.uniform(0.0, 1.0)),
                history_base=int(rng.integers(1, 256)), mesh_id_test=int(rng.integers(0, 2)))


def _sprinkle(rng, a, n):
    """n NaN / +inf / -inf texels (single channels) into a float plane."""
    flat = a.reshape(-1)
    if flat.size:
        idx = rng.integers(0, flat.size, n)
        flat[idx] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), n).astype(a.dtype)
    return a


def _sprinkle_zeros(rng, a, n):
    """n texels of -0.0, denormals and the storage type's extremes into a float plane (the kinds added later use it: the first three keep
    their draws, so that their pinned seeds keep their frames)."""
    flat = a.reshape(-1)
    if flat.size:
        tiny = np.finfo(a.dtype).tiny
        vals = np.array([-0.0, 0.0, tiny / 4, -tiny / 4, tiny, np.finfo(a.dtype).max, -np.finfo(a.dtype).max], np.float64)
        flat[rng.integers(0, flat.size, n)] = rng.choice(vals, n).astype(a.dtype)
    return a


def _blocks_of_negzero(rng, a, n):
    """n rectangles of -0.0 (a random subset of the channels each; some touch the frame's border, some cover it) plus a tenth of the texels:
    kind "negzero" — the sign a zero keeps through the reference's comparison-built clamp (Filter.cuh:57-82) and the sums that start from it."""
    H, W, C = a.shape
    for _ in range(n):
        h, w = int(rng.integers(1, max(2, H))), int(rng.integers(1, max(2, W)))
        y, x = int(rng.integers(-h // 2, H)), int(rng.integers(-w // 2, W))
        ch = np.nonzero(rng.integers(0, 2, C))[0]
        a[max(y, 0):y + h, max(x, 0):x + w, ch if len(ch) else slice(None)] = -0.0
    m = rng.random((H, W)) < 0.1
    a[m, rng.integers(0, C, int(m.sum()))] = -0.0
    return a


def _poisoned(rng, f, what):
    H, W = f["region"].shape
    if (f["region"] != synth.SKY).sum() < 4:
        return f
    return poison_gbuffer(rng, f, what=what, per_value=int(rng.integers(1, 5)))[0]


def _same_bits(a, b):
    """Bit for bit, the sign of a zero included (the reference's value clamp is built from comparisons and keeps -0.0, Filter.cuh:57-82: so do
    clamp01_ref and the streaming kernels' second pass, svgf_device.h) — except that a NaN is a NaN: one that the arithmetic MAKES (inf x 0,
    inf - inf) carries the sign the machine gives it (x86: set, gfx950: clear), one that is passed through keeps its bits on both."""
    na, nb = np.isnan(a.astype(np.float32)), np.isnan(b.astype(np.float32))
    u = {2: np.uint16, 4: np.uint32}[a.dtype.itemsize]
    return np.array_equal(na, nb) and np.array_equal(a.view(u)[~na], b.view(u)[~nb])


def _sky_mask(frame):
    """GetDepth() == the sentinel (Filter.cuh:199-207,552): depth 0, or literally 1e30 — the texels FilterKernel copies."""
    z = frame["motion"][..., 2]
    return (z == 0) | (z == np.float32(1e30))


def _close(G, got, want, storage, what, colour_abs=None):
    from tests.test_gpu_nonfinite import assert_close_with_nan
    if colour_abs is None:
        assert_close_with_nan(G, got, want, storage, what)
    else:
        assert_close_with_nan(G, got, want, storage, what, colour_abs=colour_abs)


# ------------------------------------------------------------------------------------------------------------------ stage vs oracle
def trial_stage(G, oracle, seed, zeros=False, wide=False, edge=False):
    """zeros: -0.0, denormals and the storage type's extremes in the colour / moments planes as well (kind "stage0"; drawn from a generator of
    their own, so that kind "stage" keeps the frames of its pinned seeds)."""
    from svgf_amd import filter as F
    rng = np.random.default_rng(seed)
    rz = np.random.default_rng(seed ^ 0x5A5A5A)
    W, H = _size(rng)
    if wide:                                              # (kind "wide": many column tiles — the XCD-aware tile order, rows shorter than a band)
        W, H = int(rz.choice([int(rz.integers(1024, 8200)), 1920, 3840, 4096, 7680, 8191])), int(rz.integers(1, 48))
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    dt = CDT[storage]
    tun = _tunables(rng)
    variant = str(rng.choice(["auto", "lds", "direct", "lds-general"]))
    radius = int(rng.choice([3, 3, 1]))
    step = int(2 ** rng.integers(0, 7))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    poison = bool(rng.integers(0, 2))
    if edge:                                              # (kind "edge": the ends of the tunables' ranges and beyond, steps the LDS kernel does not serve)
        pick = lambda vals: vals[int(rz.integers(0, len(vals)))]       # noqa: E731
        tun = dict(phi_colour=pick([0.0, 1e-6, 1e6, tun["phi_colour"]]), phi_normal=pick([0.0, 1e-3, 512.0, tun["phi_normal"]]),
                   depth_threshold=pick([0.0, -1.0, float("inf"), float("nan"), tun["depth_threshold"]]),
                   normal_threshold=pick([-1.0, 0.0, 1.0, 2.0, float("nan"), tun["normal_threshold"]]),
                   history_base=pick([0, 1, 2, 255, 256, 1000, -5]), mesh_id_test=tun["mesh_id_test"])
        step = pick([step, 3, 5, 7, 100, 128, 256, 512])
        radius = pick([radius, 0, 2, 3])
    f0, f1 = synth.make_frame(W, H, seed % 97, mv=mv), synth.make_frame(W, H, seed % 97 + 1, mv=mv)
    if poison:
        f0, f1 = _poisoned(rng, f0, ("motion", "depth", "ddepth", "normal", "id")), _poisoned(rng, f1, ("motion", "depth", "ddepth", "normal", "id"))
    desc = f"stage seed {seed}: {W}x{H} {storage} {variant} r{radius} step {step} poison {poison}" + (f" edge {tun}" if edge else "")
    d = F.Denoiser(W, H, F.Params(storage=storage, moments_radius=radius, variant=variant, **tun))
    # temporal: bit-exact whatever the inputs
    prev = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
    mom_prev = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hist_prev = rng.integers(0, 256, (H, W)).astype(np.uint8)
    cur = rng.uniform(-0.1, 1.4, (H, W, 4)).astype(dt)
    if poison:
        _sprinkle(rng, cur, 5), _sprinkle(rng, prev, 5), _sprinkle(rng, mom_prev, 3)
    if zeros:
        _sprinkle_zeros(rz, cur, 8), _sprinkle_zeros(rz, prev, 8), _sprinkle_zeros(rz, mom_prev, 4)
    if zeros == 2:
        _blocks_of_negzero(rz, cur, 4), _blocks_of_negzero(rz, prev, 4), _blocks_of_negzero(rz, mom_prev, 2)
    o = np.zeros_like(cur); hist = np.zeros((H, W), np.uint8); mom = np.zeros((H, W, 2), dt)
    oracle.temporal(W, H, storage, prev, cur, o, gbuf(f1), gbuf(f0), hist_prev, hist, mom, mom_prev, depth_threshold=tun["depth_threshold"],
                    normal_threshold=tun["normal_threshold"], history_base=tun["history_base"], mesh_id_test=tun["mesh_id_test"])
    o_col, o_hist, o_mom = d.new_colour(), d.new_history(), d.new_moments()
    d.TemporalFilter(G.dev(prev), G.dev(cur), o_col, G.gb_dev(f1), G.gb_dev(f0), G.dev(hist_prev), o_hist, o_mom, G.dev(mom_prev))
    try:
        assert np.array_equal(G.host(o_hist), hist), desc + ": history"
        assert _same_bits(G.host(o_col), o), desc + ": temporal colour"
        assert _same_bits(G.host(o_mom), mom), desc + ": temporal moments"
    except AssertionError as e:
        e.ctx = dict(stage="temporal", got=(G.host(o_hist), G.host(o_col), G.host(o_mom)), want=(hist, o, mom), prev=prev, cur=cur, mom_prev=mom_prev, hist_prev=hist_prev, f0=f0, f1=f1)
        raise
    # the depth-, ddepth- and normal-poisoned G-buffer for the spatial stages (a poisoned motion vector is the temporal stage's business)
    fs = synth.make_frame(W, H, seed % 97 + 1, mv=mv)
    if poison:
        fs = _poisoned(rng, fs, ("depth", "ddepth", "normal"))
    # moments
    col = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    momp = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hl = rng.integers(1, 8, (H, W)).astype(np.uint8)
    if zeros:
        flat = col.reshape(-1)
        flat[rz.integers(0, flat.size, 6)] = rz.choice(np.array([-0.0, 0.0, np.finfo(dt).tiny / 4, np.finfo(dt).tiny], np.float64), 6).astype(dt)
    want = np.zeros_like(col)
    oracle.moments(W, H, storage, col, want, momp, gbuf(fs), hl, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"], radius=radius)
    out = d.new_colour()
    d.FilterMoments(G.dev(col), out, G.dev(momp), G.gb_dev(fs), G.dev(hl))
    got = G.host(out)
    keep = hl >= 4
    assert np.array_equal(got[keep].view(np.uint8), col[keep].view(np.uint8)), desc + ": moments copy"
    _close(G, got[..., :3], want[..., :3], storage, desc + ": moments colour", colour_abs=2e-5 if storage == "f32" else 1e-3)
    g, w = got[..., 3].astype(np.float64), want[..., 3].astype(np.float64)
    lim = 8e-5 if storage == "f32" else 8e-5 + np.abs(w) * 2.0 ** -10
    assert np.all(np.abs(g - w) <= lim), desc + f": moments variance {np.abs(g - w).max():.3e}"
    # one a-trous iteration
    src = np.concatenate([rng.uniform(-0.2, 1.3, (H, W, 3)), rng.uniform(-0.01, 0.2, (H, W, 1))], -1).astype(dt)
    if poison:
        _sprinkle(rng, src, 6)
    if zeros:
        flat = src.reshape(-1)
        flat[rz.integers(0, flat.size, 8)] = rz.choice(np.array([-0.0, 0.0, np.finfo(dt).tiny / 4, -np.finfo(dt).tiny / 4, np.finfo(dt).tiny], np.float64), 8).astype(dt)
    if zeros == 2:
        _blocks_of_negzero(rz, src, 5)
    want = np.zeros_like(src); fbw = np.full_like(src, 7)
    oracle.atrous(W, H, storage, src, want, fbw, gbuf(fs), step=step, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"], iteration=0)
    out, fb = d.new_colour(), G.dev(np.full_like(src, 7))
    d.FilterKernel(G.dev(src), out, fb, G.gb_dev(fs), step, 0)
    try:
        _close(G, G.host(out), want, storage, desc + ": a-trous")
        _close(G, G.host(fb), fbw, storage, desc + ": a-trous feedback")
        skym = _sky_mask(fs)                                  # the copied texels (:554-558): raw bits, -0.0 included; no feedback for them
        assert _same_bits(G.host(out)[skym], want[skym]), desc + ": a-trous sky copy"
        assert _same_bits(G.host(fb)[skym], fbw[skym]), desc + ": a-trous feedback on sky"
        zero = (want == 0) & (G.host(out) == 0)               # a filtered zero carries the reference's sign (svgf_device.h: commit_px)
        assert np.array_equal(np.signbit(G.host(out)[zero]), np.signbit(want[zero])), desc + ": a-trous sign of zero"
    except AssertionError as e:
        e.ctx = dict(frame=fs, src=src, got=G.host(out), want=want, step=step, denoiser=d, tun=tun)      # (for whoever re-runs the seed by hand: run_trial raises with the planes attached)
        raise
    d.close()
    return desc + (" zeros" if zeros else "")


# ------------------------------------------------------------------------------------------------------------------ strips vs frame driver
def _sequence(rng, W, H, N, mv, poison, storage):
    fr = [synth.make_frame(W, H, k, mv=mv) for k in range(N)]
    if poison:
        for k in range(N):
            if rng.integers(0, 2):
                fr[k] = _poisoned(rng, fr[k], ("motion", "depth", "ddepth", "normal", "id"))
            if rng.integers(0, 2):
                fr[k] = dict(fr[k], radiance=_sprinkle(rng, fr[k]["radiance"].copy(), 4))
    return fr


def trial_strips(G, oracle, seed, wide=False, edge=False):
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    rng = np.random.default_rng(seed)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    tun = _tunables(rng)
    steps = int(rng.choice([5, 5, 5, 3, 1, 2, 4, 6, 7]))
    radius = int(rng.choice([3, 3, 1]))
    if edge:
        re_ = np.random.default_rng(seed ^ 0xA5A5A5)
        tun = _edge_tunables(re_, tun)
        radius, steps = int(re_.integers(0, 4)), int(re_.integers(0, 8))
    world = int(rng.integers(2, 9))
    plan = str(rng.choice(["ghost", "grouped", "per-iteration", "auto"]))
    reach = int(rng.integers(0, 7))
    W = int(rng.choice([int(rng.integers(1, 64)), int(rng.integers(64, 700)), 128, 129, 256]))
    # a height the plan accepts: walk up from a random start
    H = int(rng.integers(world, 1200))
    if wide:
        rz = np.random.default_rng(seed ^ 0x5A5A5A)
        W, H = int(rz.choice([int(rz.integers(1024, 8200)), 1920, 3840, 7680])), int(rz.integers(world, 400))
    for _ in range(40):
        if strips._plan_fits(W, H, 0, world, steps, plan, radius, reach):
            break
        H += int(rng.integers(16, 200))
    else:
        return f"strips seed {seed}: no height found (skipped)"
    mvy = float(rng.uniform(-reach, reach)) if reach else 0.0
    mv = (float(rng.uniform(-4, 4)), mvy)
    poison = bool(rng.integers(0, 2))
    N = int(rng.integers(2, 5))
    edge_first, own_streams = bool(rng.integers(0, 4)), bool(rng.integers(0, 2))
    desc = (f"strips seed {seed}: {W}x{H} {storage} world {world} plan {plan} reach {reach} mv ({mv[0]:.2f},{mv[1]:.2f}) steps {steps} r{radius} "
            f"poison {poison} frames {N} edge_first {edge_first} own_streams {own_streams}" + (f" edge {tun}" if edge else ""))
    fr = _sequence(rng, W, H, N, mv, poison, storage)
    P = F.Params(storage=storage, steps=steps, moments_radius=radius, **tun)
    whole = F.Denoiser(W, H, P)
    streams = [torch.cuda.Stream(priority=-1) for _ in range(world)] if own_streams else None
    drv = strips.NativeStrips(W, H, world, P, list(range(world)), [0] * world, streams=[s.cuda_stream for s in streams] if streams else None,
                              plan=plan, motion_reach=reach, transport="mailbox")
    drv.set_edge_first(edge_first)
    try:
        gbs = [G.gb_dev(f) for f in fr]
        prev_in = None
        for k in range(N):
            want = G.host(whole.Render(G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), gbs[k], gbs[k - 1] if k else None))
            torch.cuda.synchronize()
            cur_in = []
            for lay in drv.layouts:
                sl = slice(lay["y0"], lay["y1"])
                cur_in.append((G.dev(np.ascontiguousarray(fr[k]["radiance"][sl].astype(G.NPDT[storage]))),
                               F.GBuffer(*(G.dev(np.ascontiguousarray(fr[k][n][sl])) for n in ("motion", "normal", "uv")))))
            torch.cuda.synchronize()
            outs = drv.frame([c[0] for c in cur_in], [c[1] for c in cur_in], [p[1] for p in prev_in] if prev_in else None)
            drv.sync()
            got = np.concatenate([G.host(drv.owned(r, o)) for r, o in enumerate(outs)], 0)
            if not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
                bad = np.argwhere((got.view(np.uint8) != want.view(np.uint8)).reshape(H, W, -1).any(-1))
                raise AssertionError(desc + f": frame {k}: {len(bad)} px differ, first {bad[:4].tolist()}")
            prev_in = cur_in
        hist = np.concatenate([G.host(drv.owned(r, drv.state_plane(r, F.PLANE_HISTORY, 1 - drv.pingpong(r)))) for r in range(world)], 0)
        assert np.array_equal(hist, G.host(whole.state_plane(F.PLANE_HISTORY, 1 - whole.pingpong()))), desc + ": history"
    finally:
        drv.close()
        whole.close()
    return desc



_LOOP_COMM = []


def _loop_comm():
    """ONE RCCL communicator of world size 1 for the process (every send / recv of the virtual ranks has communicator rank 0 as its peer)."""
    if not _LOOP_COMM:
        from svgf_amd import strips
        _LOOP_COMM.append(strips.rccl_comm(1, 0, 0))
    return _LOOP_COMM[0]


def trial_strips2(G, oracle, seed):
    """The strip driver under the settings `strips` leaves alone: RCCL's own kernels over the loop-back communicator or the mailbox, one or two
    frames in flight (results read one call later, no synchronisation in between), svgf_set_prev_guide, iterations 0 + 1 in one launch (finite
    input) — against the plain frame driver, bit for bit."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    rng = np.random.default_rng(seed)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    tun = _tunables(rng)
    steps = int(rng.choice([5, 5, 3, 2, 4, 6, 7]))
    world = int(rng.integers(2, 9))
    plan = str(rng.choice(["ghost", "grouped", "per-iteration", "auto"]))
    reach = int(rng.integers(0, 7))
    W = int(rng.choice([int(rng.integers(1, 64)), int(rng.integers(64, 700)), 128, 129, 256]))
    H = int(rng.integers(world, 1200))
    for _ in range(40):
        if strips._plan_fits(W, H, 0, world, steps, plan, 3, reach):
            break
        H += int(rng.integers(16, 200))
    else:
        return f"strips2 seed {seed}: no height found (skipped)"
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-reach, reach)) if reach else 0.0)
    poison = bool(rng.integers(0, 2))
    N = int(rng.integers(3, 6))
    transport = str(rng.choice(["mailbox", "rccl"]))
    in_flight = int(rng.integers(1, 3))
    prev_guide = bool(rng.integers(0, 2))
    fusion = bool(rng.integers(0, 2)) and not poison and steps >= 2
    desc = (f"strips2 seed {seed}: {W}x{H} {storage} world {world} plan {plan} reach {reach} steps {steps} poison {poison} frames {N} {transport} "
            f"in flight {in_flight} prev_guide {prev_guide} fusion {fusion}")
    fr = _sequence(rng, W, H, N, mv, poison, storage)
    P = F.Params(storage=storage, steps=steps, **tun)
    whole = F.Denoiser(W, H, P)
    side = torch.cuda.Stream(priority=-1)
    kw = dict(comms=[_loop_comm()], loopback=True) if transport == "rccl" else dict(transport="mailbox")
    drv = strips.NativeStrips(W, H, world, P, list(range(world)), [0] * world, streams=[side.cuda_stream] * world, plan=plan, motion_reach=reach, **kw)
    try:
        drv.set_frames_in_flight(in_flight)
        if prev_guide:
            drv.set_prev_guide(True)
        if fusion:
            drv.set_iteration_fusion(True)
        gbs = [G.gb_dev(f) for f in fr]
        want = [G.host(whole.Render(G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), gbs[k], gbs[k - 1] if k else None)) for k in range(N)]
        torch.cuda.synchronize()
        inputs = []
        for k in range(N):                                # every frame's planes stay alive and untouched
            cur = []
            for lay in drv.layouts:
                sl = slice(lay["y0"], lay["y1"])
                cur.append((G.dev(np.ascontiguousarray(fr[k]["radiance"][sl].astype(G.NPDT[storage]))),
                            F.GBuffer(*(G.dev(np.ascontiguousarray(fr[k][n][sl])) for n in ("motion", "normal", "uv")))))
            inputs.append(cur)
        torch.cuda.synchronize()
        outs, got = {}, {}

        def collect(k):                                   # on the ranks' compute stream: ordered behind frame k by call k + 1
            with torch.cuda.stream(side):
                got[k] = [drv.owned(r, o).clone() for r, o in enumerate(outs[k])]
        for k in range(N):
            outs[k] = drv.frame([c[0] for c in inputs[k]], [c[1] for c in inputs[k]], [p[1] for p in inputs[k - 1]] if k else None)
            if in_flight == 1:
                collect(k)
            elif k >= 1:
                collect(k - 1)
        drv.sync()
        if in_flight == 2:
            collect(N - 1)
        torch.cuda.synchronize()
        for k in range(N):
            g = np.concatenate([G.host(t) for t in got[k]], 0)
            if not np.array_equal(g.view(np.uint8), want[k].view(np.uint8)):
                bad = np.argwhere((g.view(np.uint8) != want[k].view(np.uint8)).reshape(H, W, -1).any(-1))
                raise AssertionError(desc + f": frame {k}: {len(bad)} px differ, first {bad[:4].tolist()}")
        hist = np.concatenate([G.host(drv.owned(r, drv.state_plane(r, F.PLANE_HISTORY, 1 - drv.pingpong(r)))) for r in range(world)], 0)
        assert np.array_equal(hist, G.host(whole.state_plane(F.PLANE_HISTORY, 1 - whole.pingpong()))), desc + ": history"
    finally:
        drv.close()
        whole.close()
    return desc


# ------------------------------------------------------------------------------------------------------------------ frame driver settings
def _edge_tunables(rz, tun):
    pick = lambda vals: vals[int(rz.integers(0, len(vals)))]       # noqa: E731
    return dict(phi_colour=pick([0.0, 1e-6, 1e6, tun["phi_colour"]]), phi_normal=pick([0.0, 1e-3, 512.0, tun["phi_normal"]]),
                depth_threshold=pick([0.0, -1.0, float("inf"), float("nan"), tun["depth_threshold"]]),
                normal_threshold=pick([-1.0, 0.0, 1.0, 2.0, float("nan"), tun["normal_threshold"]]),
                history_base=pick([0, 1, 2, 255, 256, 1000, -5]), mesh_id_test=tun["mesh_id_test"])


def trial_driver(G, oracle, seed, edge=False):
    import torch
    from svgf_amd import filter as F
    rng = np.random.default_rng(seed)
    W, H = _size(rng)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    tun = _tunables(rng)
    rz = np.random.default_rng(seed ^ 0x5A5A5A)
    if edge:
        tun = _edge_tunables(rz, tun)
    steps = int(rng.choice([5, 5, 3, 0, 1, 2, 7]))
    radius = int(rng.choice([3, 3, 1]))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    poison = bool(rng.integers(0, 2))
    N = int(rng.integers(3, 7))
    setting = str(rng.choice(["in_flight", "general", "no_adaptive", "prev_guide", "stage_calls"]))
    if edge:
        radius, steps = int(rz.integers(0, 4)), int(rz.integers(0, 11))          # (the GUI: 0-10 iterations, GUI.cpp:988)
    desc = f"driver seed {seed}: {W}x{H} {storage} steps {steps} r{radius} poison {poison} frames {N} setting {setting}" + (f" edge {tun}" if edge else "")
    fr = _sequence(rng, W, H, N, mv, poison, storage)
    P = F.Params(storage=storage, steps=steps, moments_radius=radius, **tun)
    a = F.Denoiser(W, H, P)
    b = F.Denoiser(W, H, F.Params(storage=storage, steps=steps, moments_radius=radius, variant="lds-general", **tun) if setting == "general" else P)
    stages = G.HipPipeline(W, H, storage, steps=steps, moments_radius=radius, **tun) if setting == "stage_calls" else None
    if setting == "in_flight":
        b.set_frames_in_flight(2)
    elif setting == "no_adaptive":
        b.set_adaptive_moments(False)
    elif setting == "prev_guide":
        b.set_prev_guide(True)
    ref = oracle.Pipeline(W, H, storage, steps=steps, nthreads=8, moments_radius=radius, **tun)
    gbs = [G.gb_dev(f) for f in fr]
    rads = [G.dev(f["radiance"].astype(G.NPDT[storage])) for f in fr]
    try:
        for k in range(N):
            kp = max(k - 1, 0)
            x = G.host(a.Render(rads[k], gbs[k], gbs[kp] if k else None))
            if stages is not None:
                y = stages.frame(fr[k]["radiance"], gbs[k], gbs[kp])    # svgf_temporal / svgf_moments / svgf_atrous on caller-owned planes
            else:
                y = b.Render(rads[k], gbs[k], gbs[kp] if k else None)
                if setting == "in_flight":
                    b.flush()                                         # the view is ordered on the context's stream by the next Render / flush / sync
                y = G.host(y)
            torch.cuda.synchronize()
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), desc + f": frame {k}"
            ref.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp]))
            if not poison:
                # the accept / reject masks of a free-running device sequence equal the oracle's (NaN-free inputs: a poisoned colour plane may
                # flip a comparison the two round differently; the stage-wise trials cover those from identical inputs)
                assert np.array_equal(G.host(a.state_plane(F.PLANE_HISTORY, 1 - a.pingpong())), ref.taps["hist"]), desc + f": frame {k}: history vs oracle"
    finally:
        a.close()
        b.close()
    return desc



# ------------------------------------------------------------------------------------------------------------------ row ranges
def trial_rows(G, oracle, seed):
    """A stage call restricted to a row range (svgf_set_rows) writes those rows with the whole-frame call's bits and touches nothing else."""
    import torch
    from svgf_amd import filter as F
    rng = np.random.default_rng(seed)
    W, H = _size(rng)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    dt = CDT[storage]
    tun = _tunables(rng)
    variant = str(rng.choice(["auto", "lds", "direct", "lds-general"]))
    radius = int(rng.choice([3, 3, 1]))
    step = int(2 ** rng.integers(0, 7))
    poison = bool(rng.integers(0, 2))
    r0 = int(rng.integers(0, H))
    r1 = int(rng.integers(r0 + 1, H + 1))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    desc = f"rows seed {seed}: {W}x{H} {storage} {variant} r{radius} step {step} poison {poison} rows [{r0},{r1})"
    f0, f1 = synth.make_frame(W, H, seed % 89, mv=mv), synth.make_frame(W, H, seed % 89 + 1, mv=mv)
    if poison:
        f0, f1 = _poisoned(rng, f0, ("motion", "depth", "ddepth", "normal", "id")), _poisoned(rng, f1, ("motion", "depth", "ddepth", "normal", "id"))
    d = F.Denoiser(W, H, F.Params(storage=storage, moments_radius=radius, variant=variant, **tun))
    g0, g1 = G.gb_dev(f0), G.gb_dev(f1)
    src = np.concatenate([rng.uniform(-0.2, 1.3, (H, W, 3)), rng.uniform(-0.01, 0.2, (H, W, 1))], -1).astype(dt)
    momp = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hl = rng.integers(0, 9, (H, W)).astype(np.uint8)
    if poison:
        _sprinkle(rng, src, 6), _sprinkle(rng, momp, 3)
    _sprinkle_zeros(rng, src, 6), _sprinkle_zeros(rng, momp, 3)
    s_d, m_d, h_d = G.dev(src), G.dev(momp), G.dev(hl)

    def both(call, nout):
        """call(outs) once on the whole frame and once on [r0, r1) into planes pre-filled with a sentinel"""
        whole = [torch.full_like(t, 7) for t in nout]
        call(whole)
        part = [torch.full_like(t, 7) for t in nout]
        d.set_rows(r0, r1)
        call(part)
        d.set_rows()
        torch.cuda.synchronize()
        for i, (w_, p_) in enumerate(zip(whole, part)):
            w_, p_ = G.host(w_), G.host(p_)
            assert np.array_equal(p_[r0:r1].view(np.uint8), w_[r0:r1].view(np.uint8)), desc + f": output {i}: rows inside the range"
            rest = np.concatenate([p_[:r0], p_[r1:]], 0)
            assert (rest == 7).all(), desc + f": output {i}: rows outside the range were written"
    col, mo, hi = d.new_colour(), d.new_moments(), d.new_history()
    both(lambda o: d.FilterKernel(s_d, o[0], o[1], g1, step, 0), [col, col])
    both(lambda o: d.FilterMoments(s_d, o[0], m_d, g1, h_d), [col])
    both(lambda o: d.TemporalFilter(s_d, G.dev(src[::-1].copy()), o[0], g1, g0, h_d, o[1], o[2], m_d), [col, hi, mo])
    d.close()
    return desc


# ------------------------------------------------------------------------------------------------------------------ the pair launch
def trial_pair(G, oracle, seed):
    """Iterations 0 and 1 in one launch: finite input — the frame driver with the fusion == without, bit for bit, state planes included;
    poisoned input — svgf_atrous_pair against the oracle's two iterations within the pair launch's tolerances."""
    import torch
    from svgf_amd import filter as F
    rng = np.random.default_rng(seed)
    W, H = _size(rng)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    dt = CDT[storage]
    tun = _tunables(rng)
    poison = bool(rng.integers(0, 3) == 0)
    variant = str(rng.choice(["auto", "lds-general"]))
    desc = f"pair seed {seed}: {W}x{H} {storage} {variant} poison {poison}"
    if poison:
        fs = _poisoned(rng, synth.make_frame(W, H, seed % 83), ("depth", "ddepth", "normal"))
        src = np.concatenate([rng.uniform(-0.2, 1.3, (H, W, 3)), rng.uniform(-0.01, 0.2, (H, W, 1))], -1).astype(dt)
        _sprinkle(rng, src, 4)
        mid = np.zeros_like(src); want = np.zeros_like(src); want_fb = np.full_like(src, 7)
        oracle.atrous(W, H, storage, src, mid, want_fb, gbuf(fs), step=1, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"], iteration=0)
        oracle.atrous(W, H, storage, mid, want, None, gbuf(fs), step=2, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"], iteration=1)
        d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant, **tun))
        out, fb = d.new_colour(), G.dev(np.full_like(src, 7))
        d.FilterKernelPair(G.dev(src), out, fb, G.gb_dev(fs))
        _close(G, G.host(fb), want_fb, storage, desc + ": pair feedback")
        # iteration 1's result: against the device's own two launches (the pair launch takes the exact form for every pixel of a band that
        # holds a NaN, the two launches only for the pixels whose fast result does: the same values up to the rounding of iteration 0, which
        # iteration 1's weights amplify — by up to ~50x at the small PhiColour the sweep draws; fp16 storage: a half-ulp flip of iteration 0)
        mid_d, out2 = d.new_colour(), d.new_colour()
        d.FilterKernel(G.dev(src), mid_d, None, G.gb_dev(fs), 1, 1)
        d.FilterKernel(mid_d, out2, None, G.gb_dev(fs), 2, 1)
        got, two = G.host(out).astype(np.float32), G.host(out2).astype(np.float32)
        assert np.array_equal(np.isnan(got), np.isnan(two)) and np.array_equal(np.isnan(got), np.isnan(want.astype(np.float32))), desc + ": pair result: NaN masks"
        with np.errstate(all="ignore"):
            err = np.abs(np.nan_to_num(got, posinf=0, neginf=0) - np.nan_to_num(two, posinf=0, neginf=0)).max()
        amp = max(1.0, 50.0 / tun["phi_colour"])         # (iteration 1's weights: exp(-|dl| / (PhiColour * sqrt(variance))) of iteration 0's last bit)
        assert err <= (2e-4 * amp if storage == "f32" else 2e-2), desc + f": pair against two launches: {err:.3e} (PhiColour {tun['phi_colour']:.3g})"
        d.close()
        return desc
    steps = int(rng.choice([2, 3, 5, 7]))
    N = int(rng.integers(2, 6))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    desc += f" steps {steps} frames {N}"
    fr = [synth.make_frame(W, H, k, mv=mv) for k in range(N)]
    P = F.Params(storage=storage, steps=steps, variant=variant, **tun)
    a, b = F.Denoiser(W, H, P), F.Denoiser(W, H, P)
    a.set_iteration_fusion(True)
    gbs = [G.gb_dev(f) for f in fr]
    for k in range(N):
        rad = G.dev(fr[k]["radiance"].astype(G.NPDT[storage]))
        ra, rb = a.Render(rad, gbs[k], gbs[k - 1] if k else None), b.Render(rad, gbs[k], gbs[k - 1] if k else None)
        torch.cuda.synchronize()
        assert torch.equal(ra.view(torch.uint8), rb.view(torch.uint8)), desc + f": frame {k}"
        for plane in (F.PLANE_COLOUR, F.PLANE_MOMENTS, F.PLANE_HISTORY):
            assert torch.equal(a.state_plane(plane, 1 - a.pingpong()).view(torch.uint8), b.state_plane(plane, 1 - b.pingpong()).view(torch.uint8)), desc + f": frame {k}: plane {plane}"
    a.close(); b.close()
    return desc


# ------------------------------------------------------------------------------------------------------------------ the stages after the path
def trial_post(G, oracle, seed):
    """TAA + sRGB against the oracle (and the LDS-tiled kernel against the per-pixel one, bit for bit); albedo (de)modulation bit-exact."""
    from svgf_amd import filter as F
    from tests.helpers import half_ulp_diff
    rng = np.random.default_rng(seed)
    W, H = _size(rng)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    dt = CDT[storage]
    poison = bool(rng.integers(0, 2))
    desc = f"post seed {seed}: {W}x{H} {storage} poison {poison}"
    filt = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
    hist = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    if poison:
        _sprinkle(rng, filt, 8), _sprinkle(rng, hist, 6)
    _sprinkle_zeros(rng, filt, 6), _sprinkle_zeros(rng, hist, 4)
    want = np.zeros_like(filt)
    oracle.taa(W, H, storage, filt, hist, want)
    outs = []
    for variant in ("auto", "direct"):
        d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant))
        out = d.new_colour()
        d.TAA(G.dev(filt), G.dev(hist), out)
        outs.append(G.host(out))
        if variant == "direct":
            al = _sprinkle_zeros(rng, rng.uniform(-0.1, 1.0, (H, W, 4)).astype(dt), 4)
            if poison:
                _sprinkle(rng, al, 4)                                             # (a NaN albedo reads as the floor: fmaxf)
            for mode, fn in ((0, d.Demodulate), (1, d.Modulate)):
                w2 = np.zeros_like(filt)
                oracle.albedo(mode, W, H, storage, filt, al, w2)
                o2 = d.new_colour()
                fn(G.dev(filt), G.dev(al), o2)
                g2 = G.host(o2)
                nn = ~np.isnan(w2.astype(np.float32))                             # (a NaN the operation itself makes — 0 x inf — has the sign the machine gives it)
                assert np.array_equal(np.isnan(g2.astype(np.float32)), ~nn) and np.array_equal(g2[nn].view(np.uint8), w2[nn].view(np.uint8)), desc + f": albedo mode {mode}"
        d.close()
    got = outs[0]
    assert np.array_equal(got.view(np.uint8), outs[1].view(np.uint8)), desc + ": TAA tiled vs per-pixel"
    assert not np.isnan(got.astype(np.float32)).any(), desc
    black_w, black_g = (want[..., :3].astype(np.float32) == 0).all(-1), (got[..., :3].astype(np.float32) == 0).all(-1)
    assert np.array_equal(black_w, black_g), desc + ": black (NaN-guarded) pixels differ"
    if storage == "f32":
        assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 2e-6, desc + ": TAA"
    else:
        assert half_ulp_diff(got, want).max() <= 1, desc + ": TAA"
    return desc



def trial_driver2(G, oracle, seed):
    """The frame driver across svgf_reset_history, svgf_resize (to the same or another size) and svgf_set_params between frames: a plain driver, a
    driver under a random setting (two frames in flight, svgf_set_prev_guide, young-pixel launch only, the fusion on finite input) given the
    same calls, and — from the last restart on — a FRESH context: all bit for bit, history and moments planes included."""
    import torch
    from svgf_amd import filter as F
    rng = np.random.default_rng(seed)
    sizes = [_size(rng), _size(rng)]
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    tuns = [_tunables(rng), _tunables(rng)]
    steps = [int(rng.choice([5, 3, 0, 1, 2, 7])), int(rng.choice([5, 4, 2]))]
    radius = int(rng.choice([3, 3, 1]))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    poison = bool(rng.integers(0, 2))
    N = int(rng.integers(4, 9))
    setting = str(rng.choice(["in_flight", "prev_guide", "no_adaptive", "fusion", "plain"]))
    if setting == "fusion" and poison:
        setting = "plain"
    events = {int(k): str(rng.choice(["reset", "resize_same", "resize_other", "params"])) for k in rng.choice(np.arange(1, N), size=int(rng.integers(1, 4)), replace=False)}
    desc = f"driver2 seed {seed}: {sizes} {storage} steps {steps} r{radius} poison {poison} frames {N} setting {setting} events {dict(sorted(events.items()))}"
    P = [F.Params(storage=storage, steps=steps[i], moments_radius=radius, **tuns[i]) for i in range(2)]
    si, pi = 0, 0
    W, H = sizes[0]
    x, y = F.Denoiser(W, H, P[0]), F.Denoiser(W, H, P[0])
    z = None                                              # the fresh context of the last restart

    def configure(d):
        if setting == "in_flight":
            d.set_frames_in_flight(2)
        elif setting == "prev_guide":
            d.set_prev_guide(True)
        elif setting == "no_adaptive":
            d.set_adaptive_moments(False)
        elif setting == "fusion":
            d.set_iteration_fusion(True)
    configure(y)
    frames_of = {}
    try:
        prev_gb = None
        for k in range(N):
            ev = events.get(k)
            if ev == "reset":
                x.reset_history(); y.reset_history()
            elif ev in ("resize_same", "resize_other"):
                if ev == "resize_other":
                    si ^= 1
                W, H = sizes[si]
                x.Resize(W, H); y.Resize(W, H)
                prev_gb = None if ev == "resize_other" else prev_gb
            elif ev == "params":
                pi ^= 1
                x.set_params(P[pi]); y.set_params(P[pi])
                if z is not None:
                    z.set_params(P[pi])
            if ev in ("reset", "resize_same", "resize_other"):
                if z is not None:
                    z.close()
                z = F.Denoiser(W, H, P[pi])
            if (W, H) not in frames_of:
                frames_of[(W, H)] = _sequence(np.random.default_rng(seed + 7 * W + H), W, H, N, mv, poison, storage)
            f = frames_of[(W, H)][k]
            gb, rad = G.gb_dev(f), G.dev(f["radiance"].astype(G.NPDT[storage]))
            a = G.host(x.Render(rad, gb, prev_gb))
            b = y.Render(rad, gb, prev_gb)
            if setting == "in_flight":
                y.flush()
            b = G.host(b)
            torch.cuda.synchronize()
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), desc + f": frame {k}: setting"
            if z is not None:
                c = G.host(z.Render(rad, gb, prev_gb))
                assert np.array_equal(a.view(np.uint8), c.view(np.uint8)), desc + f": frame {k}: against the fresh context"
                for plane in (F.PLANE_HISTORY, F.PLANE_MOMENTS):
                    assert torch.equal(x.state_plane(plane, 1 - x.pingpong()).view(torch.uint8), z.state_plane(plane, 1 - z.pingpong()).view(torch.uint8)), desc + f": frame {k}: plane {plane}"
            prev_gb = gb
    finally:
        x.close(); y.close()
        if z is not None:
            z.close()
    return desc



def trial_graph(G, oracle, seed):
    """svgf_denoise_frame recorded into a HIP graph (two frames per graph, inputs at fixed addresses refilled before each replay) against the
    directly enqueued frames, bit for bit, state planes included (tests/test_gpu_graph.py's harness on random sizes and settings)."""
    from tests.test_gpu_graph import _assert_same, _direct, _replayed
    rng = np.random.default_rng(seed)
    W, H = _size(rng)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    steps = int(rng.choice([5, 5, 3, 0, 1, 2, 7]))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    poison = bool(rng.integers(0, 2))
    warm = int(rng.choice([4, 6]))
    N = warm + 2 * int(rng.integers(1, 4))
    kw = dict(steps=steps, variant=str(rng.choice(["auto", "direct", "lds-general"])), prev_guide=bool(rng.integers(0, 2)))
    kw["fusion"] = bool(rng.integers(0, 2)) and not poison and steps >= 2 and kw["variant"] != "direct"
    in_flight = int(rng.integers(1, 3))
    desc = f"graph seed {seed}: {W}x{H} {storage} poison {poison} frames {N} warm {warm} in flight {in_flight} {kw}"
    seq = _sequence(rng, W, H, N, mv, poison, storage)
    try:
        _assert_same(_direct(G, seq, storage, **kw), _replayed(G, seq, storage, warm=warm, in_flight=in_flight, **kw))
    except AssertionError as e:
        raise AssertionError(desc + ": " + str(e)) from e
    return desc



_FULL = {}


def trial_fullsize(G, oracle, seed):
    """BASELINE.json's frame sizes themselves (1920x1080, 3840x2160): the strip driver (mailbox, world 2-8, any plan) and the frame driver under a
    random setting against the plain frame driver, bit for bit, on poisoned frames; the history against the oracle's would need minutes of CPU
    per trial and is left to tests/test_gpu_round2.py's full-size cases.  (The frames of a size are made once per process: synth needs seconds.)"""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    rng = np.random.default_rng(seed)
    W, H = [(1920, 1080), (3840, 2160)][int(rng.integers(0, 2))]
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    tun = _tunables(rng)
    steps = int(rng.choice([5, 5, 3, 7]))
    world = int(rng.integers(2, 9))
    plan = str(rng.choice(["ghost", "grouped", "per-iteration", "auto"]))
    mv = [(0.0, 0.0), (1.5, -2.5)][int(rng.integers(0, 2))]
    reach = 3 if mv[1] else int(rng.integers(0, 3))
    setting = str(rng.choice(["in_flight", "general", "no_adaptive", "prev_guide"]))
    N = 3
    desc = f"fullsize seed {seed}: {W}x{H} {storage} steps {steps} world {world} plan {plan} reach {reach} mv {mv} setting {setting}"
    if not strips._plan_fits(W, H, 0, world, steps, plan, 3, reach):
        return desc + " (plan does not fit: skipped)"
    key = (W, H, mv)
    if key not in _FULL:
        _FULL.clear()                                     # one size at a time in memory
        _FULL[key] = [synth.make_frame(W, H, k, mv=mv) for k in range(N)]
    fr = []
    for k in range(N):
        f = _FULL[key][k]
        f = _poisoned(rng, f, ("motion", "depth", "ddepth", "normal", "id")) if rng.integers(0, 2) else f
        fr.append(dict(f, radiance=_sprinkle(rng, f["radiance"].copy(), 6)))
    P = F.Params(storage=storage, steps=steps, **tun)
    whole, other = F.Denoiser(W, H, P), F.Denoiser(W, H, F.Params(storage=storage, steps=steps, variant="lds-general", **tun) if setting == "general" else P)
    if setting == "in_flight":
        other.set_frames_in_flight(2)
    elif setting == "no_adaptive":
        other.set_adaptive_moments(False)
    elif setting == "prev_guide":
        other.set_prev_guide(True)
    drv = strips.NativeStrips(W, H, world, P, list(range(world)), [0] * world, plan=plan, motion_reach=reach, transport="mailbox")
    try:
        gbs = [G.gb_dev(f) for f in fr]
        prev_in = None
        for k in range(N):
            rad = G.dev(fr[k]["radiance"].astype(G.NPDT[storage]))
            want = whole.Render(rad, gbs[k], gbs[k - 1] if k else None).clone()
            y = other.Render(rad, gbs[k], gbs[k - 1] if k else None)
            if setting == "in_flight":
                other.flush()
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.uint8), want.view(torch.uint8)), desc + f": frame {k}: setting"
            cur_in = []
            for lay in drv.layouts:
                sl = slice(lay["y0"], lay["y1"])
                cur_in.append((rad[sl].contiguous(), F.GBuffer(gbs[k].motion[sl].contiguous(), gbs[k].normal[sl].contiguous(), gbs[k].uv[sl].contiguous())))
            torch.cuda.synchronize()
            outs = drv.frame([c[0] for c in cur_in], [c[1] for c in cur_in], [p[1] for p in prev_in] if prev_in else None)
            drv.sync()
            got = torch.cat([drv.owned(r, o) for r, o in enumerate(outs)], 0)
            assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), desc + f": frame {k}: strips"
            prev_in = cur_in
    finally:
        drv.close(); whole.close(); other.close()
    return desc


TRIALS = {"fullsize": trial_fullsize, "graph": trial_graph, "edgestrips": lambda G, oracle, seed: trial_strips(G, oracle, seed, edge=True), "edgedriver": lambda G, oracle, seed: trial_driver(G, oracle, seed, edge=True), "edge": lambda G, oracle, seed: trial_stage(G, oracle, seed, edge=True), "wide": lambda G, oracle, seed: trial_stage(G, oracle, seed, wide=True), "widestrips": lambda G, oracle, seed: trial_strips(G, oracle, seed, wide=True),
          "driver2": trial_driver2, "strips2": trial_strips2, "stage0": lambda G, oracle, seed: trial_stage(G, oracle, seed, zeros=True), "negzero": lambda G, oracle, seed: trial_stage(G, oracle, seed, zeros=2), "stage": trial_stage, "strips": trial_strips, "driver": trial_driver, "rows": trial_rows, "pair": trial_pair, "post": trial_post}


def run_trial(kind, seed, G=None, oracle=None):
    if G is None:
        from tests import gpu_helpers as G                    # noqa: N813
    if oracle is None:
        from oracle import oracle as oracle                   # noqa: PLW0127
    return TRIALS[kind](G, oracle, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--kinds", default=",".join(KINDS))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available(), "needs an MI355X"
    from oracle import oracle
    from tests import gpu_helpers as G                        # noqa: N813
    oracle.build()
    kinds = args.kinds.split(",")
    t_end = time.monotonic() + args.minutes * 60
    done, failed, lines = {k: 0 for k in kinds}, [], []
    seed = args.seed
    while time.monotonic() < t_end:
        kind = kinds[seed % len(kinds)]
        try:
            desc = run_trial(kind, seed, G, oracle)
            lines.append("ok   " + desc)
        except Exception as e:  # noqa: BLE001
            failed.append((kind, seed))
            msg = str(e).splitlines()[0][:400] if str(e) else type(e).__name__
            lines.append(f"FAIL {kind} seed {seed}: {type(e).__name__}: {msg}")
            if not isinstance(e, AssertionError):
                lines.append(traceback.format_exc())
            torch.cuda.synchronize()
        done[kind] += 1
        seed += 1
    summary = f"fuzz_parity: seeds {args.seed}..{seed - 1}: " + ", ".join(f"{k} {n}" for k, n in done.items()) + f"; failed {len(failed)}: {failed}"
    text = "\n".join(lines + [summary])
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text + "\n")
    print("\n".join([ln for ln in lines if not ln.startswith("ok")] + [summary]))
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
