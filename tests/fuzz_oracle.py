"""Seeded random sweep of the C++ oracle against the independently written NumPy restatement (oracle/svgf_numpy.py) — CPU only, test
infrastructure.  The reference holds no vectors for this path, so the two restatements of SURVEY.md Appendix A checking each other over
random sizes, tunables, motions and poisoned texels is the widest pin the oracle can have here.

    python -m tests.fuzz_oracle --minutes 5 --seed 1 [--out file]

A trial (a pure function of its seed; tests/test_oracle_fuzz.py pins some): frame size 1 x 1 .. 220 x 120, storage, the tunables over the GUI's
ranges, NaN / inf / -0 / denormal colour texels and poisoned G-buffer texels (tests/gbuffer_poison.py) with probability 1/2; temporal bit for
bit (history = the accept / reject mask, colour, moments); moments, one a-trous iteration and TAA within the two libms' distance (NaN masks
and infinities identical); albedo and the G-buffer adapter (GBuffer.frag:62-88, from random attribute planes and cameras) bit for bit."""
from __future__ import annotations

import argparse
import sys
import time
import traceback

import numpy as np

from oracle import svgf_numpy as snp
from svgf_amd import synth
from tests.fuzz_parity import _poisoned, _sprinkle, _sprinkle_zeros, _tunables
from tests.helpers import CDT, gbuf, half_ulp_diff


def _size(rng):
    c = rng.integers(0, 8)
    if c == 0:
        return int(rng.integers(1, 9)), int(rng.integers(1, 9))
    if c == 1:
        return int(rng.choice([63, 64, 65, 127, 128, 129])), int(rng.integers(1, 40))
    return int(rng.integers(9, 220)), int(rng.integers(9, 120))


def _same_nonfinite(a, b, what):
    a32, b32 = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert np.array_equal(np.isnan(a32), np.isnan(b32)), f"{what}: NaN masks differ ({int(np.isnan(a32).sum())} vs {int(np.isnan(b32).sum())}; first at {np.argwhere(np.isnan(a32) != np.isnan(b32))[:3].tolist()})"
    inf = np.isinf(b32)
    assert np.array_equal(a32[inf], b32[inf]), f"{what}: infinities differ"
    return np.isfinite(b32) & np.isfinite(a32)


def _near(got, want, storage, what, f32_abs, f32_rel=0.0):
    fin = _same_nonfinite(got, want, what)
    g, w = got.astype(np.float64), want.astype(np.float64)
    if storage == "f32":
        err = np.abs(g[fin] - w[fin])
        lim = f32_abs + f32_rel * np.abs(w[fin])
        assert np.all(err <= lim), f"{what}: max err {err.max():.3e}"
    else:
        d = half_ulp_diff(got[fin], want[fin])
        assert d.size == 0 or d.max() <= 1, f"{what}: {d.max()} half-ulps"
        assert (d > 0).sum() <= max(2e-3 * d.size, 2), f"{what}: {(d > 0).mean():.2e} of values differ by one half-ulp"


def run_trial(seed, oracle=None):
    if oracle is None:
        from oracle import oracle as oracle                   # noqa: PLW0127
    rng = np.random.default_rng(seed)
    W, H = _size(rng)
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    dt = CDT[storage]
    tun = _tunables(rng)
    radius = int(rng.choice([3, 3, 1, 2]))
    step = int(2 ** rng.integers(0, 7))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    poison = bool(rng.integers(0, 2))
    desc = f"oracle seed {seed}: {W}x{H} {storage} r{radius} step {step} poison {poison}"
    f0, f1 = synth.make_frame(W, H, seed % 97, mv=mv), synth.make_frame(W, H, seed % 97 + 1, mv=mv)
    if poison:
        f0, f1 = _poisoned(rng, f0, ("motion", "depth", "ddepth", "normal", "id")), _poisoned(rng, f1, ("motion", "depth", "ddepth", "normal", "id"))
    # ---- temporal: bit for bit
    prev = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
    mom_prev = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hist_prev = rng.integers(0, 256, (H, W)).astype(np.uint8)
    cur = rng.uniform(-0.1, 1.4, (H, W, 4)).astype(dt)
    if poison:
        _sprinkle(rng, cur, 5), _sprinkle(rng, prev, 5), _sprinkle(rng, mom_prev, 3)
        _sprinkle_zeros(rng, cur, 5), _sprinkle_zeros(rng, prev, 5), _sprinkle_zeros(rng, mom_prev, 3)
    out = np.zeros_like(cur); hist = np.zeros((H, W), np.uint8); mom = np.zeros((H, W, 2), dt)
    tk = dict(depth_threshold=tun["depth_threshold"], normal_threshold=tun["normal_threshold"], history_base=tun["history_base"], mesh_id_test=tun["mesh_id_test"])
    oracle.temporal(W, H, storage, prev, cur, out, gbuf(f1), gbuf(f0), hist_prev, hist, mom, mom_prev, **tk)
    with np.errstate(all="ignore"):
        w_out, w_hist, w_mom = snp.temporal(prev, cur, gbuf(f1), gbuf(f0), hist_prev, mom_prev, **tk)
    assert np.array_equal(hist, w_hist), desc + f": temporal history ({int((hist != w_hist).sum())} px)"
    u = np.uint32 if storage == "f32" else np.uint16
    for got, want, name in ((out, w_out, "colour"), (mom, w_mom, "moments")):
        fin = _same_nonfinite(got, want, desc + f": temporal {name}")
        assert np.array_equal(got.view(u)[fin], want.view(u)[fin]), desc + f": temporal {name}: finite bits ({int((got.view(u)[fin] != want.view(u)[fin]).sum())} values)"
    # ---- moments
    fs = synth.make_frame(W, H, seed % 97 + 1, mv=mv)
    if poison:
        fs = _poisoned(rng, fs, ("depth", "ddepth", "normal"))
    col = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    momp = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hl = rng.integers(0, 8, (H, W)).astype(np.uint8)
    if poison:
        _sprinkle(rng, col, 4), _sprinkle(rng, momp, 3)
    got = np.zeros_like(col)
    oracle.moments(W, H, storage, col, got, momp, gbuf(fs), hl, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"], radius=radius)
    with np.errstate(all="ignore"):
        want = snp.moments(col, momp, gbuf(fs), hl, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"], radius=radius)
    keep = hl >= 4
    assert np.array_equal(got[keep].view(u), col[keep].view(u)), desc + ": moments copy"
    young = (hl < 4) & (hl > 0)                                                   # (history 0: 4 / 0 — compared through the NaN / inf masks only)
    _same_nonfinite(got, want, desc + ": moments")
    _near(got[young][..., :3], want[young][..., :3], storage, desc + ": moments colour", 2e-6, 1e-5)
    gv, wv = got[young][..., 3].astype(np.float64), want[young][..., 3].astype(np.float64)
    fin = np.isfinite(gv) & np.isfinite(wv)
    assert np.all(np.abs(gv[fin] - wv[fin]) <= (4e-5 if storage == "f32" else 8e-5 + np.abs(wv[fin]) * 2.0 ** -9)), desc + f": moments variance {np.abs(gv[fin] - wv[fin]).max():.3e}"
    # ---- one a-trous iteration
    src = np.concatenate([rng.uniform(-0.2, 1.3, (H, W, 3)), rng.uniform(-0.01, 0.2, (H, W, 1))], -1).astype(dt)
    if poison:
        _sprinkle(rng, src, 6), _sprinkle_zeros(rng, src, 4)
    o = np.zeros_like(src); fb = np.full_like(src, 7)
    oracle.atrous(W, H, storage, src, o, fb, gbuf(fs), step=step, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"], iteration=0)
    with np.errstate(all="ignore"):
        w, wfb = snp.atrous(src, gbuf(fs), step=step, phi_colour=tun["phi_colour"], phi_normal=tun["phi_normal"])
    _near(o, w, storage, desc + ": a-trous", 2e-6, 2e-5)
    assert np.array_equal(fb[wfb].view(u), o[wfb].view(u)) and (fb[~wfb] == 7).all(), desc + ": a-trous feedback"
    # ---- TAA + sRGB, albedo
    filt = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
    histc = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    if poison:
        _sprinkle(rng, filt, 5), _sprinkle(rng, histc, 4), _sprinkle_zeros(rng, filt, 4)
    t = np.zeros_like(filt)
    oracle.taa(W, H, storage, filt, histc, t)
    with np.errstate(all="ignore"):
        tw = snp.taa(filt, histc)
    if storage == "f32":
        # TAA's decode ends in sqrt(r) of a channel r that may cancel to ~1e-7, followed by the sRGB slope of 12.92: a last-bit difference of the
        # two libms' pow(x, 2) / pow(x, 0.5) (each within an ulp) shows as up to 3e-4 in such a texel.  Compared where it is conditioned: the
        # output itself within 1e-6, or — back through sRGB and the square root — the channel r within 3e-7.
        fin = _same_nonfinite(t, tw, desc + ": TAA")
        g64, w64 = t.astype(np.float64), tw.astype(np.float64)
        lin = lambda c: np.where(c <= 0.0031308 * 12.92, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)          # noqa: E731
        ok = (np.abs(g64 - w64) <= 1e-6) | (np.abs(lin(g64) ** 2 - lin(w64) ** 2) <= 3e-7)
        assert ok[fin].all(), desc + f": TAA: {int((~ok[fin]).sum())} values, max err {np.abs(g64 - w64)[fin & ~ok].max():.3e}"
    else:
        _near(t, tw.astype(dt), storage, desc + ": TAA", 1e-6)
    al = _sprinkle_zeros(rng, rng.uniform(-0.1, 1.0, (H, W, 4)).astype(dt), 4)
    if poison:
        _sprinkle(rng, al, 4)                                                     # (a NaN albedo reads as the floor: fmaxf)
    for mode in (0, 1):
        a = np.zeros_like(filt)
        oracle.albedo(mode, W, H, storage, filt, al, a)
        with np.errstate(all="ignore"):
            aw = snp.albedo(mode, filt, al)
        fin = _same_nonfinite(a, aw, desc + f": albedo mode {mode}")
        assert np.array_equal(a.view(u)[fin], aw.view(u)[fin]), desc + f": albedo mode {mode}"
    # the G-buffer adapter (GBuffer.frag:62-88): the C++ twin against the NumPy one, bit for bit (drawn from a generator of its own: the trials above keep their frames)
    _pack_gbuffer_trial(np.random.default_rng(seed ^ 0x6B0FFE), W, H, poison, oracle, desc)
    return desc


def _pack_gbuffer_trial(rg, W, H, poison, oracle, desc):
    def look_at(eye):
        f = -np.asarray(eye, float); f /= np.linalg.norm(f)
        s_ = np.cross(f, (0.0, 1.0, 0.0)); s_ /= np.linalg.norm(s_)
        m = np.eye(4); m[0, :3], m[1, :3], m[2, :3] = s_, np.cross(s_, f), -f
        m[:3, 3] = -m[:3, :3] @ np.asarray(eye, float)
        return m
    fy = 1.0 / np.tan(rg.uniform(0.3, 1.2) / 2)
    proj = np.zeros((4, 4)); proj[0, 0], proj[1, 1], proj[2, 2], proj[2, 3], proj[3, 2] = fy * H / W, fy, -1.002, -0.2002, -1.0
    eye1 = rg.uniform(-1, 1, 3) + np.array([0.0, 0.0, 5.0])
    eye0 = eye1 + rg.uniform(-0.05, 0.05, 3) * rg.integers(0, 2)                      # (half the trials: a static camera)
    cm = lambda m: m.T.astype(np.float32).ravel()                                       # noqa: E731
    vp, pvp = cm(proj @ look_at(eye1)), cm(proj @ look_at(eye0))
    pos = np.concatenate([rg.uniform(-2, 2, (H, W, 3)), rg.integers(0, 900, (H, W, 1))], -1).astype(np.float32)
    nrm = np.concatenate([rg.normal(size=(H, W, 3)), rg.integers(0, 20, (H, W, 1))], -1).astype(np.float32)
    nrm[rg.uniform(size=(H, W)) < 0.15, :3] = 0                                       # texels without geometry
    bary = np.concatenate([rg.uniform(0, 1, (H, W, 3)), rg.integers(0, 70000, (H, W, 1))], -1).astype(np.float32)    # (ids beyond half's range: inf)
    if poison:                                                                          # non-finite / degenerate attributes; a point in the camera plane (w = 0)
        for plane in (pos, nrm, bary):
            flat = plane.reshape(-1)
            flat[rg.integers(0, flat.size, 4)] = rg.choice(np.array([np.nan, np.inf, -np.inf, -0.0, 1e-42, 3e38], np.float32), 4)
        pos[rg.integers(0, H), rg.integers(0, W), :3] = eye1.astype(np.float32)
    got = oracle.pack_gbuffer(pos, nrm, bary, vp, pvp, eye1.astype(np.float32))
    with np.errstate(all="ignore"):
        want = snp.pack_gbuffer(pos, nrm, bary, vp, pvp, eye1.astype(np.float32))
    for name, g_, w_ in zip(("motion", "normal", "uv"), got, want):
        if g_.dtype == np.float32:
            fin = _same_nonfinite(g_, w_, desc + f": pack_gbuffer {name}")
            assert np.array_equal(g_.view(np.uint32)[fin], w_.view(np.uint32)[fin]), desc + f": pack_gbuffer {name}"
        else:                                                                           # half bits: a NaN is a NaN, everything else bit for bit
            gn, wn = np.isnan(g_.view(np.float16)), np.isnan(w_.view(np.float16))
            assert np.array_equal(gn, wn) and np.array_equal(g_[~gn], w_[~wn]), desc + f": pack_gbuffer {name}"


def run_pipeline_trial(seed, oracle=None):
    """A free-running sequence through BOTH restatements' frame sequencing (oracle.Pipeline, tests/helpers.NumpyPipeline): the history plane — the
    accept / reject mask of every frame — bit for bit; the output within the two libms' distance as the feedback compounds it (finite frames), its
    NaN mask identical (poisoned frames)."""
    from tests.fuzz_parity import _sequence
    from tests.helpers import NumpyPipeline
    if oracle is None:
        from oracle import oracle as oracle                   # noqa: PLW0127
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(1, 90)), int(rng.integers(1, 60))
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    tun = _tunables(rng)
    tun["phi_colour"] = max(tun["phi_colour"], 1.0)           # (below, a free-running sequence amplifies the libms' last bit beyond any useful bound)
    steps = int(rng.choice([5, 3, 0, 1, 2]))
    radius = int(rng.choice([3, 3, 1]))
    mv = (float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)))
    poison = bool(rng.integers(0, 2))
    N = int(rng.integers(3, 7))
    desc = f"pipeline seed {seed}: {W}x{H} {storage} steps {steps} r{radius} poison {poison} frames {N}"
    fr = _sequence(rng, W, H, N, mv, poison, storage)
    a = oracle.Pipeline(W, H, storage, steps=steps, moments_radius=radius, **tun)
    b = NumpyPipeline(W, H, storage, steps=steps, moments_radius=radius, **tun)
    for k in range(N):
        ga, gp = gbuf(fr[k]), gbuf(fr[max(k - 1, 0)])
        oa = a.frame(fr[k]["radiance"], ga, gp)
        with np.errstate(all="ignore"):
            ob = b.frame(fr[k]["radiance"], ga, gp)
        if not poison:
            assert np.array_equal(a.hist[a.P ^ 1], b.taps["hist"]), desc + f": frame {k}: history"
            d = np.abs(oa.astype(np.float64) - ob.astype(np.float64))
            assert d.max() <= (2e-4 if storage == "f32" else 1e-2), desc + f": frame {k}: {d.max():.3e}"      # (fp16: half-ulp flips of five requantised iterations, compounding over the frames)
        else:
            # (a NaN in the colour history reaches the next frames' reprojected colour, never the accept / reject tests: the masks stay equal)
            assert np.array_equal(a.hist[a.P ^ 1], b.taps["hist"]), desc + f": frame {k}: history"
            assert np.array_equal(np.isnan(oa.astype(np.float32)), np.isnan(ob.astype(np.float32))), desc + f": frame {k}: NaN masks"
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=3.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--pipeline", action="store_true", help="free-running sequences through both restatements' frame sequencing instead of the stages")
    args = ap.parse_args()
    from oracle import oracle
    oracle.build()
    t_end = time.monotonic() + args.minutes * 60
    seed, failed, lines = args.seed, [], []
    while time.monotonic() < t_end:
        try:
            lines.append("ok   " + (run_pipeline_trial if args.pipeline else run_trial)(seed, oracle))
        except Exception as e:  # noqa: BLE001
            failed.append(seed)
            lines.append(f"FAIL seed {seed}: {type(e).__name__}: {(str(e).splitlines() or [''])[0][:400]}")
            if not isinstance(e, AssertionError):
                lines.append(traceback.format_exc())
        seed += 1
    summary = f"fuzz_oracle: seeds {args.seed}..{seed - 1}: {seed - args.seed} trials; failed {len(failed)}: {failed}"
    if args.out:
        with open(args.out, "w") as fh:
            fh.write("\n".join(lines + [summary]) + "\n")
    print("\n".join([ln for ln in lines if not ln.startswith("ok")][:60] + [summary]))
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
