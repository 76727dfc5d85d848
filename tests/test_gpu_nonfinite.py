"""GPU parity on NON-FINITE input: NaN, +inf, -inf in the radiance, the history and the filter planes.

The reference's imageLoad / imageStore clamp with glm::clamp = min(max(x, 0), 1) built from `(x < y) ? y : x`
(src/Filter.cuh:63-69,78-83): +-inf clamp to 1 / 0 and a NaN texel STAYS NaN.  It then poisons the history through
`mix` (:398), reaches the wavelet sums channel by channel (:608 — the weight itself stays finite, because
`max(weightLillum, 0.0)` in :424 is CUDA's fmax, which drops a NaN), and turns the zero-weight sums of sky texels
into NaN (0 x NaN, :498-499).  A 1-spp path tracer does produce such texels.  The oracle restates exactly that; the HIP
kernels must reproduce it: same NaN positions, finite values within the stage tolerances (tests/gpu_helpers.py:TOL).

Non-finite and out-of-range G-BUFFER texels (motion, depth, normal, ddepth; include/svgf.h "Non-finite / out-of-range G-buffer texels")
have their own file: tests/test_gpu_gbuffer_nonfinite.py (temporal mask-exact, moments, a-trous LDS + direct, frame and strip drivers)."""
import numpy as np
import pytest

from svgf_amd import synth
from tests.helpers import CDT, frames, gbuf

pytestmark = pytest.mark.gpu

NONFINITE = [np.nan, np.inf, -np.inf]


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


def assert_same_bits_or_nan(got, want, what):
    """Bit-exact stages: identical NaN masks, identical bits everywhere else (a NaN's payload is not compared)."""
    gn, wn = np.isnan(got.astype(np.float32)), np.isnan(want.astype(np.float32))
    assert np.array_equal(gn, wn), f"{what}: NaN masks differ ({gn.sum()} vs {wn.sum()} values)"
    u = np.uint32 if got.dtype == np.float32 else np.uint16
    assert np.array_equal(got.view(u)[~wn], want.view(u)[~wn]), f"{what}: finite values differ"


def assert_close_with_nan(G, got, want, storage, what, colour_abs=None):
    """Toleranced stages: identical non-finite masks (assert_colour_close checks them), finite values within tolerance."""
    gn, wn = np.isnan(got.astype(np.float32)), np.isnan(want.astype(np.float32))
    assert np.array_equal(gn, wn), f"{what}: NaN masks differ ({gn.sum()} vs {wn.sum()} values; first at {np.argwhere(gn != wn)[:4].tolist()})"
    wi = np.isinf(want.astype(np.float32))
    assert np.array_equal(got.astype(np.float32)[wi], want.astype(np.float32)[wi]), f"{what}: infinities differ"
    if colour_abs is None:
        G.assert_colour_close(got, want, storage, what)
    else:
        fin = np.isfinite(want.astype(np.float32))
        assert np.array_equal(np.isfinite(got.astype(np.float32)), fin), f"{what}: finite masks differ"
        d = np.abs(np.where(fin, got.astype(np.float64), 0) - np.where(fin, want.astype(np.float64), 0))
        assert d.max() <= colour_abs, f"{what}: max err {d.max():.3e}"


def poison(rng, plane, region, n_per_kind=6, channels=4, near_sky=True):
    """Scatter NaN / +inf / -inf over `plane` (H, W, C): every kind in every channel on surface pixels, some on sky texels and
    some right next to the sky (the zero-weight sums of sky texels see them).  -> list of (y, x, channel, value)."""
    H, W = region.shape
    surf = np.argwhere(region != synth.SKY)
    sky = np.argwhere(region == synth.SKY)
    # surface pixels within 2 px of a sky texel
    edge = []
    if near_sky and len(sky):
        skym = region == synth.SKY
        grown = np.zeros_like(skym)
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                grown |= np.roll(np.roll(skym, dy, 0), dx, 1)
        edge = np.argwhere(grown & ~skym)
    placed = []
    for v in NONFINITE:
        for ch in range(channels):
            for k in range(n_per_kind):
                pool = surf if (k % 3 == 0 or not len(edge)) else (edge if k % 3 == 1 else (sky if len(sky) else surf))
                y, x = pool[rng.integers(len(pool))]
                plane[y, x, ch] = v
                placed.append((int(y), int(x), ch, v))
    return placed


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_temporal_nonfinite_is_reference_exact(G, oracle, storage):
    """NaN / +-inf in the radiance and in the previous colour / moments: the temporal stage stays bit-exact against the oracle
    (NaN where the oracle has NaN, identical bits elsewhere), with and without motion."""
    from svgf_amd import filter as F
    W, H = 331, 203
    dt = CDT[storage]
    for mv in ((0.0, 0.0), (-2.5, 1.5)):
        rng = np.random.default_rng(11)
        f0, f1 = synth.make_frame(W, H, 3, mv=mv), synth.make_frame(W, H, 4, mv=mv)
        prev = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
        mom_prev = rng.uniform(0, 1, (H, W, 2)).astype(dt)
        hist_prev = rng.integers(0, 40, (H, W)).astype(np.uint8)
        cur = (f1["radiance"] * 1.3 - 0.1).astype(dt)
        poison(rng, cur, f1["region"])
        poison(rng, prev, f0["region"])
        poison(rng, mom_prev, f0["region"], channels=2)
        out = np.zeros_like(cur); hist = np.zeros((H, W), np.uint8); mom = np.zeros((H, W, 2), dt)
        oracle.temporal(W, H, storage, prev, cur, out, gbuf(f1), gbuf(f0), hist_prev, hist, mom, mom_prev,
                        depth_threshold=0.8, normal_threshold=0.9, history_base=24, mesh_id_test=1)
        assert np.isnan(out.astype(np.float32)).any() and np.isnan(mom.astype(np.float32)).any()     # the case is not vacuous
        d = F.Denoiser(W, H, F.Params(storage=storage))
        o_col, o_hist, o_mom = d.new_colour(), d.new_history(), d.new_moments()
        d.TemporalFilter(G.dev(prev), G.dev(cur), o_col, G.gb_dev(f1), G.gb_dev(f0), G.dev(hist_prev), o_hist, o_mom, G.dev(mom_prev))
        assert np.array_equal(G.host(o_hist), hist)
        assert_same_bits_or_nan(G.host(o_col), out, f"temporal colour mv={mv}")
        assert_same_bits_or_nan(G.host(o_mom), mom, f"temporal moments mv={mv}")


@pytest.mark.parametrize("variant", ["direct", "lds", "lds-general"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("radius", [3, 1])
def test_moments_nonfinite(G, oracle, storage, radius, variant):
    """The spatial estimate on planes that hold NaN / +-inf (this stage does not clamp, :450,479): per-pixel kernel (direct), the
    LDS-streaming kernel with and without its uniform-normal form, the wave-shuffle 3x3 kernel.  A sky texel (zero normal: every
    weight exactly 0) next to a NaN becomes NaN (0 x NaN), one with a finite window stays 0."""
    from svgf_amd import filter as F
    W, H = 203, 131
    rng = np.random.default_rng(21)
    f = synth.make_frame(W, H, 0)
    dt = CDT[storage]
    col = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    mom = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hist = rng.integers(1, 8, (H, W)).astype(np.uint8)
    poison(rng, col, f["region"], n_per_kind=4, channels=3)
    poison(rng, mom, f["region"], n_per_kind=3, channels=2)
    want = np.zeros_like(col)
    oracle.moments(W, H, storage, col, want, mom, gbuf(f), hist, phi_colour=10.0, phi_normal=128.0, radius=radius)
    sky_young = (f["region"] == synth.SKY) & (hist < 4)
    wn = np.isnan(want.astype(np.float32))
    assert wn[sky_young].any() and not wn[sky_young].all(), "wanted sky texels with and without a non-finite window"
    d = F.Denoiser(W, H, F.Params(storage=storage, moments_radius=radius, variant=variant))
    out = d.new_colour()
    d.FilterMoments(G.dev(col), out, G.dev(mom), G.gb_dev(f), G.dev(hist))
    got = G.host(out)
    keep = hist >= 4
    assert_same_bits_or_nan(got[keep], col[keep], "moments copy (history >= 4)")
    assert_close_with_nan(G, got[..., :3], want[..., :3], storage, f"moments colour r={radius} {variant}",
                          colour_abs=2e-5 if storage == "f32" else 1e-3)
    g, w = got[..., 3].astype(np.float64), want[..., 3].astype(np.float64)
    fin = np.isfinite(w)
    assert np.array_equal(np.isfinite(g), fin)
    lim = 8e-5 if storage == "f32" else 8e-5 + np.abs(w[fin]) * 2.0 ** -10
    assert np.all(np.abs(g[fin] - w[fin]) <= lim)


@pytest.mark.parametrize("variant", ["direct", "lds", "lds-general"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("step", [1, 2, 4, 8, 16, 32])
def test_atrous_nonfinite(G, oracle, storage, step, variant):
    """One wavelet iteration on a plane that holds NaN / +-inf in every channel (variance included), on surface pixels, on sky
    texels and next to the sky: +-inf clamp, a NaN reaches exactly the channels of the pixels whose 5x5 (dilated) window holds it —
    also through a zero-weight sky tap — and a sky centre is copied whatever its neighbours hold."""
    from svgf_amd import filter as F
    W, H = 333, 207
    rng = np.random.default_rng(30 + step)
    f = synth.make_frame(W, H, 0)
    dt = CDT[storage]
    src = np.concatenate([f["radiance"][..., :3] * 1.2 - 0.05, rng.uniform(-0.01, 0.05, (H, W, 1)).astype(np.float32)], -1).astype(dt)
    poison(rng, src, f["region"], n_per_kind=5)
    want = np.zeros_like(src); want_fb = np.full_like(src, 7)
    oracle.atrous(W, H, storage, src, want, want_fb, gbuf(f), step=step, phi_colour=10.0, phi_normal=128.0, iteration=0)
    wn = np.isnan(want.astype(np.float32))
    assert wn.any() and wn.sum() > 3 * 3 * 5 * 4, "the NaNs must have spread to their neighbours"
    # a NaN in ONE channel of a surface texel must not have made all four channels of its neighbours NaN
    assert (wn.sum(-1) == 1).any()
    d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant))
    out, fb = d.new_colour(), G.dev(np.full_like(src, 7))
    d.FilterKernel(G.dev(src), out, fb, G.gb_dev(f), step, 0)
    assert_close_with_nan(G, G.host(out), want, storage, f"a-trous step {step} {variant}")
    assert_close_with_nan(G, G.host(fb), want_fb, storage, f"a-trous feedback step {step} {variant}")


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_atrous_pair_nonfinite(G, oracle, storage):
    """Iterations 0 + 1 in one launch (svgf_atrous_pair) on a poisoned plane == the oracle's two iterations: the NaNs iteration 0
    produces travel to iteration 1 through the LDS ring, not through memory."""
    from svgf_amd import filter as F
    W, H = 333, 207
    rng = np.random.default_rng(41)
    f = synth.make_frame(W, H, 0)
    dt = CDT[storage]
    src = np.concatenate([f["radiance"][..., :3] * 1.2 - 0.05, rng.uniform(-0.01, 0.05, (H, W, 1)).astype(np.float32)], -1).astype(dt)
    poison(rng, src, f["region"], n_per_kind=3)
    mid = np.zeros_like(src); want = np.zeros_like(src); want_fb = np.full_like(src, 7)
    oracle.atrous(W, H, storage, src, mid, want_fb, gbuf(f), step=1, phi_colour=10.0, phi_normal=128.0, iteration=0)
    oracle.atrous(W, H, storage, mid, want, None, gbuf(f), step=2, phi_colour=10.0, phi_normal=128.0, iteration=1)
    d = F.Denoiser(W, H, F.Params(storage=storage))
    out, fb = d.new_colour(), G.dev(np.full_like(src, 7))
    d.FilterKernelPair(G.dev(src), out, fb, G.gb_dev(f))
    assert_close_with_nan(G, G.host(fb), want_fb, storage, "pair feedback")
    # iteration 1's input differs from the oracle's by iteration 0's rounding: the looser bound of a two-stage chain
    assert_close_with_nan(G, G.host(out), want, storage, "pair result", colour_abs=1e-4 if storage == "f32" else 4e-3)


@pytest.mark.parametrize("variant", ["auto", "direct"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("mv", [(0.0, 0.0), (-2.5, 1.5)])
def test_free_running_sequence_with_nonfinite_radiance(G, oracle, storage, mv, variant):
    """Eight free-running frames through svgf_denoise_frame — cold frames (LDS moments kernel), then the steady state (fused
    pass-through, young list, exact sky zeros written by the temporal launch) — with NaN / +-inf radiance texels in frames 0, 2, 4
    and 5, on surfaces, on sky texels and next to the sky.  The NaNs persist in the history exactly as in the reference: NaN masks
    of the result and of the fed-back colour must be IDENTICAL to the oracle's in every frame, finite values within the free-running
    bounds of test_pipeline_free_running."""
    from svgf_amd import filter as F
    W, H, N = 256, 144, 8
    fr = frames(W, H, N, mv=mv)
    rng = np.random.default_rng(51)
    for k in (0, 2, 4, 5):
        fr[k]["radiance"] = fr[k]["radiance"].copy()
        poison(rng, fr[k]["radiance"], fr[k]["region"], n_per_kind=2, channels=3)
    ref = oracle.Pipeline(W, H, storage, steps=5, nthreads=8)
    d = F.Denoiser(W, H, F.Params(storage=storage, steps=5, variant=variant))
    gbs = [G.gb_dev(f) for f in fr]
    tight = 2e-5 if storage == "f32" else 1e-3
    loose = 5e-4 if storage == "f32" else 2e-2
    frac = 1e-3 if storage == "f32" else 2e-3
    for k in range(N):
        kp = max(k - 1, 0)
        want = ref.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp])).astype(np.float64)
        got = G.host(d.Render(G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), gbs[k], gbs[kp] if k else None)).astype(np.float64)
        assert np.array_equal(G.host(d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())), ref.taps["hist"]), f"frame {k}: history"
        wn = np.isnan(want)
        assert wn.any()
        assert np.array_equal(np.isnan(got), wn), f"frame {k}: NaN masks of the result differ ({np.isnan(got).sum()} vs {wn.sum()}; first at {np.argwhere(np.isnan(got) != wn)[:4].tolist()})"
        mom_got, mom_want = G.host(d.state_plane(F.PLANE_MOMENTS, 1 - d.pingpong())), ref.taps["mom"]
        assert np.array_equal(np.isnan(mom_got.astype(np.float32)), np.isnan(mom_want.astype(np.float32))), f"frame {k}: NaN masks of the moments differ"
        fb_got, fb_want = G.host(d.state_plane(F.PLANE_COLOUR, 1 - d.pingpong())).astype(np.float64), ref.taps["feedback"].astype(np.float64)
        # the driver stores the temporal colour only where something reads it again: compare where the next frame can read it
        # (feedback-written surface texels, and sky / young texels the temporal launch stored)
        assert np.array_equal(np.isnan(fb_got), np.isnan(fb_want)), f"frame {k}: NaN masks of the fed-back colour differ"
        err = np.abs(np.where(wn, 0, got) - np.where(wn, 0, want))[..., :3]
        assert err.max() <= loose, f"frame {k}: max colour error {err.max():.3e}"
        assert (err > tight + 1e-5 * np.abs(np.where(wn, 0, want)[..., :3])).mean() <= frac, f"frame {k}"


def test_frame_driver_nonfinite_equals_stage_calls(G):
    """With NaNs in the sequence the frame driver's fusions (exact sky zeros + the non-finite list that undoes them, sparse temporal
    colour, young list) still give the bits of the plain stage sequence — variant direct, where both run the same tap code."""
    from svgf_amd import filter as F
    W, H, N = 203, 77, 7
    fr = frames(W, H, N, mv=(-2.5, 1.5))
    rng = np.random.default_rng(61)
    for k in (1, 4, 5):
        fr[k]["radiance"] = fr[k]["radiance"].copy()
        poison(rng, fr[k]["radiance"], fr[k]["region"], n_per_kind=2, channels=3)
    for storage in ("f32", "f16"):
        hip = G.HipPipeline(W, H, storage, variant="direct", steps=3)
        d = F.Denoiser(W, H, F.Params(storage=storage, variant="direct", steps=3))
        gbs = [G.gb_dev(f) for f in fr]
        for k in range(N):
            kp = max(k - 1, 0)
            a = hip.frame(fr[k]["radiance"], gbs[k], gbs[kp])
            b = G.host(d.Render(G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), gbs[k], gbs[kp] if k else None))
            assert_same_bits_or_nan(b, a, f"{storage} frame {k}")
            if k >= 1:
                assert np.isnan(a.astype(np.float32)).any()


def test_nonfinite_list_overflow(G, oracle):
    """More non-finite pixels than the temporal launch's list holds (65 536): the moments launch falls back to every pixel of the
    launch rows.  A 1024 x 128 frame whose radiance is NaN on three quarters of the pixels, steady state (frame 5 of a sequence)."""
    from svgf_amd import filter as F
    W, H, N = 1024, 128, 6
    fr = frames(W, H, N)
    rng = np.random.default_rng(71)
    bad = rng.uniform(size=(H, W)) < 0.75
    rad = fr[5]["radiance"].copy()
    rad[bad, 1] = np.nan
    fr[5]["radiance"] = rad
    ref = oracle.Pipeline(W, H, "f32", steps=2, nthreads=8)
    d = F.Denoiser(W, H, F.Params(storage="f32", steps=2))
    gbs = [G.gb_dev(f) for f in fr]
    for k in range(N):
        kp = max(k - 1, 0)
        want = ref.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp]))
        got = G.host(d.Render(G.dev(fr[k]["radiance"]), gbs[k], gbs[kp] if k else None))
    assert bad.sum() > 65536
    assert np.array_equal(np.isnan(got), np.isnan(want))
    sky = fr[5]["region"] == synth.SKY
    assert np.isnan(want[sky]).any()


@pytest.mark.parametrize("variant", ["auto", "direct"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_taa_nonfinite(G, oracle, storage, variant):
    """svgf_taa (the LDS-tiled kernel and the per-pixel one) on planes that hold NaN / +-inf, in the filtered frame and in the history
    (alpha included): the reference keeps a NaN through imageLoad, its glm min / max see it position by position (:330-338), the NaN
    test of :351 writes black.  Against the oracle within the stage's tolerance — and the same set of black pixels."""
    from svgf_amd import filter as F
    W, H = 333, 207
    rng = np.random.default_rng(81)
    dt = CDT[storage]
    f = synth.make_frame(W, H, 0)
    filt = np.concatenate([f["base"] * 1.1, np.ones((H, W, 1), np.float32)], -1).astype(dt)
    hist = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    poison(rng, filt, f["region"], n_per_kind=8)
    poison(rng, hist, f["region"], n_per_kind=6)
    want = np.zeros_like(filt)
    oracle.taa(W, H, storage, filt, hist, want)
    d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant))
    out = d.new_colour()
    d.TAA(G.dev(filt), G.dev(hist), out)
    got = G.host(out)
    assert not np.isnan(got.astype(np.float32)).any()
    black_w, black_g = (want[..., :3].astype(np.float32) == 0).all(-1), (got[..., :3].astype(np.float32) == 0).all(-1)
    assert black_w.sum() >= 20 and np.array_equal(black_w, black_g), "black (NaN-guarded) pixels differ"
    if storage == "f32":
        assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 2e-6
    else:
        from tests.helpers import half_ulp_diff
        assert half_ulp_diff(got, want).max() <= 1


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_nan_policy_zero_reads_a_nan_as_zero(G, oracle, storage):
    """svgf_params::nan_policy = SVGF_NAN_ZERO (an extension): the temporal stage reads a NaN channel of the radiance, of the previous colour
    and of the previous moments as 0.  (i) The stage call on poisoned planes == the oracle on the same planes with every NaN replaced by
    0, bit for bit.  (ii) A free-running sequence through svgf_denoise_frame with NaN radiance in most frames never shows a NaN and stays
    within the free-running bounds of the oracle fed the cleaned radiance.  (iii) With finite input the two policies give the same bits."""
    from svgf_amd import filter as F
    W, H = 331, 203
    dt = CDT[storage]
    rng = np.random.default_rng(95)
    f0, f1 = synth.make_frame(W, H, 3, mv=(1.0, 0.0)), synth.make_frame(W, H, 4, mv=(1.0, 0.0))
    prev = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
    mom_prev = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hist_prev = rng.integers(0, 40, (H, W)).astype(np.uint8)
    cur = (f1["radiance"] * 1.3 - 0.1).astype(dt)
    poison(rng, cur, f1["region"]); poison(rng, prev, f0["region"]); poison(rng, mom_prev, f0["region"], channels=2)
    clean = lambda a: np.where(np.isnan(a.astype(np.float32)), dt(0), a).astype(dt)     # noqa: E731
    out = np.zeros_like(cur); hist = np.zeros((H, W), np.uint8); mom = np.zeros((H, W, 2), dt)
    oracle.temporal(W, H, storage, clean(prev), clean(cur), out, gbuf(f1), gbuf(f0), hist_prev, hist, mom, clean(mom_prev),
                    depth_threshold=0.8, normal_threshold=0.9, history_base=24, mesh_id_test=1)
    d = F.Denoiser(W, H, F.Params(storage=storage, nan_policy="zero"))
    o_col, o_hist, o_mom = d.new_colour(), d.new_history(), d.new_moments()
    d.TemporalFilter(G.dev(prev), G.dev(cur), o_col, G.gb_dev(f1), G.gb_dev(f0), G.dev(hist_prev), o_hist, o_mom, G.dev(mom_prev))
    assert np.array_equal(G.host(o_hist), hist)
    assert np.array_equal(G.host(o_col).view(np.uint8), out.view(np.uint8)) and np.array_equal(G.host(o_mom).view(np.uint8), mom.view(np.uint8))
    # (ii)
    Wf, Hf, N = 256, 144, 8
    fr = frames(Wf, Hf, N, mv=(-2.5, 1.5))
    rads = []
    for k in range(N):
        r = fr[k]["radiance"].copy()
        if k != 3:
            poison(rng, r, fr[k]["region"], n_per_kind=2, channels=3)
        rads.append(r)
    ref = oracle.Pipeline(Wf, Hf, storage, steps=5, nthreads=8)
    dz = F.Denoiser(Wf, Hf, F.Params(storage=storage, steps=5, nan_policy="zero"))
    gbs = [G.gb_dev(f) for f in fr]
    tight, loose, frac = (2e-5, 5e-4, 1e-3) if storage == "f32" else (1e-3, 2e-2, 2e-3)
    for k in range(N):
        kp = max(k - 1, 0)
        want = ref.frame(np.where(np.isnan(rads[k]), np.float32(0), rads[k]), gbuf(fr[k]), gbuf(fr[kp])).astype(np.float64)
        got = G.host(dz.Render(G.dev(rads[k].astype(G.NPDT[storage])), gbs[k], gbs[kp] if k else None)).astype(np.float64)
        assert not np.isnan(got).any(), f"frame {k}: a NaN came through"
        assert np.array_equal(G.host(dz.state_plane(F.PLANE_HISTORY, 1 - dz.pingpong())), ref.taps["hist"]), f"frame {k}: history"
        err = np.abs(got - want)[..., :3]
        assert err.max() <= loose and (err > tight + 1e-5 * np.abs(want[..., :3])).mean() <= frac, f"frame {k}: {err.max():.3e}"
    # (iii)
    a, b = F.Denoiser(Wf, Hf, F.Params(storage=storage, steps=3)), F.Denoiser(Wf, Hf, F.Params(storage=storage, steps=3, nan_policy="zero"))
    for k in range(4):
        rad = G.dev(fr[k]["radiance"].astype(G.NPDT[storage]))
        ra, rb = a.Render(rad, gbs[k], gbs[k - 1] if k else None), b.Render(rad, gbs[k], gbs[k - 1] if k else None)
        assert np.array_equal(G.host(ra).view(np.uint8), G.host(rb).view(np.uint8)), k
