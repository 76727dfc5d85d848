"""GPU parity on NON-FINITE / OUT-OF-RANGE G-BUFFER texels (include/svgf.h "Non-finite G-buffer texels").

The reference defines what happens (tests/gbuffer_poison.py lists it): ivec2(MotionVector) is a saturating float -> int conversion with
NaN -> 0 (Filter.cuh:232), a NaN depth / NaN normal ACCEPTS the reprojection (:242,252: the comparisons are false), max(weightZ, 0.0)
drops a NaN depth term (:424), saturate(NaN) = 0 (:419), max(ddepth, 1e-6f) is fmaxf (:563).  The oracle restates it (CPU known answers:
tests/test_oracle_stages.py::test_nonfinite_gbuffer_matches_numpy_and_known_answers); here the HIP kernels against the oracle: temporal
bit-exact (accept / reject masks = the history plane), moments and a-trous with identical NaN masks and the stage tolerances, the frame
driver free-running and against the stage calls, the strip driver against the whole frame."""
import numpy as np
import pytest

from svgf_amd import synth
from tests.gbuffer_poison import poison_gbuffer
from tests.helpers import CDT, frames, free_running_bounds, free_running_envelope, gbuf
from tests.test_gpu_nonfinite import assert_close_with_nan, assert_same_bits_or_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("mesh", [0, 1])
def test_temporal_with_poisoned_gbuffers_is_bit_exact(G, oracle, storage, mesh):
    """Both G-buffers poisoned (motion NaN / +-inf / +-1e20 / +-3e9 / just inside the int range, depth NaN / negative / denormal / 1e30 / 0,
    ddepth NaN / negative, NaN and zero-length normals, NaN / inf instance IDs), static and panning: colour, moments and the history plane —
    the accept / reject mask — equal the oracle's bit for bit; also through the guide plane of a previous frame (svgf_set_prev_guide)."""
    from svgf_amd import filter as F
    W, H = 331, 203
    dt = CDT[storage]
    for mv in ((0.0, 0.0), (-2.5, 1.5)):
        rng = np.random.default_rng(7)
        f0, f1 = synth.make_frame(W, H, 3, mv=mv), synth.make_frame(W, H, 4, mv=mv)
        p0, _ = poison_gbuffer(rng, f0)
        p1, placed = poison_gbuffer(rng, f1)
        prev = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
        mom_prev = rng.uniform(0, 1, (H, W, 2)).astype(dt)
        hist_prev = rng.integers(0, 40, (H, W)).astype(np.uint8)
        cur = (f1["radiance"] * 1.3 - 0.1).astype(dt)
        out = np.zeros_like(cur); hist = np.zeros((H, W), np.uint8); mom = np.zeros((H, W, 2), dt)
        oracle.temporal(W, H, storage, prev, cur, out, gbuf(p1), gbuf(p0), hist_prev, hist, mom, mom_prev,
                        depth_threshold=0.8, normal_threshold=0.9, history_base=24, mesh_id_test=mesh)
        clean = np.zeros((H, W), np.uint8); o2 = np.zeros_like(cur); m2 = np.zeros((H, W, 2), dt)
        oracle.temporal(W, H, storage, prev, cur, o2, gbuf(f1), gbuf(f0), hist_prev, clean, m2, mom_prev,
                        depth_threshold=0.8, normal_threshold=0.9, history_base=24, mesh_id_test=mesh)
        assert (clean != hist).sum() >= 10, "the poisoned texels must change some accept / reject decisions"
        d = F.Denoiser(W, H, F.Params(storage=storage, mesh_id_test=mesh))
        o_col, o_hist, o_mom = d.new_colour(), d.new_history(), d.new_moments()
        d.TemporalFilter(G.dev(prev), G.dev(cur), o_col, G.gb_dev(p1), G.gb_dev(p0), G.dev(hist_prev), o_hist, o_mom, G.dev(mom_prev))
        assert np.array_equal(G.host(o_hist), hist), f"history / accept mask, mv={mv}: {np.argwhere(G.host(o_hist) != hist)[:5].tolist()}"
        assert_same_bits_or_nan(G.host(o_col), out, f"temporal colour mv={mv}")
        assert_same_bits_or_nan(G.host(o_mom), mom, f"temporal moments mv={mv}")


@pytest.mark.parametrize("variant", ["direct", "lds", "lds-general"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("radius", [3, 1])
def test_moments_with_a_poisoned_gbuffer(G, oracle, storage, radius, variant):
    """The spatial estimate reads depth, ddepth and normals of 49 (9) taps: a NaN depth drops the depth term of every weight it enters (tap or
    centre), a NaN normal zeroes the weight, a NaN ddepth gives the 1e-8 floor.  Finite colour planes: the result stays finite."""
    from svgf_amd import filter as F
    W, H = 203, 131
    rng = np.random.default_rng(23)
    f, placed = poison_gbuffer(rng, synth.make_frame(W, H, 0), what=("depth", "ddepth", "normal"), per_value=6)
    dt = CDT[storage]
    col = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    mom = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hist = rng.integers(1, 6, (H, W)).astype(np.uint8)
    want = np.zeros_like(col)
    oracle.moments(W, H, storage, col, want, mom, gbuf(f), hist, phi_colour=10.0, phi_normal=128.0, radius=radius)
    assert np.isfinite(want.astype(np.float32)).all()
    d = F.Denoiser(W, H, F.Params(storage=storage, moments_radius=radius, variant=variant))
    out = d.new_colour()
    d.FilterMoments(G.dev(col), out, G.dev(mom), G.gb_dev(f), G.dev(hist))
    got = G.host(out)
    assert_close_with_nan(G, got[..., :3], want[..., :3], storage, f"moments colour r={radius} {variant}", colour_abs=2e-5 if storage == "f32" else 1e-3)
    g, w = got[..., 3].astype(np.float64), want[..., 3].astype(np.float64)
    lim = 8e-5 if storage == "f32" else 8e-5 + np.abs(w) * 2.0 ** -10
    assert np.all(np.abs(g - w) <= lim), f"variance: {np.abs(g - w).max():.3e} at {np.argwhere(np.abs(g - w) > lim)[:4].tolist()}"


@pytest.mark.parametrize("variant", ["direct", "lds", "lds-general"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("step", [1, 2, 4, 16, 32])
def test_atrous_with_a_poisoned_gbuffer(G, oracle, storage, step, variant):
    """One wavelet iteration whose G-buffer holds NaN / negative / denormal / 1e30 / 0 depths, NaN / negative ddepth, NaN and zero-length
    normals — on surfaces, next to the sky and on sky texels.  A texel whose depth is exactly 1e30 is copied like a sky texel and gets no
    feedback store (:552-558); the streaming kernel's fast taps turn a NaN depth into a NaN result, see it, and redo exactly those pixels the
    reference's way."""
    from svgf_amd import filter as F
    W, H = 333, 207
    rng = np.random.default_rng(130 + step)
    f, placed = poison_gbuffer(rng, synth.make_frame(W, H, 0), what=("depth", "ddepth", "normal"), per_value=6)
    dt = CDT[storage]
    src = np.concatenate([f["radiance"][..., :3] * 1.2 - 0.05, rng.uniform(-0.01, 0.05, (H, W, 1)).astype(np.float32)], -1).astype(dt)
    want = np.zeros_like(src); want_fb = np.full_like(src, 7)
    oracle.atrous(W, H, storage, src, want, want_fb, gbuf(f), step=step, phi_colour=10.0, phi_normal=128.0, iteration=0)
    assert np.isfinite(want.astype(np.float32)).all()
    e30 = [(y, x) for kind, y, x, v in placed if kind == "depth" and v == 1e30]
    assert e30 and all((want_fb[y, x] == 7).all() for y, x in e30 if f["motion"][y, x, 2] == np.float32(1e30))
    d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant))
    out, fb = d.new_colour(), G.dev(np.full_like(src, 7))
    d.FilterKernel(G.dev(src), out, fb, G.gb_dev(f), step, 0)
    assert_close_with_nan(G, G.host(out), want, storage, f"a-trous step {step} {variant}")
    assert_close_with_nan(G, G.host(fb), want_fb, storage, f"a-trous feedback step {step} {variant}")


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_atrous_pair_with_a_poisoned_gbuffer(G, oracle, storage):
    from svgf_amd import filter as F
    W, H = 333, 207
    rng = np.random.default_rng(141)
    f, _ = poison_gbuffer(rng, synth.make_frame(W, H, 0), what=("depth", "ddepth", "normal"), per_value=4)
    dt = CDT[storage]
    src = np.concatenate([f["radiance"][..., :3] * 1.2 - 0.05, rng.uniform(-0.01, 0.05, (H, W, 1)).astype(np.float32)], -1).astype(dt)
    mid = np.zeros_like(src); want = np.zeros_like(src); want_fb = np.full_like(src, 7)
    oracle.atrous(W, H, storage, src, mid, want_fb, gbuf(f), step=1, phi_colour=10.0, phi_normal=128.0, iteration=0)
    oracle.atrous(W, H, storage, mid, want, None, gbuf(f), step=2, phi_colour=10.0, phi_normal=128.0, iteration=1)
    d = F.Denoiser(W, H, F.Params(storage=storage))
    out, fb = d.new_colour(), G.dev(np.full_like(src, 7))
    d.FilterKernelPair(G.dev(src), out, fb, G.gb_dev(f))
    assert_close_with_nan(G, G.host(fb), want_fb, storage, "pair feedback")
    assert_close_with_nan(G, G.host(out), want, storage, "pair result", colour_abs=1e-4 if storage == "f32" else 4e-3)


def _poisoned_sequence(W, H, N, mv, seed, which=(1, 2, 4, 5)):
    fr = frames(W, H, N, mv=mv)
    rng = np.random.default_rng(seed)
    for k in which:
        fr[k], _ = poison_gbuffer(rng, fr[k], per_value=2)
    return fr


@pytest.mark.parametrize("variant", ["auto", "direct"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("mv", [(0.0, 0.0), (-2.5, 1.5)])
def test_free_running_sequence_with_poisoned_gbuffers(G, oracle, storage, mv, variant):
    """Eight free-running frames through svgf_denoise_frame, the G-buffers of frames 1, 2, 4 and 5 poisoned (each is the CURRENT G-buffer of
    its frame and the PREVIOUS one of the next — through the guide plane when the shortcut is on): the history plane equals the oracle's in
    every frame, the colours stay finite and inside the free-running bounds of test_pipeline_free_running."""
    from svgf_amd import filter as F
    W, H, N = 256, 144, 8
    fr = _poisoned_sequence(W, H, N, mv, 151)
    ref = oracle.Pipeline(W, H, storage, steps=5, nthreads=8)
    d = F.Denoiser(W, H, F.Params(storage=storage, steps=5, variant=variant))
    d.set_prev_guide(True)
    gbs = [G.gb_dev(f) for f in fr]
    # bounds: those of the clean sequences (tests/helpers.py:FREE_RUNNING) with twice the maximum — a texel whose NaN depth drops the depth term
    # of its weights blends across edges, where the ill-conditioned luminance term (phi_l = 1e-4 at zero variance) is all that is left — and, for
    # the cases the clean sequences sit inside it, the envelope of THESE frames (two correct CPU builds of the reference's source)
    b = free_running_bounds(storage, mv)
    tight, loose, frac = b["tight"], 2 * b["loose"], b["frac"]
    worst = 0.0
    for k in range(N):
        kp = max(k - 1, 0)
        want = ref.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp])).astype(np.float64)
        got = G.host(d.Render(G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), gbs[k], gbs[kp] if k else None)).astype(np.float64)
        assert np.array_equal(G.host(d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())), ref.taps["hist"]), f"frame {k}: history"
        assert np.isfinite(want).all() and np.isfinite(got).all(), f"frame {k}"
        err = np.abs(got - want)[..., :3]
        worst = max(worst, float(err.max()))
        assert err.max() <= loose, f"frame {k}: max colour error {err.max():.3e}"
        assert (err > tight + 1e-5 * np.abs(want[..., :3])).mean() <= frac, f"frame {k}"
    env = free_running_envelope(oracle, fr, storage, flavour=b["inside_envelope"])      # (fp16 under a pan: the 1-ulp transcendental model, tests/helpers.py)
    assert env["mask_mismatches"] == 0
    assert worst <= env["max_abs"], f"HIP-vs-oracle {worst:.3e} is outside oracle-vs-oracle' ({b['inside_envelope']}) {env['max_abs']:.3e}"


@pytest.mark.parametrize("variant", ["direct", "auto", "lds-general"])
def test_frame_driver_with_poisoned_gbuffers_equals_stage_calls(G, variant):
    """The frame driver's fusions (guide plane, sparse temporal colour, young masks / list, exact sky zeros) on poisoned G-buffers == the plain
    stage sequence, bit for bit, both storages.  Under the default variants svgf_moments runs the streaming kernel for every pixel and the
    driver serves steady-state young pixels with the young-pixel launch: the two follow the same rule next to a NaN (include/svgf.h,
    "Bit-identity next to a NaN": a pixel whose fused-exponent sums hold a NaN is evaluated again the reference's way, no other)."""
    from svgf_amd import filter as F
    W, H, N = 203, 77, 7
    fr = _poisoned_sequence(W, H, N, (-2.5, 1.5), 161, which=(1, 3, 4))
    for storage in ("f32", "f16"):
        hip = G.HipPipeline(W, H, storage, variant=variant, steps=3)
        d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant, steps=3))
        gbs = [G.gb_dev(f) for f in fr]
        for k in range(N):
            kp = max(k - 1, 0)
            a = hip.frame(fr[k]["radiance"], gbs[k], gbs[kp])
            b = G.host(d.Render(G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), gbs[k], gbs[kp] if k else None))
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (storage, k)
        assert np.array_equal(G.host(d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())), hip.taps["hist"])


@pytest.mark.parametrize("plan", ["per-iteration", "ghost"])
def test_strip_driver_with_poisoned_gbuffers_equals_the_whole_frame(G, plan):
    """Three ranks with real peer addressing (the mailbox transport), a pan with motion reach 3, poisoned G-buffers: every frame equals the
    single-context FRAME DRIVER's BIT FOR BIT.  (A saturated motion vector lands outside the FRAME, so it is a rejection on every rank and
    never a halo violation; a NaN depth makes the streaming kernels redo pixels — only those whose fast result held a NaN, so the finite ones
    do not depend on how the strips cut tiles and bands.)"""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, N, storage = 320, 420, 3, 6, "f32"
    fr = _poisoned_sequence(W, H, N, (1.0, -2.5), 171, which=(1, 2, 4))
    whole = F.Denoiser(W, H, F.Params(storage=storage, steps=5))
    drv = strips.NativeStrips(W, H, world, F.Params(storage=storage, steps=5), list(range(world)), [0] * world, plan=plan, motion_reach=3, transport="mailbox")
    gbs = [G.gb_dev(f) for f in fr]
    prev_in = None
    for k in range(N):
        want = G.host(whole.Render(G.dev(fr[k]["radiance"]), gbs[k], gbs[k - 1] if k else None))
        torch.cuda.synchronize()
        cur_in = []
        for lay in drv.layouts:
            sl = slice(lay["y0"], lay["y1"])
            cur_in.append((G.dev(np.ascontiguousarray(fr[k]["radiance"][sl])), F.GBuffer(*(G.dev(np.ascontiguousarray(fr[k][n][sl])) for n in ("motion", "normal", "uv")))))
        torch.cuda.synchronize()
        outs = drv.frame([c[0] for c in cur_in], [c[1] for c in cur_in], [p[1] for p in prev_in] if prev_in else None)
        drv.sync()
        got = np.concatenate([G.host(drv.owned(r, o)) for r, o in enumerate(outs)], 0)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"plan {plan}: frame {k}"
        prev_in = cur_in
    drv.close()
