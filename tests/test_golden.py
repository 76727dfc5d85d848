"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep reproducing them
(CPU), and the HIP path must match them within the stage tolerance (GPU)."""
import os

import numpy as np
import pytest

from svgf_amd import synth
from tests.conftest import ROOT
from tests.helpers import gbuf

GOLD = os.path.join(ROOT, "tests", "golden")


def test_oracle_reproduces_atrous_golden(oracle):
    z = np.load(os.path.join(GOLD, "atrous_96x64.npz"))
    H, W = z["src"].shape[:2]
    gb = {k: z[k] for k in ("motion", "normal", "uv")}
    f = synth.make_frame(W, H, 0)
    for k in ("motion", "normal", "uv"):
        assert np.array_equal(f[k], z[k]), f"synthetic generator drifted ({k})"
    for st, dt in (("f32", np.float32), ("f16", np.float16)):
        for step in (1, 4):
            out = np.zeros((H, W, 4), dt)
            oracle.atrous(W, H, st, z["src"].astype(dt), out, None, gb, step=step, phi_colour=10.0, phi_normal=128.0, iteration=1)
            np.testing.assert_allclose(out.astype(np.float32), z[f"out_{st}_step{step}"].astype(np.float32), rtol=2e-6, atol=1e-7)


def _config1_inputs():
    W = H = 256
    f = synth.make_frame(W, H, 0)
    rng = np.random.default_rng(256)
    src = np.concatenate([f["radiance"][..., :3], rng.uniform(0.0, 0.05, (H, W, 1)).astype(np.float32)], -1)
    return W, H, f, src


def test_baseline_config1_256x256_single_atrous_on_the_cpu_oracle(oracle):
    """BASELINE.json configs[0]: 256x256 synthetic G-buffer + noisy radiance, a single a-trous iteration through the scalar
    C++ loop (plumbing, no GPU) — against the committed sample, the NumPy restatement and the iteration-0 feedback rule."""
    from oracle import svgf_numpy as snp
    z = np.load(os.path.join(GOLD, "config1_256x256.npz"))
    W, H, f, src = _config1_inputs()
    sums = np.array([int(np.frombuffer(f[k].tobytes(), np.uint8).astype(np.uint64).sum()) for k in ("motion", "normal", "uv", "radiance")], np.uint64)
    assert np.array_equal(sums, z["input_sums"]), "synthetic generator drifted"
    for st, dt in (("f32", np.float32), ("f16", np.float16)):
        s = src.astype(dt)
        out = np.zeros_like(s); fb = np.full_like(s, 7)
        oracle.atrous(W, H, st, s, out, fb, gbuf(f), step=1, phi_colour=10.0, phi_normal=128.0, iteration=0)
        tol = 2e-6 if st == "f32" else 1e-3
        assert np.abs(out[::8, ::8].astype(np.float32) - z[f"sample_{st}"].astype(np.float32)).max() <= tol
        np.testing.assert_allclose(out.astype(np.float64).mean((0, 1)), z[f"mean_{st}"], rtol=1e-5)
        sky = f["region"] == synth.SKY
        assert np.array_equal(fb[~sky], out[~sky]) and np.all(fb[sky] == 7)            # Filter.cuh:619-622
    want, fbmask = snp.atrous(src, gbuf(f), step=1, phi_colour=10.0, phi_normal=128.0)   # independent restatement, fp32
    assert np.array_equal(fbmask, f["region"] != synth.SKY)
    got = np.zeros_like(src)
    oracle.atrous(W, H, "f32", src, got, None, gbuf(f), step=1, phi_colour=10.0, phi_normal=128.0, iteration=1)
    assert np.abs(got - want).max() <= 5e-6


@pytest.mark.gpu
def test_hip_matches_config1_golden():
    """The same configuration through the HIP kernels (both variants) against the committed sample."""
    import torch
    from svgf_amd import filter as F
    z = np.load(os.path.join(GOLD, "config1_256x256.npz"))
    W, H, f, src = _config1_inputs()
    gb = F.GBuffer(*(torch.from_numpy(f[k]).cuda() for k in ("motion", "normal", "uv")))
    for st, dt in (("f32", np.float32), ("f16", np.float16)):
        for variant in ("lds", "direct"):
            d = F.Denoiser(W, H, F.Params(storage=st, variant=variant))
            out, fb = d.new_colour(), d.new_colour()
            d.FilterKernel(torch.from_numpy(src.astype(dt)).cuda(), out, fb, gb, 1, 0)
            got = out.cpu().numpy()
            tol = 2e-5 if st == "f32" else 2e-3
            assert np.abs(got[::8, ::8].astype(np.float32) - z[f"sample_{st}"].astype(np.float32)).max() <= tol, (st, variant)


def test_oracle_reproduces_pipeline_golden(oracle):
    z = np.load(os.path.join(GOLD, "pipeline_64x48.npz"))
    W, H, N, mv = int(z["W"]), int(z["H"]), int(z["N"]), tuple(float(v) for v in z["mv"])
    for st in ("f32", "f16"):
        p = oracle.Pipeline(W, H, st, steps=5)
        frs = [synth.make_frame(W, H, k, mv=mv) for k in range(N)]
        for k in range(N):
            o = p.frame(frs[k]["radiance"], gbuf(frs[k]), gbuf(frs[max(k - 1, 0)]))
            if k in (3, N - 1):
                assert np.array_equal(p.hist[p.P ^ 1], z[f"hist_{st}_frame{k}"])
                tol = 2e-5 if st == "f32" else 2e-3
                assert np.abs(o.astype(np.float32) - z[f"out_{st}_frame{k}"].astype(np.float32)).max() <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["direct", "lds"])
def test_hip_matches_atrous_golden(variant):
    from svgf_amd import filter as F
    from tests import gpu_helpers as G
    z = np.load(os.path.join(GOLD, "atrous_96x64.npz"))
    H, W = z["src"].shape[:2]
    gb = G.gb_dev({k: z[k] for k in ("motion", "normal", "uv")})
    for st, dt in (("f32", np.float32), ("f16", np.float16)):
        d = F.Denoiser(W, H, F.Params(storage=st, variant=variant))
        for step in (1, 4):
            out = d.new_colour()
            d.FilterKernel(G.dev(z["src"].astype(dt)), out, None, gb, step, 1)
            G.assert_colour_close(G.host(out), z[f"out_{st}_step{step}"], st, f"golden a-trous {st} step {step}")


@pytest.mark.gpu
def test_hip_matches_pipeline_golden():
    """The free-running HIP pipeline against the committed 8-frame golden: accept/reject masks exact (mismatch count printed,
    must be 0), colour in two tiers as in test_pipeline_free_running — a tight bound that all but 0.1 % (fp32) / 0.2 % (fp16) of the values must
    meet (a 1e-3 regression fails it) and the loose bound of the ill-conditioned zero-variance pixels (Tolerance)."""
    from tests import gpu_helpers as G
    z = np.load(os.path.join(GOLD, "pipeline_64x48.npz"))
    W, H, N, mv = int(z["W"]), int(z["H"]), int(z["N"]), tuple(float(v) for v in z["mv"])
    for st in ("f32", "f16"):
        hip = G.HipPipeline(W, H, st, steps=5)
        frs = [synth.make_frame(W, H, k, mv=mv) for k in range(N)]
        gbs = [G.gb_dev(f) for f in frs]
        tight, loose, frac = (2e-5, 5e-4, 1e-3) if st == "f32" else (1e-3, 2e-2, 2e-3)      # 2x the maxima measured on MI355X (profiles/r0N_parity_report.json)
        for k in range(N):
            o = hip.frame(frs[k]["radiance"], gbs[k], gbs[max(k - 1, 0)])
            if k in (3, N - 1):
                mism = int((hip.taps["hist"] != z[f"hist_{st}_frame{k}"]).sum())
                want = z[f"out_{st}_frame{k}"].astype(np.float64)
                err = np.abs(o.astype(np.float64) - want)[..., :3]
                beyond = float((err > tight + 1e-5 * np.abs(want[..., :3])).mean())
                print(f"golden pipeline {st} frame {k}: mask mismatches {mism}, max err {err.max():.3e}, mean err {err.mean():.3e}, beyond tight {beyond:.2e}")
                assert mism == 0
                assert err.max() <= loose
                assert beyond <= frac, f"{st} frame {k}: {beyond:.2e} of the values beyond the tight tolerance {tight}"
