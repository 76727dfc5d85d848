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
    from tests import gpu_helpers as G
    z = np.load(os.path.join(GOLD, "pipeline_64x48.npz"))
    W, H, N, mv = int(z["W"]), int(z["H"]), int(z["N"]), tuple(float(v) for v in z["mv"])
    for st in ("f32", "f16"):
        hip = G.HipPipeline(W, H, st, steps=5)
        frs = [synth.make_frame(W, H, k, mv=mv) for k in range(N)]
        gbs = [G.gb_dev(f) for f in frs]
        for k in range(N):
            o = hip.frame(frs[k]["radiance"], gbs[k], gbs[max(k - 1, 0)])
            if k in (3, N - 1):
                assert np.array_equal(hip.taps["hist"], z[f"hist_{st}_frame{k}"])
                loose = 2e-3 if st == "f32" else 3e-2
                assert np.abs(o.astype(np.float32) - z[f"out_{st}_frame{k}"].astype(np.float32))[..., :3].max() <= loose
