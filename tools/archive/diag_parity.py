"""Diagnostic (GPU box): error statistics of the HIP path vs the oracle, per stage and per frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from svgf_amd import synth, filter as F
from tests import gpu_helpers as G
from tests.helpers import CDT, frames, gbuf, half_ulp_diff

def stats(name, got, want):
    g, w = got.astype(np.float64), want.astype(np.float64)
    dc = np.abs(g[..., :3] - w[..., :3]); dv = np.abs(g[..., 3] - w[..., 3])
    over = dc > (2e-5 + 1e-5 * np.abs(w[..., :3]))
    print(f"{name}: colour max {dc.max():.3e} p99.9 {np.quantile(dc, 0.999):.3e} over {over.sum()} | var max {dv.max():.3e} "
          f"relmax {np.max(dv / (np.abs(w[..., 3]) + 1e-9)):.3e}")
    return over.any(-1)

variant = sys.argv[1] if len(sys.argv) > 1 else "direct"
for storage in ("f32", "f16"):
    for step in (1, 16):
        W, H = 333, 207
        rng = np.random.default_rng(3 + step)
        f = synth.make_frame(W, H, 0)
        dt = CDT[storage]
        src = np.concatenate([f["radiance"][..., :3] * 1.2 - 0.05, rng.uniform(-0.01, 0.05, (H, W, 1)).astype(np.float32)], -1).astype(dt)
        want = np.zeros_like(src)
        orc.atrous(W, H, storage, src, want, None, gbuf(f), step=step, phi_colour=10.0, phi_normal=128.0, iteration=1)
        d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant))
        out = d.new_colour()
        d.FilterKernel(G.dev(src), out, None, G.gb_dev(f), step, 1)
        got = G.host(out)
        stats(f"atrous {storage} step {step}", got, want)
        if storage == "f16":
            dd = half_ulp_diff(got, want)
            print("   half-ulp hist:", np.bincount(dd.ravel())[:6], "frac>0", (dd > 0).mean())
            ys, xs, cs = np.nonzero(dd > 1)
            for y, x, c in list(zip(ys, xs, cs))[:5]:
                print("    ", y, x, c, got[y, x], want[y, x], "src", src[y, x], "region", f["region"][y, x])

for storage in ("f32",):
    for mv in ((0.0, 0.0), (-2.5, 1.5)):
        W, H, N = 256, 144, 8
        fr = frames(W, H, N, mv=mv)
        ref = orc.Pipeline(W, H, storage, steps=5, nthreads=8)
        hip = G.HipPipeline(W, H, storage, steps=5, variant=variant)
        gbs = [G.gb_dev(f) for f in fr]
        for k in range(N):
            kp = max(k - 1, 0)
            want = ref.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp]))
            got = hip.frame(fr[k]["radiance"], gbs[k], gbs[kp])
            stats(f"mv{mv} frame {k} temporal", hip.taps["temporal"], ref.taps["temporal"])
            stats(f"mv{mv} frame {k} moments ", hip.taps["moments"], ref.taps["moments"])
            bad = stats(f"mv{mv} frame {k} output  ", got, want)
            ys, xs = np.nonzero(bad)
            for y, x in list(zip(ys, xs))[:4]:
                print("     px", y, x, "region", fr[k]["region"][y, x], "hist", hip.taps["hist"][y, x], "got", got[y, x], "want", want[y, x],
                      "mom-stage var got/want", hip.taps["moments"][y, x, 3], ref.taps["moments"][y, x, 3])
