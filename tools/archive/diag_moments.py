import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from svgf_amd import synth, filter as F
from tests import gpu_helpers as G
from tests.helpers import CDT, gbuf
W, H = 203, 131
rng = np.random.default_rng(2)
f = synth.make_frame(W, H, 0)
for storage in ("f32",):
    dt = CDT[storage]
    col = rng.uniform(0, 1, (H, W, 4)).astype(dt); mom = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hist = rng.integers(1, 8, (H, W)).astype(np.uint8)
    want = np.zeros_like(col)
    orc.moments(W, H, storage, col, want, mom, gbuf(f), hist, phi_colour=10.0, phi_normal=128.0, radius=3)
    for variant in ("direct", "lds"):
        d = F.Denoiser(W, H, F.Params(storage=storage, variant=variant))
        out = d.new_colour()
        d.FilterMoments(G.dev(col), out, G.dev(mom), G.gb_dev(f), G.dev(hist))
        got = G.host(out)
        e = np.abs(got.astype(np.float64) - want.astype(np.float64))
        print(variant, "max err colour", e[..., :3].max(), "var", e[..., 3].max(), "nan", np.isnan(got).sum())
        bad = np.argwhere(e.max(-1) > 1e-4)
        print("  bad px:", len(bad), bad[:8].tolist())
        for y, x in bad[:4]:
            print("   ", y, x, "hist", hist[y, x], "region", f["region"][y, x], "got", got[y, x], "want", want[y, x])
