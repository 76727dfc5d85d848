import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from tests import gpu_helpers as G
from tests.helpers import frames, gbuf
for storage in ("f32", "f16"):
    for mv in ((0.0, 0.0), (-2.5, 1.5)):
        W, H, N = 256, 144, 8
        fr = frames(W, H, N, mv=mv)
        ref = orc.Pipeline(W, H, storage, steps=5, nthreads=8)
        hip = G.HipPipeline(W, H, storage, steps=5)
        gbs = [G.gb_dev(f) for f in fr]
        for k in range(N):
            kp = max(k - 1, 0)
            want = ref.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp])).astype(np.float64)
            got = hip.frame(fr[k]["radiance"], gbs[k], gbs[kp]).astype(np.float64)
            e = np.abs(got - want)[..., :3]
            print(storage, mv, k, f"max {e.max():.2e} p99.9 {np.quantile(e,0.999):.2e} frac>2e-5 {(e>2e-5).mean():.2e} frac>1e-3 {(e>1e-3).mean():.2e} hist_eq {np.array_equal(hip.taps['hist'], ref.taps['hist'])}")
