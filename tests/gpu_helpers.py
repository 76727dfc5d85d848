"""GPU-side helpers: numpy <-> device planes, HIP pipeline sequencing through the C ABI."""
from __future__ import annotations

import numpy as np

from svgf_amd import filter as F

NPDT = {"f32": np.float32, "f16": np.float16}


def dev(a, device="cuda:0"):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def gb_dev(frame, device="cuda:0"):
    return F.GBuffer(dev(frame["motion"], device), dev(frame["normal"], device), dev(frame["uv"], device))


def host(t):
    return t.detach().cpu().numpy()


# Stated parity tolerances (confirmed empirically on MI355X):
#   fp32 storage: the kernels use hardware exp2/log2/rcp in fp32 where the reference promotes to fp64.
#   fp16 storage: every stage re-quantises to 11-bit significands; a result within rounding distance of a
#   half boundary may flip by one half-ulp.
TOL = {
    "f32": dict(colour_abs=2e-5, colour_rel=1e-5, var_abs=2e-6, var_rel=2e-4),
    "f16": dict(max_ulp=1, frac=2e-3),
}


def assert_colour_close(got, want, storage, what=""):
    """got/want: (..., 4) arrays {r,g,b,variance} in storage dtype."""
    if storage == "f32":
        t = TOL["f32"]
        g, w = got.astype(np.float64), want.astype(np.float64)
        assert np.array_equal(np.isfinite(g), np.isfinite(w)), f"{what}: finite masks differ"
        fin = np.isfinite(w)
        g, w = np.where(fin, g, 0), np.where(fin, w, 0)
        dc = np.abs(g[..., :3] - w[..., :3])
        lim = t["colour_abs"] + t["colour_rel"] * np.abs(w[..., :3])
        assert np.all(dc <= lim), f"{what}: colour max err {dc.max():.3e} (limit {t['colour_abs']:.1e}+{t['colour_rel']:.0e}|v|), {np.sum(dc > lim)} px over"
        dv = np.abs(g[..., 3] - w[..., 3])
        limv = t["var_abs"] + t["var_rel"] * np.abs(w[..., 3])
        assert np.all(dv <= limv), f"{what}: variance max err {dv.max():.3e} rel {np.max(dv / (np.abs(w[..., 3]) + 1e-12)):.3e}, {np.sum(dv > limv)} px over"
    else:
        from tests.helpers import half_ulp_diff
        t = TOL["f16"]
        fin = np.isfinite(want.astype(np.float32))
        assert np.array_equal(np.isfinite(got.astype(np.float32)), fin), f"{what}: finite masks differ"
        d = half_ulp_diff(got[fin], want[fin])
        assert d.max() <= t["max_ulp"], f"{what}: {d.max()} half-ulps"
        # (a share of the values — or two of them: on a frame of a few dozen texels one flip is already beyond the share)
        assert (d > 0).sum() <= max(t["frac"] * d.size, 2), f"{what}: {(d > 0).mean():.2e} of values off by one half-ulp"


class HipPipeline:
    """Stage-by-stage sequencing over caller-owned planes through svgf_temporal/moments/atrous
    (mirrors oracle.Pipeline so intermediate taps can be compared)."""

    def __init__(self, W, H, storage="f32", device="cuda:0", **params):
        import torch
        p = dict(steps=3, depth_threshold=0.8, normal_threshold=0.9, history_base=24, phi_colour=10.0,
                 phi_normal=128.0, moments_radius=3, mesh_id_test=1, variant="auto")
        p.update(params)
        self.params = F.Params(storage=storage, **p)
        self.d = F.Denoiser(W, H, self.params, device=torch.device(device).index or 0)
        self.colour = [self.d.new_colour() for _ in range(2)]
        self.mom = [self.d.new_moments() for _ in range(2)]
        self.filt = [self.d.new_colour() for _ in range(2)]
        self.hist = [self.d.new_history() for _ in range(2)]
        self.P = 0
        self.taps = {}
        self.storage = storage
        self.device = device

    def frame(self, radiance_np, gb_cur: F.GBuffer, gb_prev: F.GBuffer | None = None):
        P, d = self.P, self.d
        if gb_prev is None:
            gb_prev = gb_cur
        rad = dev(radiance_np.astype(NPDT[self.storage]), self.device)
        d.TemporalFilter(self.colour[1 - P], rad, self.colour[P], gb_cur, gb_prev, self.hist[1 - P], self.hist[P],
                         self.mom[P], self.mom[1 - P])
        self.taps["temporal"] = host(self.colour[P])
        self.taps["hist"] = host(self.hist[P])
        self.taps["mom"] = host(self.mom[P])
        d.FilterMoments(self.colour[P], self.filt[0], self.mom[P], gb_cur, self.hist[P])
        self.taps["moments"] = host(self.filt[0])
        out = d.WaveletFilter(self.filt, self.colour[P], gb_cur)
        self.P ^= 1
        return host(out)
