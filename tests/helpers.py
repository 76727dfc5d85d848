"""Shared helpers for the test-suite (CPU side)."""
from __future__ import annotations

import numpy as np

from oracle import svgf_numpy as snp
from svgf_amd import synth

CDT = {"f32": np.float32, "f16": np.float16}


def gbuf(frame):
    return {k: frame[k] for k in ("motion", "normal", "uv")}


def frames(W, H, n, mv=(0.0, 0.0), noise="1spp"):
    return [synth.make_frame(W, H, f, mv=mv, noise=noise) for f in range(n)]


class NumpyPipeline:
    """Frame sequencing (App.cu:552-556) over the independent NumPy restatement."""

    def __init__(self, W, H, storage="f32", **params):
        from oracle.oracle import DEFAULTS
        self.p = dict(DEFAULTS)
        self.p.update(params)
        dt = CDT[storage]
        self.dt = dt
        self.colour_prev = np.zeros((H, W, 4), dt)
        self.mom_prev = np.zeros((H, W, 2), dt)
        self.hist_prev = np.zeros((H, W), np.uint8)
        self.taps = {}

    def frame(self, radiance, gb_cur, gb_prev=None):
        p = self.p
        if gb_prev is None:
            gb_prev = gb_cur
        cur = radiance.astype(self.dt)
        col, hist, mom = snp.temporal(self.colour_prev, cur, gb_cur, gb_prev, self.hist_prev, self.mom_prev,
                                      depth_threshold=p["depth_threshold"], normal_threshold=p["normal_threshold"],
                                      history_base=p["history_base"], mesh_id_test=p["mesh_id_test"])
        self.taps["temporal"] = col
        self.taps["hist"] = hist
        f = snp.moments(col, mom, gb_cur, hist, phi_colour=p["phi_colour"], phi_normal=p["phi_normal"],
                        radius=p["moments_radius"])
        self.taps["moments"] = f
        feedback = col.copy()
        for i in range(p["steps"]):
            f, fb = snp.atrous(f, gb_cur, step=1 << i, phi_colour=p["phi_colour"], phi_normal=p["phi_normal"])
            if i == 0:
                feedback[fb] = f[fb]
        self.colour_prev, self.mom_prev, self.hist_prev = feedback, mom, hist
        return f


def half_ulp_diff(a, b):
    """|a-b| in units of half ULPs, for float16 arrays of finite values."""
    ai = a.view(np.int16).astype(np.int32)
    bi = b.view(np.int16).astype(np.int32)
    ai = np.where(ai < 0, -32768 - ai, ai)
    bi = np.where(bi < 0, -32768 - bi, bi)
    return np.abs(ai - bi)
