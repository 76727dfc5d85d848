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


HW_ULP_SEEDS = 12


def free_running_envelope(oracle, fr, storage, steps=5, flavour="fp32fma", tight=None, **params):
    """How far apart two CORRECT implementations of Filter.cuh end up on the frames `fr`, each feeding itself: the oracle (fp64 islands, no FMA
    contraction) against an envelope build of itself (oracle/Makefile).  flavour "fp32fma" (default): all fp32 + FMA contraction, nvcc's defaults on
    the reference's source.  flavour "hwulp": on top of that the fused exponent with log2 / exp2 / rcp / rsq results moved by -1 / 0 / +1 ulp — a
    model of a GPU's transcendental unit — over HW_ULP_SEEDS assignments of the nudges; the envelope is the largest distance over them.
    -> dict(max_abs, frac_beyond_tight, mask_mismatches): the bound a free-running device sequence is held against."""
    try:
        oracle.lib(flavour)                         # (built on first use; -mfma: x86 with FMA only)
    except oracle.EnvelopeUnavailable as e:
        import pytest
        pytest.skip(str(e))
    H, W = fr[0]["radiance"].shape[:2]
    tight = tight if tight is not None else (2e-5 if storage == "f32" else 1e-3)
    a = oracle.Pipeline(W, H, storage, steps=steps, nthreads=8, **params)
    ref, hists = [], []
    for k in range(len(fr)):
        ref.append(a.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[max(k - 1, 0)])).astype(np.float64))
        hists.append(a.taps["hist"].copy())
    worst, worst_frac, mism, per_seed = 0.0, 0.0, 0, []
    for seed in (range(HW_ULP_SEEDS) if flavour == "hwulp" else (None,)):
        if seed is not None:
            oracle.set_hw_ulp_seed(seed)
        b = oracle.Pipeline(W, H, storage, steps=steps, nthreads=8, **params)
        mine = 0.0
        for k in range(len(fr)):
            with oracle.using(flavour):
                wb = b.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[max(k - 1, 0)])).astype(np.float64)
            mism += int((hists[k] != b.taps["hist"]).sum())
            with np.errstate(all="ignore"):
                err = np.abs(np.nan_to_num(ref[k]) - np.nan_to_num(wb))[..., :3]
            mine = max(mine, float(err.max()))
            worst_frac = max(worst_frac, float((err > tight + 1e-5 * np.abs(np.nan_to_num(ref[k])[..., :3])).mean()))
        per_seed.append(mine)
        worst = max(worst, mine)
    if flavour == "hwulp":
        oracle.set_hw_ulp_seed(0)
    return dict(max_abs=worst, frac_beyond_tight=worst_frac, mask_mismatches=mism, per_seed=per_seed)


# Free-running bounds of the device sequences (tests/test_gpu_parity.py): per (storage, camera).  `loose` is 2x the maximum measured on MI355X
# (profiles/r05_parity_report.json), `frac` the share of values allowed beyond the tight (stage-wise) tolerance.  Next to them every test also
# holds the device against the ENVELOPE of its own frames (free_running_envelope): fp32 storage sits inside the fp32fma envelope in both motions;
# fp16 storage sits inside it with a static camera and OUTSIDE it under the pan (2.1e-3 against 7.3e-4: the two oracle builds share libm's
# correctly rounded exp / pow, so their halfs rarely flip, while v_exp_f32 / v_log_f32 / v_rcp_f32 / v_rsq_f32 are 1-ulp approximations whose
# half-ulp flips — <= 1 half-ulp per stage on < 0.2 % of the values, the stage-wise claim — compound over five requantised iterations and eight
# frames).  Round 6 measures THAT too: the envelope build "hwulp" moves exactly those four results by -1 / 0 / +1 ulp, and over twelve
# assignments of the nudges the oracle ends up 3.7e-4 ... 2.4e-3 away from it on those frames (profiles/r06_parity_envelope.json) — the device's
# 2.1e-3 is one draw of that distribution.  inside_envelope names the flavour the case is held against.
FREE_RUNNING = {
    ("f32", False): dict(tight=2e-5, loose=5e-4, frac=1e-3, inside_envelope="fp32fma"),
    ("f32", True): dict(tight=2e-5, loose=5e-4, frac=1e-3, inside_envelope="fp32fma"),
    ("f16", False): dict(tight=1e-3, loose=2e-2, frac=2e-3, inside_envelope="fp32fma"),
    ("f16", True): dict(tight=1e-3, loose=5e-3, frac=2e-3, inside_envelope="hwulp"),
}


def free_running_bounds(storage, mv):
    return FREE_RUNNING[(storage, bool(mv[0] or mv[1]))]
