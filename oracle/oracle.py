"""ctypes front end of the CPU oracle (oracle/svgf_oracle.cpp) + the frame sequencing of
src/App.cu:552-556 on top of it.

TEST INFRASTRUCTURE ONLY — see the header of svgf_oracle.cpp.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by svgf_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvgf_oracle.so")
# the envelope build (Makefile: fp64 islands in fp32, FMA contraction on): flavour="fp32fma".  Never the checker — see svgf_oracle.cpp, wide_t
_ENV_PATH = os.path.join(_HERE, "libsvgf_oracle_fp32fma.so")
_libs = {}

STORAGE = {"f32": 0, "f16": 1}
_CDT = {"f32": np.float32, "f16": np.float16}

DEFAULTS = dict(steps=3, depth_threshold=0.8, normal_threshold=0.9, history_base=24,
                phi_colour=10.0, phi_normal=128.0, moments_radius=3, mesh_id_test=1)  # src/App.h:109-114


_FLAVOURS = {"oracle": "libsvgf_oracle.so", "fp32fma": "libsvgf_oracle_fp32fma.so", "fp32": "libsvgf_oracle_fp32.so", "fma": "libsvgf_oracle_fma.so",
             "fused": "libsvgf_oracle_fused.so", "hwulp": "libsvgf_oracle_hwulp.so"}


def _stale(path):
    src = os.path.join(_HERE, "svgf_oracle.cpp")
    return not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "Makefile")))


def build(force: bool = False, flavour: str = "oracle") -> str:
    """Builds ONE library: the checker by default.  The envelope flavours (fp32fma, fma, fused: -mfma, x86 with FMA only) are built when something asks
    for them (lib(flavour)) — a host whose compiler cannot build them still gets the checker, and only the envelope tests skip.
    force=True builds the checker and, as far as this host can, every flavour (__graft_entry__.build())."""
    path = os.path.join(_HERE, _FLAVOURS[flavour])
    if force or _stale(path):
        subprocess.check_call(["make", "-C", _HERE, "-B", _FLAVOURS[flavour]], stdout=subprocess.DEVNULL)
    if force and flavour == "oracle":
        for other in _FLAVOURS:
            if other != "oracle":
                try:
                    subprocess.check_call(["make", "-C", _HERE, "-B", _FLAVOURS[other]], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                except subprocess.CalledProcessError:
                    pass                       # (no FMA on this host: lib(other) will raise EnvelopeUnavailable, the envelope tests skip)
    return path


class EnvelopeUnavailable(RuntimeError):
    """An envelope flavour of the oracle cannot be built on this host (its compiler or CPU lacks x86 FMA)."""


_flavour = "oracle"


def lib(flavour=None):
    """flavour None: the library the module-level stage functions currently run on (the oracle, unless inside `using("fp32fma")`)."""
    flavour = flavour or _flavour
    if flavour not in _libs:
        try:
            path = build(flavour=flavour)
        except subprocess.CalledProcessError as e:
            if flavour == "oracle":
                raise
            raise EnvelopeUnavailable(f"oracle flavour {flavour!r} cannot be built here: {e}") from e
        L = C.CDLL(path)
        L.svgf_oracle_f2h.restype = C.c_uint16
        L.svgf_oracle_f2h.argtypes = [C.c_float]
        L.svgf_oracle_h2f.restype = C.c_float
        L.svgf_oracle_h2f.argtypes = [C.c_uint16]
        _libs[flavour] = L
    return _libs[flavour]


def set_hw_ulp_seed(seed: int):
    """Envelope flavour "hwulp" only: which assignment of -1 / 0 / +1 ulp nudges to the operands of log2 / exp2 / rcp / rsq the build models
    (svgf_oracle.cpp: hw_ulp) — one "transcendental unit" per seed; the envelope is the largest distance over a handful of them."""
    L = lib("hwulp")
    L.svgf_oracle_set_hw_ulp_seed.argtypes = [C.c_uint32]
    L.svgf_oracle_set_hw_ulp_seed.restype = None
    L.svgf_oracle_set_hw_ulp_seed(int(seed))


class using:
    """with oracle.using("fp32fma"): ... — the stage functions and Pipeline.frame run on the envelope build inside the block."""

    def __init__(self, flavour):
        self.flavour = flavour

    def __enter__(self):
        global _flavour
        self.prev, _flavour = _flavour, self.flavour
        return self

    def __exit__(self, *exc):
        global _flavour
        _flavour = self.prev
        return False


def _p(a):
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"], "oracle planes must be C-contiguous"
    return C.c_void_p(a.ctypes.data)


def _geo(W, H, geo):
    """geo = (y0, rows, yb, ye) or None for the whole frame."""
    if geo is None:
        return (0, H, 0, H)
    return tuple(int(v) for v in geo)


def temporal(W, H, storage, prev_colour, cur_in, cur_out, gb_cur, gb_prev, hist_prev, hist_cur, mom_cur, mom_prev,
             *, depth_threshold, normal_threshold, history_base, mesh_id_test=1, geo=None, nthreads=1):
    y0, rows, yb, ye = _geo(W, H, geo)
    rc = lib().svgf_oracle_temporal(
        W, H, y0, rows, yb, ye, STORAGE[storage], _p(prev_colour), _p(cur_in), _p(cur_out),
        _p(gb_cur["motion"]), _p(gb_cur["normal"]), _p(gb_cur["uv"]),
        _p(gb_prev["motion"]), _p(gb_prev["normal"]), _p(gb_prev["uv"]),
        _p(hist_prev), _p(hist_cur), _p(mom_cur), _p(mom_prev),
        C.c_float(depth_threshold), C.c_float(normal_threshold), int(history_base), int(mesh_id_test), int(nthreads))
    assert rc == 0


def moments(W, H, storage, colour, out, mom, gb, hist, *, phi_colour, phi_normal, radius=3, geo=None, nthreads=1):
    y0, rows, yb, ye = _geo(W, H, geo)
    rc = lib().svgf_oracle_moments(
        W, H, y0, rows, yb, ye, STORAGE[storage], _p(colour), _p(out), _p(mom), _p(gb["motion"]), _p(gb["normal"]),
        _p(hist), C.c_float(phi_colour), C.c_float(phi_normal), int(radius), int(nthreads))
    assert rc == 0


def atrous(W, H, storage, src, dst, feedback, gb, *, step, phi_colour, phi_normal, iteration, geo=None, nthreads=1):
    y0, rows, yb, ye = _geo(W, H, geo)
    rc = lib().svgf_oracle_atrous(
        W, H, y0, rows, yb, ye, STORAGE[storage], _p(src), _p(dst), _p(feedback), _p(gb["motion"]), _p(gb["normal"]),
        int(step), C.c_float(phi_colour), C.c_float(phi_normal), int(iteration), int(nthreads))
    assert rc == 0


def taa(W, H, storage, filtered, history, out, *, geo=None, nthreads=1):
    """TAAFilterKernel (src/Filter.cuh:288-357) with a separate previous-output plane."""
    y0, rows, yb, ye = _geo(W, H, geo)
    rc = lib().svgf_oracle_taa(W, H, y0, rows, yb, ye, STORAGE[storage], _p(filtered), _p(history), _p(out), int(nthreads))
    assert rc == 0


def albedo(mode, W, rows, storage, inp, alb, out):
    """Albedo demodulation (mode 0) / re-modulation (mode 1): the build's own definition (the reference has none)."""
    rc = lib().svgf_oracle_albedo(int(mode), W, rows, STORAGE[storage], _p(inp), _p(alb), _p(out))
    assert rc == 0


def pack_gbuffer(position, normal, bary, view_proj, prev_view_proj, cam):
    """GBuffer.frag:62-88 + GBuffer.vert:21-34 from linear attribute planes (float32 (H, W, 4) each; matrices: 16 floats, column-major)
    -> (motion float32 (H, W, 4), normal uint16 (H, W, 4), uv uint16 (H, W, 4)).  The C++ twin of oracle/svgf_numpy.py:pack_gbuffer."""
    H, W = position.shape[:2]
    f = lambda a, n=None: np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1) if n else a, dtype=np.float32)      # noqa: E731
    pos, nrm, bar = f(position), f(normal), f(bary)
    vp, pvp, cm = f(view_proj, 16), f(prev_view_proj, 16), f(cam, 3)
    assert vp.size == 16 and pvp.size == 16 and cm.size == 3
    motion, nout, uvout = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 4), np.uint16)
    with np.errstate(all="ignore"):
        rc = lib().svgf_oracle_pack_gbuffer(W, H, _p(pos), _p(nrm), _p(bar), _p(vp), _p(pvp), _p(cm), _p(motion), _p(nout), _p(uvout))
    assert rc == 0
    return motion, nout, uvout


class Pipeline:
    """Whole-frame sequencing: TemporalFilter -> FilterMoments -> WaveletFilter (src/App.cu:552-556,
    469-514), with the host-side fixes SURVEY.md App. B lists (#1 history ping-pong, #4 current
    moments, #9 zero-initialised state, #12 no odd-N copy)."""

    def __init__(self, W, H, storage="f32", nthreads=1, **params):
        self.W, self.H, self.storage, self.nthreads = W, H, storage, nthreads
        self.p = dict(DEFAULTS)
        self.p.update(params)
        dt = _CDT[storage]
        self.colour = [np.zeros((H, W, 4), dt) for _ in range(2)]   # RenderBuffer[2]   App.h:138
        self.mom = [np.zeros((H, W, 2), dt) for _ in range(2)]      # MomentsBuffer[2]  App.h:139
        self.filt = [np.zeros((H, W, 4), dt) for _ in range(2)]     # FilterBuffer[2]   App.h:140
        self.hist = [np.zeros((H, W), np.uint8) for _ in range(2)]  # HistoryLengthBuffer (ping-ponged)
        self.P = 0                                                  # PingPongInx       App.cu:374
        self.taps = {}

    def frame(self, radiance, gb_cur, gb_prev=None):
        """One frame.  self.taps keeps copies of every stage's inputs and outputs so that a device stage can
        be compared from bit-identical inputs."""
        p, P, W, H, st, nt = self.p, self.P, self.W, self.H, self.storage, self.nthreads
        if gb_prev is None:
            gb_prev = gb_cur
        rad = np.ascontiguousarray(radiance.astype(_CDT[st]))
        self.taps = {"prev_colour": self.colour[1 - P].copy(), "prev_hist": self.hist[1 - P].copy(),
                     "prev_mom": self.mom[1 - P].copy(), "radiance": rad, "atrous_in": [], "atrous_out": []}
        temporal(W, H, st, self.colour[1 - P], rad, self.colour[P], gb_cur, gb_prev, self.hist[1 - P], self.hist[P],
                 self.mom[P], self.mom[1 - P], depth_threshold=p["depth_threshold"],
                 normal_threshold=p["normal_threshold"], history_base=p["history_base"],
                 mesh_id_test=p["mesh_id_test"], nthreads=nt)
        self.taps["temporal"] = self.colour[P].copy()
        self.taps["hist"] = self.hist[P].copy()
        self.taps["mom"] = self.mom[P].copy()
        moments(W, H, st, self.colour[P], self.filt[0], self.mom[P], gb_cur, self.hist[P], phi_colour=p["phi_colour"],
                phi_normal=p["phi_normal"], radius=p["moments_radius"], nthreads=nt)
        self.taps["moments"] = self.filt[0].copy()
        pp = 0
        for i in range(p["steps"]):
            self.taps["atrous_in"].append(self.filt[pp].copy())
            atrous(W, H, st, self.filt[pp], self.filt[1 - pp], self.colour[P] if i == 0 else None, gb_cur, step=1 << i,
                   phi_colour=p["phi_colour"], phi_normal=p["phi_normal"], iteration=i, nthreads=nt)
            self.taps["atrous_out"].append(self.filt[1 - pp].copy())
            pp ^= 1
        self.taps["feedback"] = self.colour[P].copy()
        self.P ^= 1
        return self.filt[pp]
