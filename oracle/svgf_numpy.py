"""Independent NumPy restatement of the SVGF hot path, written from SURVEY.md Appendix A
(not from oracle/svgf_oracle.cpp) so that agreement of the two pins the oracle.

TEST INFRASTRUCTURE ONLY (same rules as svgf_oracle.cpp).  Whole frames only, vectorised
over pixels, one tap at a time.  fp32 arithmetic, with the fp64 islands of App. A.5.

Reference lines: src/Filter.cuh:55-83 (load/store), :199-207 (depth), :225-258 (reprojection),
:260-263 (luminance), :359-404 (temporal), :407-427 (weight), :430-525 (moments), :527-624 (a-trous).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
SKY_Z = f32(1e30)


def _ld(a):                      # storage -> float32
    return a.astype(np.float32)


def _clamp01(a):
    # glm::clamp = min(max(x, 0), 1) from `(x < y) ? y : x` (Filter.cuh:63-69,78-83): a NaN stays — and so does -0.0, which np.maximum(-0.0, 0.0)
    # turns into +0.0 (tests/fuzz_oracle.py seed 800425: one texel in 107 000 trials where the sign of a zero survived to the output)
    with np.errstate(invalid="ignore"):
        lo = np.where(a < f32(0), f32(0), a)
        return np.where(f32(1) < lo, f32(1), lo).astype(np.float32)


def _lum(c):                     # A.2 / Filter.cuh:262
    return f32(0.2126) * c[..., 0] + f32(0.7152) * c[..., 1] + f32(0.0722) * c[..., 2]


def _depth(motion):              # A.0 GetDepth
    z = motion[..., 2].copy()
    dz = motion[..., 3].copy()
    sky = z == 0
    z[sky] = SKY_Z
    dz[sky] = 0
    return z, dz


def _normal(normal_bits):
    return normal_bits[..., :3].view(np.float16).astype(np.float32)


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cvt_rzi_s32(a):
    """PTX cvt.rzi.s32.f32 — what int(x) / ivec2(vec2) compile to on the device (Filter.cuh:232,245-246): toward zero, saturating, NaN -> 0;
    returned as int64 so that the caller's two's-complement add can be written out (_add_wrap)."""
    with np.errstate(all="ignore"):
        t = np.trunc(a.astype(np.float64))
    t = np.where(np.isnan(t), 0.0, np.clip(t, -2147483648.0, 2147483647.0))
    return t.astype(np.int64)


def _add_wrap(a, b):
    """int32 + int32 with wrap-around (int64 in, int64 out)."""
    return ((a + b + 2**31) % 2**32) - 2**31


def _weight(zc, zp, phi_d, nc, npix, phi_n, lc, lp, phi_l):      # A.5
    with np.errstate(all="ignore"):
        d = _clamp01(_dot(nc, npix))
        d = np.where(np.isnan(d), f32(0), d)
        wn = np.power(d, f32(phi_n)).astype(np.float32)
        wz = np.where(phi_d == 0, f32(0), np.abs(zc - zp) / phi_d).astype(np.float32)
        wl = (np.abs(lc - lp) / phi_l).astype(np.float32)
        e = np.exp(0.0 - np.fmax(wl.astype(np.float64), 0.0) - np.fmax(wz.astype(np.float64), 0.0)) * wn.astype(np.float64)
    return e.astype(np.float32)


def temporal(prev_colour, cur, gb_cur, gb_prev, hist_prev, mom_prev, *, depth_threshold, normal_threshold,
             history_base, mesh_id_test=1):
    """A.2.  Returns (colour_out, hist_cur, mom_cur) in the storage dtype of `cur`."""
    sdt = cur.dtype
    H, W = cur.shape[:2]
    history_base = min(max(int(history_base), 1), 255)
    c = _clamp01(_ld(cur))[..., :3]
    Y, X = np.mgrid[0:H, 0:W]
    mv = gb_cur["motion"][..., :2]
    qx = _add_wrap(X.astype(np.int64), _cvt_rzi_s32(mv[..., 0]))
    qy = _add_wrap(Y.astype(np.int64), _cvt_rzi_s32(mv[..., 1]))
    inb = (qx >= 0) & (qx < W) & (qy >= 0) & (qy < H)
    qxc, qyc = np.clip(qx, 0, W - 1), np.clip(qy, 0, H - 1)
    zc, _ = _depth(gb_cur["motion"])
    zp_all, _ = _depth(gb_prev["motion"])
    zp = zp_all[qyc, qxc]
    with np.errstate(all="ignore"):
        ok = inb & ~(np.abs(zp - zc) > f32(depth_threshold))
    if mesh_id_test:
        idc = _cvt_rzi_s32(gb_cur["uv"][..., 3].view(np.float16).astype(np.float32))
        idp = _cvt_rzi_s32(gb_prev["uv"][..., 3].view(np.float16).astype(np.float32))[qyc, qxc]
        ok &= idc == idp
    nc = _normal(gb_cur["normal"])
    npv = _normal(gb_prev["normal"])[qyc, qxc]
    with np.errstate(all="ignore"):
        ok &= ~(_dot(nc, npv) < f32(normal_threshold))

    cp = np.where(ok[..., None], _clamp01(_ld(prev_colour))[qyc, qxc][..., :3], f32(0))
    mp = np.where(ok[..., None], _ld(mom_prev)[qyc, qxc], f32(0))
    hp = hist_prev[qyc, qxc].astype(np.int32)
    h = np.where(ok, np.minimum(history_base, hp + 1), 1).astype(np.int32)
    alpha = np.where(ok, (1.0 / h.astype(np.float64)).astype(np.float32), f32(1))
    L = _lum(c)
    m = np.stack([L, L * L], -1)
    a = alpha[..., None]
    m2 = mp * (f32(1) - a) + m * a
    # :396 max(0.f, x): CUDA's max(float, float) is fmaxf (and glm::max(0, x) = (0 < x) ? x : 0): a NaN x gives 0 — np.fmax, not np.maximum
    var = np.fmax(f32(0), m2[..., 1] - m2[..., 0] * m2[..., 0])
    c2 = cp * (f32(1) - a) + c * a
    out = _clamp01(np.concatenate([c2, var[..., None]], -1)).astype(sdt)
    return out, h.astype(np.uint8), m2.astype(sdt)


def moments(colour, mom, gb, hist, *, phi_colour, phi_normal, radius=3):
    """A.3."""
    sdt = colour.dtype
    H, W = colour.shape[:2]
    craw = _ld(colour)
    mraw = _ld(mom)
    h = hist.astype(np.float32)
    zc, dzc = _depth(gb["motion"])
    nc = _normal(gb["normal"])
    lc = _lum(craw)
    phi_d = (np.fmax(dzc.astype(np.float64), 1e-8) * 3.0).astype(np.float32)      # :461 max(float, double) = fmax: a NaN ddepth gives 1e-8
    Y, X = np.mgrid[0:H, 0:W]
    sw = np.zeros((H, W), np.float32)
    sc = np.zeros((H, W, 3), np.float32)
    sm = np.zeros((H, W, 2), np.float32)
    for yy in range(-radius, radius + 1):
        for xx in range(-radius, radius + 1):
            px, py = X + xx, Y + yy
            inside = (px < W) & (py < H) & (px >= 0) & (py >= 0)
            pxc, pyc = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
            cpix = craw[pyc, pxc]
            ln = np.sqrt(f32(xx * xx + yy * yy))
            w = _weight(zc, zc[pyc, pxc], phi_d * ln, nc, nc[pyc, pxc], phi_normal, lc, _lum(cpix), f32(phi_colour))
            w = np.where(inside, w, f32(0))
            # skipped taps must not touch the sums at all (0*inf would poison them)
            sw = np.where(inside, sw + w, sw)
            sc = np.where(inside[..., None], sc + cpix[..., :3] * w[..., None], sc)
            sm = np.where(inside[..., None], sm + mraw[pyc, pxc] * w[..., None], sm)
    sw = np.fmax(sw, f32(1e-6))
    C = sc / sw[..., None]
    M = sm / sw[..., None]
    var = M[..., 1] - M[..., 0] * M[..., 0]
    with np.errstate(all="ignore"):
        var = (var.astype(np.float64) * (4.0 / h.astype(np.float64))).astype(np.float32)
    filt = np.concatenate([C, var[..., None]], -1)
    out = np.where((h < 4)[..., None], filt, craw)
    return out.astype(sdt)


_K = np.array([1.0, 2.0 / 3.0, 1.0 / 6.0]).astype(np.float32)


def atrous(src, gb, *, step, phi_colour, phi_normal):
    """A.4.  Returns (out, feedback_mask): feedback_mask marks the pixels whose value the
    iteration-0 launch also stores into the feedback plane (all but sky)."""
    sdt = src.dtype
    H, W = src.shape[:2]
    c = _clamp01(_ld(src))
    lc = _lum(c)
    var = c[..., 3]
    zc, dzc = _depth(gb["motion"])
    nc = _normal(gb["normal"])
    sky = zc == SKY_Z
    # :562 max(0.0, eps + variance): the double literal selects CUDA's fmax, which drops a NaN variance (phi_l = 0: only equal luminances pass)
    phi_l = (np.float64(f32(phi_colour)) * np.sqrt(np.fmax(0.0, (f32(1e-10) + var).astype(np.float64)))).astype(np.float32)
    phi_d = np.fmax(dzc, f32(1e-6)) * f32(step)                                      # :563 max(float, float) = fmaxf
    Y, X = np.mgrid[0:H, 0:W]
    S = np.ones((H, W), np.float32)
    acc = c.copy()
    for yy in range(-2, 3):
        for xx in range(-2, 3):
            if xx == 0 and yy == 0:
                continue
            px, py = X + xx * step, Y + yy * step
            inside = (px < W) & (py < H) & (px >= 0) & (py >= 0)
            pxc, pyc = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
            k = _K[abs(xx)] * _K[abs(yy)]
            q = c[pyc, pxc]
            ln = np.sqrt(f32(xx * xx + yy * yy))
            w = _weight(zc, zc[pyc, pxc], phi_d * ln, nc, nc[pyc, pxc], phi_normal, lc, _lum(q), phi_l)
            g = w * k
            S = np.where(inside, S + g, S)
            acc[..., :3] = np.where(inside[..., None], acc[..., :3] + g[..., None] * q[..., :3], acc[..., :3])
            acc[..., 3] = np.where(inside, acc[..., 3] + (g * g) * q[..., 3], acc[..., 3])
    with np.errstate(all="ignore"):
        out = np.concatenate([acc[..., :3] / S[..., None], (acc[..., 3] / (S * S))[..., None]], -1)
    out = np.where(sky[..., None], c, out)
    return out.astype(sdt), ~sky


# ------------------------------------------------------------------ TAA + sRGB (next stage) -------
def _tex(k, n):
    """textureSample's coordinate: floor(uv * (n-1)) clamped, uv = k * (1/n) (+- 1/n for neighbours), all fp32."""
    return np.clip(np.floor(k * f32(n - 1)).astype(np.int64), 0, n - 1)


def _enc(rgb):
    # pow(rgb, 2.0) (:268): the square.  (np.power on a strided float32 view takes a SIMD path that is off by an ulp in a fifth of the values, on a
    # contiguous one it squares; the C++ oracle's powf(x, 2) is x * x in all but ~0.07 % of the values — tests/fuzz_oracle.py.  The difference matters
    # where the decoded channel cancels to ~1e-7 and the square root and the sRGB slope of 12.92 amplify its last bit to 1e-4.)
    c = np.square(rgb).astype(np.float32)
    r, g, b = c[..., 0], c[..., 1], c[..., 2]
    return np.stack([(r * f32(0.299) + g * f32(0.587)) + b * f32(0.114),
                     (r * f32(-0.14713) + g * f32(-0.28886)) + b * f32(0.436),
                     (r * f32(0.615) + g * f32(-0.51499)) + b * f32(-0.10001)], -1)


def _srgb(c):
    with np.errstate(all="ignore"):
        hi = f32(1 + np.float32(0.055)) * np.power(c, f32(1) / f32(2.4)).astype(np.float32) - f32(0.055)
    return np.where(c <= f32(0.0031308), f32(12.92) * c, hi).astype(np.float32)


def taa(filtered, history):
    """Filter.cuh:288-357, history read from a separate plane.  Returns the new output plane."""
    sdt = filtered.dtype
    H, W = filtered.shape[:2]
    inp = _clamp01(_ld(filtered))
    hist = _clamp01(_ld(history))
    iw, ih = f32(1.0) / f32(W), f32(1.0) / f32(H)
    u = (np.arange(W, dtype=np.float32) * iw)[None, :].repeat(H, 0)
    v = (np.arange(H, dtype=np.float32) * ih)[:, None].repeat(W, 1)

    def samp(img, uu, vv):
        return img[_tex(vv, H), _tex(uu, W)]
    last = samp(hist, u, v)
    mix = np.fmin(last[..., 3], f32(0.5))[..., None]            # :302 min(float, double literal): CUDA's overload is fmin (a NaN alpha gives 0.5)
    in0 = samp(inp, u, v)[..., :3]
    with np.errstate(all="ignore"):
        aa = np.sqrt((last[..., :3] * last[..., :3]) * (f32(1) - mix) + (in0 * in0) * mix).astype(np.float32)
    offs = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1)]
    ys = [_enc(samp(inp, u + f32(a) * iw if a else u, v + f32(b) * ih if b else v)[..., :3]) for a, b in offs]
    ya = _enc(aa)
    # :330-338 glm::min / glm::max on vec3: (y < x) ? y : x and (x < y) ? y : x per component — what a NaN does depends on its position
    gmin = lambda a, b: np.where(b < a, b, a)      # noqa: E731
    gmax = lambda a, b: np.where(a < b, b, a)      # noqa: E731
    with np.errstate(all="ignore"):
        mn = gmin(gmin(gmin(ys[0], ys[1]), gmin(ys[2], ys[3])), ys[4])
        mx = gmax(gmax(gmax(ys[0], ys[1]), gmax(ys[2], ys[3])), ys[4])
        mn2 = gmin(gmin(gmin(ys[5], ys[6]), gmin(ys[7], ys[8])), mn)
        mx2 = gmax(gmax(gmax(ys[5], ys[6]), gmax(ys[7], ys[8])), mx)
        mn = mn * f32(0.5) + mn2 * f32(0.5)
        mx = mx * f32(0.5) + mx2 * f32(0.5)
        ya = gmin(gmax(ya, mn), mx)
    with np.errstate(all="ignore"):
        r = (ya[..., 0] * f32(1) + ya[..., 1] * f32(0)) + ya[..., 2] * f32(1.13983)
        g = (ya[..., 0] * f32(1) + ya[..., 1] * f32(-0.39465)) + ya[..., 2] * f32(-0.58060)
        b = (ya[..., 0] * f32(1) + ya[..., 1] * f32(2.03211)) + ya[..., 2] * f32(0)
        rgb = np.power(np.stack([r, g, b], -1), f32(0.5)).astype(np.float32)
    bad = np.isnan(rgb).any(-1)
    rgb = np.where(bad[..., None], f32(0), rgb)
    out = np.concatenate([_srgb(rgb), np.where(bad, f32(0), f32(1))[..., None]], -1)
    out[bad] = 0          # fragColor = vec4(0) then ToSRGB(0) = 0, alpha forced back to 1 by :353
    out[..., 3] = 1
    return _clamp01(out).astype(sdt)


# ------------------------------------------------------------------ G-buffer adapter --------------
def pack_gbuffer(position, normal, bary, view_proj, prev_view_proj, cam):
    """resources/shaders/GBuffer.frag:62-88 (+ GBuffer.vert:21-34) from linear attribute planes, static geometry.
    Matrices: 16 floats, column-major.  Unfused fp32 in the order of svgf_amd/csrc (bit-exact contract)."""
    H, W = position.shape[:2]
    p = position.astype(np.float32)
    n = normal.astype(np.float32)
    covered = ~((n[..., 0] == 0) & (n[..., 1] == 0) & (n[..., 2] == 0))

    def mul(m, q):
        m = np.asarray(m, np.float32)
        return [((m[r] * q[..., 0] + m[4 + r] * q[..., 1]) + m[8 + r] * q[..., 2]) + m[12 + r] for r in range(4)]
    cam = np.asarray(cam, np.float32)
    dx, dy, dz = cam[0] - p[..., 0], cam[1] - p[..., 1], cam[2] - p[..., 2]
    depth = np.sqrt((dx * dx + dy * dy) + dz * dz).astype(np.float32)
    with np.errstate(all="ignore"):
        cur, prev = mul(view_proj, p), mul(prev_view_proj, p)
        mvx = (prev[0] / prev[3] - cur[0] / cur[3]) * (f32(0.5) * f32(W))
        mvy = (prev[1] / prev[3] - cur[1] / cur[3]) * (f32(0.5) * f32(H))
    Y, X = np.mgrid[0:H, 0:W]
    xp, yp = X ^ 1, Y ^ 1
    okx, oky = xp < W, yp < H
    xpc, ypc = np.minimum(xp, W - 1), np.minimum(yp, H - 1)
    ddx = np.where(okx & covered[Y, xpc], np.abs(depth[Y, xpc] - depth), f32(0))
    ddy = np.where(oky & covered[ypc, X], np.abs(depth[ypc, X] - depth), f32(0))
    motion = np.stack([mvx, mvy, depth, np.fmax(ddx, ddy)], -1).astype(np.float32)      # GLSL max() leaves a NaN operand undefined; the GPUs' max instruction (and fmaxf in the kernel) drops it
    motion[~covered] = 0
    with np.errstate(all="ignore"):
        ln = np.sqrt((n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1]) + n[..., 2] * n[..., 2]).astype(np.float32)
        nn = np.stack([n[..., 0] / ln, n[..., 1] / ln, n[..., 2] / ln, n[..., 3]], -1).astype(np.float32)
    nout = nn.astype(np.float16).view(np.uint16)
    nout[~covered] = 0
    uvout = bary.astype(np.float32).astype(np.float16).view(np.uint16).copy()
    uvout[~covered] = 0
    return motion, nout, uvout


def albedo(mode, inp, alb):
    """Albedo demodulation (mode 0) / re-modulation (mode 1), SURVEY.md 8f-4 (the build's own definition; the reference has
    none, README.md:14).  inp, alb: (..., 4) arrays in the storage dtype; fp32 arithmetic, one rounding per operation."""
    c, a = inp.astype(np.float32), alb.astype(np.float32)
    d = np.fmax(a[..., :3], np.float32(1e-3))               # fmaxf: a NaN albedo reads as the floor (include/svgf.h)
    o = c.copy()
    o[..., :3] = (c[..., :3] / d) if mode == 0 else (c[..., :3] * d)
    return o.astype(inp.dtype)
