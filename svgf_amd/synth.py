"""Deterministic synthetic G-buffer + 1-spp radiance frames.

The reference's inputs come from an OpenGL rasteriser (resources/shaders/GBuffer.frag:62-88)
and a CUDA path tracer (src/PathTrace.cuh:618-619); neither exists on the compute box, so
tests, smoke() and bench.py feed the filter with frames of the same *plane formats*:

  motion  float32[rows, W, 4]  (mv.x, mv.y, depth, ddepth)   mv in pixels, prev - cur
                               (GBuffer.frag:67-71,81-82); depth == 0 marks sky (App.cu:383)
  normal  uint16 [rows, W, 4]  IEEE-half bits (nx, ny, nz, matID)        (GBuffer.frag:65,78,85)
  uv      uint16 [rows, W, 4]  IEEE-half bits (b0, b1, b2, instanceID)   (GBuffer.frag:64,77,86)
  radiance float32[rows, W, 4] (r, g, b, 1)                              (PathTrace.cuh:618)

Noise is counter based (splitmix64 keyed on seed, frame, y, x, channel): any row range of any
frame can be produced independently, so strips generated on different ranks tile exactly into
the single-GPU frame.  SEED is fixed and recorded in bench output.
"""
from __future__ import annotations

import numpy as np

SEED = 0x5356474600000001  # "SVGF" 0 0 0 1

_U64 = np.uint64


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + _U64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
    return z ^ (z >> _U64(31))


def uniform01(seed: int, frame: int, ys: np.ndarray, xs: np.ndarray, channel: int) -> np.ndarray:
    """float32 uniform in [0,1) for every (y, x) of the outer product ys × xs."""
    with np.errstate(over="ignore"):
        ky = _splitmix64(ys.astype(np.uint64) * _U64(0xD1B54A32D192ED03) + _U64(seed & 0xFFFFFFFFFFFFFFFF))
        kx = _splitmix64(xs.astype(np.uint64) * _U64(0x8CB92BA72F3D8DD7) + _U64((frame * 0x100 + channel) & 0xFFFFFFFFFFFFFFFF))
        h = _splitmix64(ky[:, None] ^ (kx[None, :] * _U64(0x9E3779B97F4A7C15)))
    return ((h >> _U64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


# region ids (instanceID); matID = id + 10
SKY, QUAD_A, QUAD_B, SPHERE, GROUND, QUAD_A2 = 0, 1, 2, 3, 4, 5

_ALBEDO = np.array(
    [[0.0, 0.0, 0.0], [0.80, 0.25, 0.20], [0.20, 0.55, 0.85], [0.90, 0.85, 0.30], [0.45, 0.50, 0.40], [0.30, 0.75, 0.35]],
    dtype=np.float64,
)
_N_GROUND = np.array([0.0, 0.9486833, -0.31622777])
_N_QUAD_A = np.array([0.0, 0.0, -1.0])
_N_QUAD_B = np.array([0.6, 0.0, -0.8])


def _curved_geometry(u, v, H):
    """The "curved" scene: every surface is smooth-shaded — what GBuffer.frag:65 writes for a mesh with interpolated vertex normals,
    `normalize(FragNormal)`, a different normal in (nearly) every texel.  Rolling terrain instead of the ground plane, a large sphere and an
    upright cylinder instead of the quads; depth and its screen-space derivative analytic.  -> region, z, dz, n."""
    a, ku, kv = 0.35, 9.0, 7.0
    region = np.full(u.shape, GROUND, dtype=np.int32)
    z = 12.0 + 3.0 * u - 6.0 * (v - 0.5) + a * np.sin(ku * u) * np.cos(kv * v)
    zu = 3.0 + a * ku * np.cos(ku * u) * np.cos(kv * v)
    zv = -6.0 - a * kv * np.sin(ku * u) * np.sin(kv * v)
    dz = np.maximum(np.abs(zu), np.abs(zv)) / H
    n = np.stack([-0.1 * zu, 0.9486833 - 0.05 * (zv + 6.0), np.full_like(u, -0.31622777) - 0.02 * zu], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    # sphere
    cx, cy, r = 0.55, 0.62, 0.30
    dx, dy = (u - cx) / r, (v - cy) / r
    rho2 = dx * dx + dy * dy
    m = rho2 < 1.0
    if m.any():
        s = np.sqrt(np.clip(1.0 - rho2, 0.0, 1.0))
        region[m] = SPHERE
        z[m] = (6.0 - 2.0 * r * s)[m]
        dz[m] = (2.0 * np.maximum(np.abs(dx), np.abs(dy)) / np.maximum(s, 0.2) / H)[m]
        n[m] = np.stack([dx[m], dy[m], -s[m]], -1)
    # upright cylinder (axis along v)
    cc, rc = 1.25, 0.2
    t = (u - cc) / rc
    m = (np.abs(t) < 1.0) & (v >= 0.15) & (v < 0.9)
    if m.any():
        c = np.sqrt(np.clip(1.0 - t * t, 0.0, 1.0))
        region[m] = QUAD_B
        z[m] = (7.0 - 2.0 * rc * c)[m]
        dz[m] = (2.0 * np.abs(t) / np.maximum(c, 0.2) / H)[m]
        n[m] = np.stack([t[m], np.zeros(int(m.sum())), -c[m]], -1)
    return region, z, dz, n


SCENES = ("planar", "curved")


def _scene_rows(width, height, frame, mv, row_begin, row_end, seed, col_begin=0, scene="planar"):
    """Geometry of rows [row_begin,row_end) x columns [col_begin, col_begin+width): motion/normal/uv planes, region ids and
    the noise-free radiance.  `height` is the scale of the world coordinates (the frame height), whatever the ranges.
    scene: "planar" (a tilted ground plane, quads, one sphere — piecewise constant normals) or "curved" (_curved_geometry)."""
    rows = row_end - row_begin
    ys = np.arange(row_begin, row_end, dtype=np.int64)
    xs = np.arange(col_begin, col_begin + width, dtype=np.int64)
    H = float(height)
    u = np.empty((rows, width), np.float64)
    v = np.empty((rows, width), np.float64)
    u[...] = ((xs.astype(np.float64) + frame * float(mv[0])) / H)[None, :]
    v[...] = ((ys.astype(np.float64) + frame * float(mv[1])) / H)[:, None]

    if scene == "curved":
        region, z, dz, n = _curved_geometry(u, v, H)
        return _finish_rows(width, rows, ys, xs, u, v, mv, seed, region, z, dz, n)
    if scene != "planar":
        raise ValueError(scene)
    region = np.full((rows, width), GROUND, dtype=np.int32)
    z = 12.0 + 3.0 * u - 6.0 * (v - 0.5)
    dz = np.full_like(z, 6.0 / H)
    n = np.empty((rows, width, 3), np.float64)
    n[...] = _N_GROUND

    # sphere
    cx, cy, r = 0.45, 0.78, 0.18
    dx, dy = (u - cx) / r, (v - cy) / r
    rho2 = dx * dx + dy * dy
    m = rho2 < 1.0
    if m.any():
        s = np.sqrt(np.clip(1.0 - rho2, 0.0, 1.0))
        region[m] = SPHERE
        z[m] = (6.0 - 2.0 * r * s)[m]
        sd = np.maximum(s, 0.2)
        dz[m] = (2.0 * np.maximum(np.abs(dx), np.abs(dy)) / sd / H)[m]
        n[m] = np.stack([dx[m], dy[m], -s[m]], -1)

    # tilted quad B
    m = (u >= 0.9) & (u < 1.4) & (v >= 0.2) & (v < 0.7)
    if m.any():
        region[m] = QUAD_B
        z[m] = (7.0 + 1.5 * (u - 0.9))[m]
        dz[m] = 1.5 / H
        n[m] = _N_QUAD_B

    # fronto-parallel quad A and its coplanar twin A2 (differs by instanceID only)
    m = (u >= 0.2) & (u < 0.6) & (v >= 0.25) & (v < 0.55)
    if m.any():
        region[m] = QUAD_A
        z[m] = 4.0
        dz[m] = 0.0
        n[m] = _N_QUAD_A
    m = (u >= 0.6) & (u < 0.7) & (v >= 0.25) & (v < 0.55)
    if m.any():
        region[m] = QUAD_A2
        z[m] = 4.0
        dz[m] = 0.0
        n[m] = _N_QUAD_A

    return _finish_rows(width, rows, ys, xs, u, v, mv, seed, region, z, dz, n)


def _finish_rows(width, rows, ys, xs, u, v, mv, seed, region, z, dz, n):
    """The sky band, the plane formats and the noise-free radiance of a scene's geometry."""
    # sky band (>= 5 % of the frame): depth 0, normal/uv all-zero bits
    sky = v < (0.08 + 0.02 * np.sin(7.0 * u))
    if sky.any():
        region[sky] = SKY
        z[sky] = 0.0
        dz[sky] = 0.0
        n[sky] = 0.0

    motion = np.empty((rows, width, 4), dtype=np.float32)
    motion[..., 0] = np.float32(mv[0])
    motion[..., 1] = np.float32(mv[1])
    motion[..., 2] = z.astype(np.float32)
    motion[..., 3] = dz.astype(np.float32)

    normal = np.empty((rows, width, 4), dtype=np.uint16)
    normal[..., :3] = n.astype(np.float32).astype(np.float16).view(np.uint16)
    normal[..., 3] = np.where(sky, 0, region + 10).astype(np.float16).view(np.uint16)

    uv = np.empty((rows, width, 4), dtype=np.uint16)
    b0 = uniform01(seed, 0, ys, xs, 8)
    b1 = uniform01(seed, 0, ys, xs, 9) * (1.0 - b0)
    uv[..., 0] = b0.astype(np.float16).view(np.uint16)
    uv[..., 1] = b1.astype(np.float16).view(np.uint16)
    uv[..., 2] = (1.0 - b0 - b1).astype(np.float16).view(np.uint16)
    uv[..., 3] = region.astype(np.float16).view(np.uint16)
    uv[sky] = 0

    # noise-free radiance = albedo(region) * shade(normal) * texture
    light = np.array([0.35, -0.5, -0.79])
    shade = 0.55 + 0.45 * np.clip(n @ light, 0.0, 1.0)
    tex = 0.8 + 0.2 * np.sin(37.0 * u) * np.sin(41.0 * v)
    base = _ALBEDO[region] * (shade * tex)[..., None] * 0.6 + 0.05
    base[sky] = np.array([0.25, 0.45, 0.80])
    return motion, normal, uv, region, base.astype(np.float32)


def _noise_rows(base, width, frame, row_begin, row_end, seed, noise):
    ys = np.arange(row_begin, row_end, dtype=np.int64)
    xs = np.arange(width, dtype=np.int64)
    radiance = np.empty(base.shape[:2] + (4,), dtype=np.float32)
    if noise == "1spp":        # a path either finds the light (p = 1/4, carrying 4x the radiance) or returns black
        hit = uniform01(seed, frame + 1, ys, xs, 0) < np.float32(0.25)
        val = np.where(hit[..., None], base * np.float32(4.0), np.float32(0.0))
    elif noise == "mul":
        k = uniform01(seed, frame + 1, ys, xs, 0)
        val = base * (np.float32(0.5) + k)[..., None]
    elif noise == "none":
        val = base
    else:
        raise ValueError(noise)
    radiance[..., :3] = np.clip(val, 0.0, 1.0)
    radiance[..., 3] = 1.0
    return radiance


_CHUNK = 128


def make_scene(width: int, height: int, frame: int = 0, *, mv=(0.0, 0.0), row_begin: int = 0, row_end: int | None = None,
               seed: int = SEED, col_begin: int = 0, col_end: int | None = None, scene: str = "planar"):
    """G-buffer planes + region ids + noise-free radiance ('base') of rows [row_begin,row_end) of frame `frame`.

    `mv` is the constant per-frame pan (prev - cur, pixels): the surface point seen at pixel p in frame f was at
    p + mv in frame f-1, i.e. frame f shows the static world at p + f*mv.  Rows and columns may lie outside the frame
    (a canvas larger than the frame: bench.py cuts the frames of a pan out of two such canvases)."""
    if row_end is None:
        row_end = height
    if col_end is None:
        col_end = width
    rows, width_out = row_end - row_begin, col_end - col_begin
    out = {"motion": np.empty((rows, width_out, 4), np.float32), "normal": np.empty((rows, width_out, 4), np.uint16),
           "uv": np.empty((rows, width_out, 4), np.uint16), "region": np.empty((rows, width_out), np.int32),
           "base": np.empty((rows, width_out, 3), np.float32)}
    for a in range(row_begin, row_end, _CHUNK):
        b = min(a + _CHUNK, row_end)
        mo, no, uv, rg, ba = _scene_rows(width_out, height, frame, mv, a, b, seed, col_begin, scene)
        sl = slice(a - row_begin, b - row_begin)
        out["motion"][sl], out["normal"][sl], out["uv"][sl], out["region"][sl], out["base"][sl] = mo, no, uv, rg, ba
    return out


def make_radiance(base: np.ndarray, width: int, frame: int, *, row_begin: int = 0, seed: int = SEED, noise: str = "1spp"):
    """1-spp style radiance {r,g,b,1} for frame `frame` from the noise-free `base` rows starting at row_begin."""
    rows = base.shape[0]
    out = np.empty((rows, width, 4), np.float32)
    for a in range(0, rows, _CHUNK):
        b = min(a + _CHUNK, rows)
        out[a:b] = _noise_rows(base[a:b], width, frame, row_begin + a, row_begin + b, seed, noise)
    return out


def make_frame(width: int, height: int, frame: int, *, mv=(0.0, 0.0), row_begin: int = 0, row_end: int | None = None,
               seed: int = SEED, noise: str = "1spp", scene: str = "planar"):
    """Rows [row_begin, row_end) of synthetic frame `frame`: dict(motion, normal, uv, radiance, region, base)."""
    sc = make_scene(width, height, frame, mv=mv, row_begin=row_begin, row_end=row_end, seed=seed, scene=scene)
    sc["radiance"] = make_radiance(sc["base"], width, frame, row_begin=row_begin, seed=seed, noise=noise)
    return sc


def to_storage(a: np.ndarray, storage: str) -> np.ndarray:
    """float32 colour/moment plane -> storage dtype ('f32' keeps, 'f16' rounds to nearest even)."""
    if storage == "f32":
        return np.ascontiguousarray(a, dtype=np.float32)
    if storage == "f16":
        return np.ascontiguousarray(a.astype(np.float16))
    raise ValueError(storage)
