"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
    python tests/golden/make_golden.py).
The reference holds no vectors for this path, so these pin the ORACLE's behaviour (and the synthetic
generator) against silent drift; GPU tests compare the HIP path with them as well."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from svgf_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def atrous_single(W=96, H=64):
    """BASELINE config #1 in miniature: one à-trous iteration (step 1) on a noisy synthetic frame, fp32 and fp16."""
    f = synth.make_frame(W, H, 0)
    rng = np.random.default_rng(2024)
    src = np.concatenate([f["radiance"][..., :3], rng.uniform(0.0, 0.05, (H, W, 1)).astype(np.float32)], -1)
    out = {}
    for st, dt in (("f32", np.float32), ("f16", np.float16)):
        s = src.astype(dt)
        for step in (1, 4):
            o = np.zeros_like(s)
            orc.atrous(W, H, st, s, o, None, {k: f[k] for k in ("motion", "normal", "uv")}, step=step, phi_colour=10.0,
                       phi_normal=128.0, iteration=1)
            out[f"out_{st}_step{step}"] = o
    np.savez_compressed(os.path.join(HERE, "atrous_96x64.npz"), motion=f["motion"], normal=f["normal"], uv=f["uv"], src=src, **out)


def pipeline(W=64, H=48, N=8, mv=(-2.5, 1.5)):
    """8-frame panning sequence through the whole path (history feedback), final frame + history + frame-3 output."""
    out = {}
    for st in ("f32", "f16"):
        p = orc.Pipeline(W, H, st, steps=5)
        frs = [synth.make_frame(W, H, k, mv=mv) for k in range(N)]
        for k in range(N):
            gb = {n: frs[k][n] for n in ("motion", "normal", "uv")}
            gp = {n: frs[max(k - 1, 0)][n] for n in ("motion", "normal", "uv")}
            o = p.frame(frs[k]["radiance"], gb, gp)
            if k in (3, N - 1):
                out[f"out_{st}_frame{k}"] = o.copy()
                out[f"hist_{st}_frame{k}"] = p.hist[p.P ^ 1].copy()
    np.savez_compressed(os.path.join(HERE, "pipeline_64x48.npz"), W=W, H=H, N=N, mv=np.array(mv), **out)


def config1(W=256, H=256):
    """BASELINE.json configs[0] at full size: 256x256 synthetic G-buffer + noisy radiance, ONE a-trous iteration (step 1) through
    the scalar C++ loop.  Only a strided sample of the output and its statistics are stored (the inputs are regenerated from
    the seeded generator, whose planes are pinned by checksums)."""
    f = synth.make_frame(W, H, 0)
    rng = np.random.default_rng(256)
    src = np.concatenate([f["radiance"][..., :3], rng.uniform(0.0, 0.05, (H, W, 1)).astype(np.float32)], -1)
    gb = {k: f[k] for k in ("motion", "normal", "uv")}
    out = {}
    for st, dt in (("f32", np.float32), ("f16", np.float16)):
        s = src.astype(dt)
        o = np.zeros_like(s); fb = np.zeros_like(s)
        orc.atrous(W, H, st, s, o, fb, gb, step=1, phi_colour=10.0, phi_normal=128.0, iteration=0)
        out[f"sample_{st}"] = o[::8, ::8].copy()
        out[f"mean_{st}"] = o.astype(np.float64).mean((0, 1))
        out[f"fb_equal_{st}"] = np.array(np.array_equal(fb[f["region"] != synth.SKY], o[f["region"] != synth.SKY]))
    sums = np.array([int(np.frombuffer(f[k].tobytes(), np.uint8).astype(np.uint64).sum()) for k in ("motion", "normal", "uv", "radiance")], np.uint64)
    np.savez_compressed(os.path.join(HERE, "config1_256x256.npz"), input_sums=sums, **out)


if __name__ == "__main__":
    atrous_single()
    pipeline()
    config1()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
