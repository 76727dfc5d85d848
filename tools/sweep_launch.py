#!/usr/bin/env python3
"""Launch-geometry sweep of the à-trous kernel in ONE process on ONE device (devices differ by +-4 %, so configurations are
compared inside one run and repeated): needs a library built with -DSVGF_DIAG (SVGF_LIBRARY=build/libsvgf_DG.so), whose launch
code reads SVGF_ATROUS_ONLY_STEP / SVGF_ATROUS_SLOTS_S / SVGF_ATROUS_MIN_BAND / SVGF_ATROUS_XM from the environment at every launch.

    SVGF_LIBRARY=build/libsvgf_DG.so python3 tools/sweep_launch.py --size 1920x1080 --steps 1,2,4,8,16 --slots 1280,2560,5120 --min-band 4,8,16
    ... --rows 678 --width 7680      # the planes of one 8K strip (540 rows + halos)
Prints, per step, the launch time of every (slots, min band) pair, rounds interleaved.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--steps", default="1,2,4,8,16")
    ap.add_argument("--slots", default="1280,2560,3840,5120")
    ap.add_argument("--min-band", default="8")
    ap.add_argument("--xm", default="")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--storage", default="f32")
    args = ap.parse_args()
    import torch
    import bench
    from svgf_amd import filter as F
    W, H = (int(v) for v in args.size.split("x"))
    dev = torch.device("cuda", 0)
    scene = bench.Scene(W, H, dev)
    pool = bench.FramePool(scene, args.storage, "static")
    d = F.Denoiser(W, H, F.Params(storage=args.storage, steps=5), device=0)
    n = 0
    for _ in range(40):
        d.Render(*pool.frame(n)); n += 1

    def measure():
        nonlocal n
        d.timing_enable(1)
        torch.cuda.synchronize()
        for _ in range(args.frames):
            d.Render(*pool.frame(n)); n += 1
        ms, frames = d.timing_read()
        d.timing_enable(False)
        return [m / max(frames, 1) for m in ms]

    keys = ("SVGF_ATROUS_ONLY_STEP", "SVGF_ATROUS_SLOTS_S", "SVGF_ATROUS_MIN_BAND", "SVGF_ATROUS_XM")
    def setenv(**kw):
        for k in keys:
            os.environ.pop(k, None)
        for k, v in kw.items():
            if v is not None:
                os.environ[k] = str(v)

    steps = [int(v) for v in args.steps.split(",")]
    slots = [int(v) for v in args.slots.split(",")]
    bands = [int(v) for v in args.min_band.split(",")]
    xms = [int(v) for v in args.xm.split(",")] if args.xm else [None]
    setenv()
    base = measure()
    print(f"{W}x{H} {args.storage}: product configuration: stages {[round(m, 4) for m in base]} sum {sum(base):.4f}", flush=True)
    for s in steps:
        i = 2 + steps_index(s)
        res = {}
        for r in range(args.rounds):
            setenv()
            res.setdefault("product", []).append(measure()[i])
            for sl in slots:
                for mb in bands:
                    for xm in xms:
                        setenv(SVGF_ATROUS_ONLY_STEP=s, SVGF_ATROUS_SLOTS_S=sl, SVGF_ATROUS_MIN_BAND=mb, SVGF_ATROUS_XM=xm)
                        res.setdefault((sl, mb, xm), []).append(measure()[i])
        print(f"step {s}:")
        for k, v in sorted(res.items(), key=lambda kv: min(kv[1])):
            print(f"   {str(k):28s} min {min(v):.4f}  all {[round(x, 4) for x in v]}", flush=True)
    setenv()
    d.close()


def steps_index(s):
    return {1: 0, 2: 1, 4: 2, 8: 3, 16: 4}[s]


if __name__ == "__main__":
    main()
