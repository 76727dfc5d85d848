#!/usr/bin/env python3
"""Iterations 0 + 1 as one launch (svgf_atrous_pair) against one launch per iteration, interleaved in ONE process on ONE device:
    python tools/archive/ab_fuse.py [rounds] [workload ...]        (workloads: 4k 1080p 8k; default 4k 1080p)
Prints per round the frame time and the stage times (temporal, moments, iterations; the pair sits in the first iteration slot)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wls = sys.argv[2:] or ["4k", "1080p"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for wl in wls:
    W, H = bench.WORKLOADS[wl]
    scene = bench.Scene(W, H, dev, pool=2)
    for storage in os.environ.get("AB_STORAGE", "f32 f16").split():
        pool = bench.FramePool(scene, storage, "static")
        for r in range(rounds):
            for fuse in [m == "fused" for m in os.environ.get("AB_MODES", "fused single").split()]:
                res = bench.run_single(pool, W, H, storage, 5, os.environ.get("AB_VARIANT", "auto"), 40, 5, dev, fuse=fuse)
                print(f"{os.environ.get('AB_TAG', '')} {wl} {storage} {'fused ' if fuse else 'single'} {res['ms_per_step']:.4f} ms  stages {[round(m, 4) for m in res['stage_ms']]}", flush=True)
