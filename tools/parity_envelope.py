"""The envelope of the free-running parity tests: how far apart do two CORRECT implementations of Filter.cuh end up?

Runs the oracle (fp64 islands kept, no FMA contraction) and its envelope build (oracle/Makefile: -DSVGF_ORACLE_ALL_FP32 -ffp-contract=fast
-mfma — every island in fp32, a*b+c contracted as nvcc's defaults would) free for 8 frames on the frames of
tests/test_gpu_parity.py::test_pipeline_free_running, each feeding itself, and records per storage / motion: the largest colour difference,
the fraction of values beyond the tight tolerance, accept / reject mask mismatches.  CPU only.  -> profiles/r05_parity_envelope.json, which
the GPU tests read: HIP-vs-oracle must sit inside oracle-vs-oracle'.
usage: python tools/parity_envelope.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc          # noqa: E402
from tests.helpers import frames, gbuf    # noqa: E402

TIGHT = {"f32": 2e-5, "f16": 1e-3}


def run(W=256, H=144, N=8, steps=5):
    rep = {"frame": f"{W}x{H}", "frames": N, "steps": steps,
           "builds": {"oracle": "fp64 islands, -ffp-contract=off (the checker)", "fp32": "all fp32, -ffp-contract=off", "fma": "fp64 islands, -ffp-contract=fast -mfma",
                      "fp32fma": "all fp32, -ffp-contract=fast -mfma (nvcc's defaults on the reference's source)",
                      "fused": "fp32fma + the weight as one exp2 of a fused fp32 exponent, divisions as reciprocal multiplies (the HIP kernels' formulation, libm-evaluated)"}, "cases": {}}
    for storage in ("f32", "f16"):
        for mv in ((0.0, 0.0), (-2.5, 1.5)):
            fr = frames(W, H, N, mv=mv)
            case = {}
            for flavour in ("fp32", "fma", "fp32fma", "fused"):
                a = orc.Pipeline(W, H, storage, steps=steps, nthreads=8)
                b = orc.Pipeline(W, H, storage, steps=steps, nthreads=8)
                per_frame, worst, worst_frac, mism = [], 0.0, 0.0, 0
                for k in range(N):
                    kp = max(k - 1, 0)
                    wa = a.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp])).astype(np.float64)
                    ha = a.taps["hist"].copy()
                    with orc.using(flavour):
                        wb = b.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp])).astype(np.float64)
                    mism += int((ha != b.taps["hist"]).sum())
                    err = np.abs(wa - wb)[..., :3]
                    frac = float((err > TIGHT[storage] + 1e-5 * np.abs(wa[..., :3])).mean())
                    per_frame.append({"max_abs": float(err.max()), "frac_beyond_tight": frac, "variance_max_abs": float(np.abs(wa - wb)[..., 3].max())})
                    worst, worst_frac = max(worst, float(err.max())), max(worst_frac, frac)
                case[flavour] = {"max_abs": worst, "frac_beyond_tight": worst_frac, "mask_mismatches": mism, "per_frame": per_frame}
            rep["cases"][f"{storage} mv={list(mv)}"] = case
    return rep


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_parity_envelope.json")
    rep = run()
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    for name, case in rep["cases"].items():
        for flavour, c in case.items():
            print(f"{name:22s} oracle vs {flavour:8s}: max {c['max_abs']:.3e}  frac beyond tight {c['frac_beyond_tight']:.2e}  mask mismatches {c['mask_mismatches']}")
