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
                      "fused": "fp32fma + the weight as one exp2 of a fused fp32 exponent, divisions as reciprocal multiplies (the HIP kernels' formulation, libm-evaluated)",
                      "hwulp": "fused + log2 / exp2 / the two reciprocals of the weight moved by -1 / 0 / +1 ulp (a model of v_log_f32 / v_exp_f32 / v_rcp_f32: 1-ulp approximations)"}, "cases": {}}
    for storage in ("f32", "f16"):
        for mv in ((0.0, 0.0), (-2.5, 1.5)):
            fr = frames(W, H, N, mv=mv)
            case = {}
            from tests.helpers import free_running_envelope
            for flavour in ("fp32", "fma", "fp32fma", "fused", "hwulp"):
                # (hwulp: the largest distance over tests.helpers.HW_ULP_SEEDS assignments of the 1-ulp nudges, every seed's own maximum in per_seed)
                case[flavour] = free_running_envelope(orc, fr, storage, steps=steps, flavour=flavour, tight=TIGHT[storage])
            rep["cases"][f"{storage} mv={list(mv)}"] = case
    return rep


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_parity_envelope.json")
    rep = run()
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    for name, case in rep["cases"].items():
        for flavour, c in case.items():
            print(f"{name:22s} oracle vs {flavour:8s}: max {c['max_abs']:.3e}  frac beyond tight {c['frac_beyond_tight']:.2e}  mask mismatches {c['mask_mismatches']}")
