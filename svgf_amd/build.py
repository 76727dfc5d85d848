"""Builds libsvgf_mi355x.so (the C-ABI product library) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsvgf_mi355x.so")
SOURCES = ["svgf_kernels.hip", "svgf_api.hip", "svgf_strip.hip"]
HEADERS = ["svgf_kernels.h", "svgf_ctx.h", "svgf_device.h", "svgf_atrous_taps.h", "svgf_atrous_lds.h", "svgf_atrous_fused.h", "svgf_moments_lds.h", os.path.join("..", "..", "include", "svgf.h"),
           os.path.join("..", "..", "include", "svgf_ext.h"), os.path.join("..", "..", "include", "svgf_test.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsvgf_mi355x.so cannot be built (there is no CPU fallback)")


def have_hipcc() -> bool:
    try:
        hipcc()
        return True
    except RuntimeError:
        return False


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Default: the product library.  `out` + extra_flags (e.g. -DSVGF_DIAG) build a diagnostic twin elsewhere."""
    if out is None and not force and not stale():
        return LIB
    out = out or LIB
    cmd = [hipcc(), *FLAGS, *extra_flags, "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
