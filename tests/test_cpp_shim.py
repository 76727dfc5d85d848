"""include/SVGF.h (the C++ drop-in for the reference's src/SVGF.h) compiles with plain g++ against the C-ABI
library, and — on a GPU — drives six frames with hand-derivable results."""
import os
import subprocess

import pytest

from tests.conftest import ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "shim_frame")


def _build():
    from svgf_amd import build as b
    b.build_library()
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "cpp", "shim_frame.cpp"), "-o", EXE, "-L", os.path.join(ROOT, "svgf_amd"), "-lsvgf_mi355x",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "svgf_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_shim_compiles_with_gxx():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_shim_six_frames():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "shim ok" in r.stdout, r.stdout + r.stderr
