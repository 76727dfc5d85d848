"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/svgf.h
declares, and refuses to work (loudly) without a GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "svgf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svgf_[a-z_]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    from svgf_amd import filter as F
    lib = F.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/svgf.h but not exported"
    assert sorted(F.EXPORTS) == syms
    assert lib.svgf_abi_version() == 1


def test_default_params_match_reference_defaults():
    from svgf_amd import filter as F
    lib = F.load_library()
    p = F.ParamsC()
    lib.svgf_default_params(C.byref(p))
    # src/App.h:109-114
    assert (p.steps, p.history_base, p.moments_radius) == (3, 24, 3)
    assert (round(p.depth_threshold, 6), round(p.normal_threshold, 6), p.phi_colour, p.phi_normal) == (0.8, 0.9, 10.0, 128.0)
    assert p.storage == F.SVGF_F16


def test_no_cpu_fallback():
    import torch
    from svgf_amd import filter as F
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(F.SvgfError, match="no CPU fallback"):
        F.Denoiser(64, 64)
    lib = F.load_library()
    h = C.c_void_p()
    p = F.Params().to_c()
    rc = lib.svgf_create(C.byref(h), 64, 64, C.byref(p), 0, None)
    assert rc == -3 and not h.value                               # SVGF_ERR_NO_DEVICE
    assert lib.svgf_status_string(rc) == b"no usable gfx950 device"


def test_product_never_touches_the_oracle():
    """The product package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "svgf_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.+oracle|oracle[/\\.]|libsvgf_oracle|#include[^\n]*oracle|svgf_oracle_)")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                m = bad.search(src)
                assert not m, f"{f} references the oracle: {m.group(0)!r}"
