"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/svgf.h, svgf_ext.h and svgf_test.h
declare, and refuses to work (loudly) without a GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT


HEADERS = ("svgf.h", "svgf_ext.h", "svgf_test.h")     # what a host binds; opt-ins and diagnostics; test hooks — one library


def _declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svgf_[a-z_]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    from svgf_amd import filter as F
    lib = F.load_library()
    per = {h: _declared_symbols(h) for h in HEADERS}
    assert len(per["svgf.h"]) >= 18
    for h, syms in per.items():
        for s in syms:
            assert hasattr(lib, s), f"{s} declared in include/{h} but not exported"
    assert not (set(per["svgf.h"]) & set(per["svgf_ext.h"])) and not (set(per["svgf.h"]) & set(per["svgf_test.h"])) and not (set(per["svgf_ext.h"]) & set(per["svgf_test.h"]))
    assert sorted(F.EXPORTS) == sorted(s for syms in per.values() for s in syms)
    assert lib.svgf_abi_version() == F.ABI_VERSION == 8
    # the header a host of the reference reads stays short, and the test hooks stay out of it
    assert len(open(os.path.join(ROOT, "include", "svgf.h")).read().splitlines()) <= 260
    assert "mailbox_fault" not in open(os.path.join(ROOT, "include", "svgf.h")).read()


def test_default_params_match_reference_defaults():
    from svgf_amd import filter as F
    lib = F.load_library()
    p = F.ParamsC()
    lib.svgf_default_params(C.byref(p))
    # src/App.h:109-114
    assert (p.steps, p.history_base, p.moments_radius) == (3, 24, 3)
    assert (round(p.depth_threshold, 6), round(p.normal_threshold, 6), p.phi_colour, p.phi_normal) == (0.8, 0.9, 10.0, 128.0)
    assert p.storage == F.SVGF_F16
    assert p.nan_policy == 0 and F.NAN_POLICY["reference"] == 0      # a NaN texel behaves as in Filter.cuh unless the host asks otherwise


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


def test_strip_plan_matches_python_geometry():
    """svgf_strips_plan (C++, what the strip driver runs on) == svgf_amd.strips.Geometry.make (what the CPU gloo tests run on),
    over frame sizes, world sizes, iteration counts, plans, radii and motion reaches; and the same refusals."""
    from tests.strip_geometry_checks import check_plan_against_python
    n = 0
    for (W, H) in ((7680, 4320), (3840, 2160), (320, 420), (96, 312), (64, 200)):
        for world in (1, 2, 3, 8):
            for steps in (0, 1, 3, 5):
                for plan in ("ghost", "grouped", "per-iteration", "auto"):
                    for mr, reach in ((3, 4), (1, 0), (3, 9)):
                        n += check_plan_against_python(W, H, world, steps, plan, mr, reach, ranks=sorted({0, world // 2, world - 1}))
    assert n > 500


def test_every_posted_send_has_its_mirror_receive_on_the_neighbour():
    """svgf_strips_messages is what svgf_strips_frame hands to its transport (svgf_strip.hip: exchange_msgs).  For world = 2..8, every plan,
    both storages, frame sizes with uneven strips and several motion reaches: within every exchange, the sends of rank a to rank b and the
    receives rank b posts with peer a are the SAME sequence of (plane, global rows, bytes) — RCCL matches the sends and receives of a pair
    of ranks in posting order, so an unmatched or mis-ordered message is a deadlock (or rows in the wrong place) on a real node.  Also: peers
    are direct neighbours, a rank receives only rows outside its own strip and inside what it holds, sends only rows it owns.
    (tests/strip_geometry_checks.py; `python -m tests.strip_geometry_checks --minutes 2` sweeps random partitions with the same checks.)"""
    from tests.strip_geometry_checks import check_messages
    cases = 0
    for (W, H) in ((7680, 4320), (3840, 2160), (640, 1003), (96, 700)):
        for world in range(2, 9):
            for steps in (0, 1, 3, 5):
                for plan in ("ghost", "grouped", "per-iteration", "auto"):
                    for mr, reach in ((3, 4), (1, 0), (3, 9)):
                        for storage in ("f32", "f16"):
                            cases += check_messages(W, H, world, steps, plan, mr, reach, storage)
    assert cases > 1000


def test_random_partitions_seeded():
    """The same two checks on 400 random partitions (world up to 16, 0-10 iterations, moments radius 0-3, motion reach 0-24, ragged sizes)."""
    from tests.strip_geometry_checks import random_case
    done = sum(random_case(seed) for seed in range(400))
    assert done > 150          # (the others: partitions the plan refuses — checked to be refused by both sides)


def test_strip_driver_refuses_without_gpu_or_bad_arguments():
    import torch
    from svgf_amd import filter as F
    lib = F.load_library()
    lay = F.StripLayoutC()
    assert lib.svgf_strips_plan(64, 64, 0, 4, 5, F.HALO_PLAN["ghost"], 3, 4, C.byref(lay)) == -4          # SVGF_ERR_HALO: 16-row strips, 69-row halo
    assert lib.svgf_strips_plan(64, 64, 5, 4, 5, 0, 3, 4, C.byref(lay)) == -1                              # rank outside the world
    assert lib.svgf_status_string(-6) == b"RCCL error"
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    p = F.Params(storage="f32", steps=5).to_c()
    ranks, devs = (C.c_int * 1)(0), (C.c_int * 1)(0)
    rc = lib.svgf_strips_create(C.byref(h), 640, 480, 1, C.byref(p), 0, 0, 1, ranks, devs, None, None, 0)
    assert rc == -3 and not h.value                                                                        # no device: no driver


def test_headers_are_plain_c():
    """The boundary is a C ABI: each of the three headers compiles on its own as C99 (gcc -pedantic), the C++ shim as C++17 against them."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    with tempfile.TemporaryDirectory() as tmp:
        for h in HEADERS:
            src = os.path.join(tmp, "t.c")
            open(src, "w").write(f'#include "{h}"\nint main(void) {{ return SVGF_ABI_VERSION == 8 ? 0 : 1; }}\n')
            subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, src])
