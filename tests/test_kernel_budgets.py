"""The register / scratch budgets the launch geometry of the kernels is built on, read off the gfx950 ISA hipcc produces for the product sources
(cross-compiled here: no GPU needed).  A kernel that slips over its budget still computes the same bits — it just loses a resident wave per SIMD,
silently: round 4 met that twice (an inlined second pass that spilled in the a-trous hot loop, +15 %; a run-time cap that took the young-pixel
launch from 163 to 172 registers, 27 -> 34 us under the pan)."""
import os
import re
import subprocess

import pytest

from svgf_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vgpr_budget(waves_per_simd):
    """512 registers per SIMD lane, handed out in blocks of 8."""
    return 512 // waves_per_simd // 8 * 8


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "kernels.s")
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([build.hipcc(), *flags, "-S", "--cuda-device-only", "-o", out, os.path.join(build.CSRC, "svgf_kernels.hip")],
                          stderr=subprocess.DEVNULL)
    table, name = {}, None
    for line in open(out):
        m = re.match(r"\s+\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name = m.group(1)
            table[name] = {}
            continue
        m = re.match(r"\s+\.amdhsa_(next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size)\s+(\d+)", line)
        if m and name:
            table[name][m.group(1)] = int(m.group(2))
    assert table, "no kernels found in the ISA"
    return table


def pick(kernels, fragment):
    hits = {k: v for k, v in kernels.items() if fragment in k}
    assert hits, f"no kernel matches {fragment}"
    return hits


@pytest.mark.parametrize("fragment,waves,scratch_free", [
    # atrous_lds_kernel<ST, S, 128>: five resident waves per SIMD up to step 8, four at 16 (its ring allows four workgroups per CU), three at 32, two at 64
    *[(f"atrous_lds_kernelILi{st}ELi{s}ELi128E", w, True) for st in (0, 1) for s, w in ((1, 5), (2, 5), (4, 5), (8, 5), (16, 4), (32, 3), (64, 2))],
    ("moments_young_kernel", 3, True),       # a launch of latency-bound passes: every resident wave counts
    ("moments_lds_kernel", 4, True),         # 38.6 KB of LDS: four workgroups of four waves per CU
    ("temporal_kernel", 8, True),            # HBM-bound: as many waves in flight as the SIMD holds
    ("atrous_fused12_kernel", 4, True),
    ("taa_lds_kernel", 8, True),
])
def test_kernel_stays_inside_its_register_budget(kernels, fragment, waves, scratch_free):
    for name, k in pick(kernels, fragment).items():
        assert k["next_free_vgpr"] <= vgpr_budget(waves), f"{name}: {k['next_free_vgpr']} registers, {waves} waves per SIMD allow {vgpr_budget(waves)}"
        if scratch_free:
            assert k["private_segment_fixed_size"] == 0, f"{name}: {k['private_segment_fixed_size']} B of scratch"


def test_static_lds_of_the_young_pixel_launch(kernels):
    """Three workgroups per CU by registers: the static LDS (masks, work items of the over-the-cap path) must not be what limits them."""
    for name, k in pick(kernels, "moments_young_kernel").items():
        assert k["group_segment_fixed_size"] * 3 <= 160 * 1024 // 4, (name, k["group_segment_fixed_size"])
