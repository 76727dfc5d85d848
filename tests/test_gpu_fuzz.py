"""Pinned trials of the seeded sweep (tests/fuzz_parity.py): every seed that ever failed there, with what it found, and a few that never did.

What the sweep found in round 5 (all in frames whose G-buffer holds NaN / out-of-range texels, tests/gbuffer_poison.py):
  * a SKY texel (depth 0) holding a NaN normal: the general a-trous taps read it, turn NaN and send the pixel to the exact form; the
    uniform-normal taps never look at it — which of the two roundings the pixels around it got depended on what else the workgroup's tile
    held, so strips and row ranges differed from the whole frame in the last bit (svgf_device.h:commit_px now counts such a texel);
  * a reference normal (a workgroup's first texel) holding a NaN while every other counted texel of a tiny frame carried the same bits: the
    uniform form's exponent bases were NaN for the sky centres too, which are to be copied (svgf_atrous_lds.h: no uniform form then);
  * the young-pixel launch dropped only the NaN luminance / depth terms of a window, the streaming moments kernel evaluates the whole pixel
    the reference's way once its sums hold a NaN: the frame driver (which picks one of the two per frame) differed from itself with
    svgf_set_adaptive_moments(0), and from the stage calls (svgf_kernels.hip:moments_group8 now follows the streaming kernel's rule);
  * svgf_modulate in fp16 storage turned -0 x albedo into +0 in one channel: hipcc folds a product that is only rounded to half into
    v_fma_mixlo_f16 with a +0 addend (svgf_kernels.hip:albedo_kernel pins the products in fp32 registers);
  * PhiColour = 0 (the GUI's drag starts there, GUI.cpp:992): |dl| / 0 is inf, or NaN for the taps of the centre's own luminance — the centre
    among them — and `max(., 0.0)` = fmax reads that as "no term" (Filter.cuh:424); the streaming moments kernel wrote NaN for every young pixel
    (its second, exact evaluation was tied to a non-finite TEXEL having been staged; svgf_moments_lds.h now also takes it when 1 / PhiColour is inf);
  * the sign of a zero through the value clamp (seed 4007112 of kind stage0): `(x < 0) ? 0 : x` keeps -0.0, the hardware's result clamp returned
    +0.0 — fixed in round 6 (svgf_device.h: clamp01_ref; commit_px + atrous_band for the copied sky texels); the sweep compares raw bits again."""
import pytest

pytestmark = pytest.mark.gpu

STRIPS = [1192, 1483, 1531, 1621, 1651, 1969, 2311, 2329, 2755, 3841, 5230, 6463, 40000, 40003, 40006]
DRIVER = [7313, 1916, 1640, 1787, 2843, 2993, 5447, 7034, 24140, 40001, 40004, 40007, 40010]
STAGE = [6657, 1266, 3795, 3807, 5883, 2529, 40002, 40005, 40008, 40011]
OTHER = [("post", 213350), ("post", 215738), ("post", 320621), ("post", 400001), ("rows", 400000), ("rows", 400003), ("rows", 400006), ("rows", 400009),
         ("pair", 400004), ("pair", 400007), ("pair", 400010), ("pair", 400013), ("stage0", 500000), ("stage0", 500001), ("stage0", 500002), ("stage0", 500003), ("stage0", 4007112),
         ("strips2", 600000), ("strips2", 600001), ("strips2", 600002), ("strips2", 600003), ("strips2", 600004), ("strips2", 600005),
         ("edge", 955563), ("edge", 955568), ("edge", 955878), ("edge", 950042), ("edge", 955806), ("edge", 950208), ("edge", 960000), ("edge", 960001),
         ("graph", 1100000), ("graph", 1100001), ("graph", 1100002), ("graph", 1100003), ("graph", 1100004), ("fullsize", 1200001),
         ("edgestrips", 990000), ("edgestrips", 990001), ("edgestrips", 990011), ("edgestrips", 990002),
         ("edgedriver", 970000), ("edgedriver", 970001), ("edgedriver", 970002), ("edgedriver", 970003),
         ("wide", 800000), ("wide", 800002), ("widestrips", 800001), ("widestrips", 800003),
         ("driver2", 700000), ("driver2", 700001), ("driver2", 700002), ("driver2", 700003), ("driver2", 700004), ("driver2", 700005)]


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


@pytest.mark.parametrize("seed", STRIPS)
def test_strips_equal_the_frame_driver(G, oracle, seed):
    from tests import fuzz_parity
    fuzz_parity.run_trial("strips", seed, G, oracle)


@pytest.mark.parametrize("seed", DRIVER)
def test_frame_driver_settings_and_stage_calls_give_the_same_bits(G, oracle, seed):
    from tests import fuzz_parity
    fuzz_parity.run_trial("driver", seed, G, oracle)


@pytest.mark.parametrize("seed", STAGE)
def test_stages_against_the_oracle(G, oracle, seed):
    from tests import fuzz_parity
    fuzz_parity.run_trial("stage", seed, G, oracle)


@pytest.mark.parametrize("kind,seed", OTHER)
def test_the_kinds_added_later(G, oracle, kind, seed):
    from tests import fuzz_parity
    fuzz_parity.run_trial(kind, seed, G, oracle)
