"""The oracle's software half converters against numpy's IEEE float16 (round-to-nearest-even),
i.e. the behaviour of __float2half/__half2float the reference stores with (Filter.cuh:15-52)."""
import ctypes as C

import numpy as np


def test_h2f_all_bit_patterns(oracle):
    lib = oracle.lib()
    bits = np.arange(0x10000, dtype=np.uint32).astype(np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([lib.svgf_oracle_h2f(int(b)) for b in bits], dtype=np.float32)
    finite = ~np.isnan(want)
    assert np.array_equal(got[finite].view(np.uint32), want[finite].view(np.uint32))
    assert np.all(np.isnan(got[~finite]))


def test_f2h_rne(oracle):
    lib = oracle.lib()
    rng = np.random.default_rng(7)
    # every half value, its neighbours' midpoints (ties) and random floats over the half range
    h = np.arange(0x7c00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = (h[:-1] + h[1:]) / 2
    cases = np.concatenate([
        h, mids, np.nextafter(mids.astype(np.float32), np.float32(np.inf)).astype(np.float64),
        np.nextafter(mids.astype(np.float32), np.float32(-np.inf)).astype(np.float64),
        rng.uniform(-70000, 70000, 20000), rng.uniform(-1, 1, 20000), rng.uniform(-1e-4, 1e-4, 20000),
        rng.uniform(-2e-7, 2e-7, 5000), [65504.0, 65519.9, 65520.0, 65536.0, 1e30, np.inf, -np.inf, 0.0, -0.0],
    ]).astype(np.float32)
    cases = np.concatenate([cases, -cases])
    with np.errstate(over="ignore"):
        want = cases.astype(np.float16).view(np.uint16)
    got = np.array([lib.svgf_oracle_f2h(C.c_float(float(v))) for v in cases], dtype=np.uint16)
    assert np.array_equal(got, want)
