"""The a-trous kernel's launch geometry, emulated on the host, against the device's own counters (CPU test; no GPU).

`bench.py` reports which share of the wave-steps of the streaming a-trous launches took the uniform-normal tap path, counted on the device
(svgf_path_stats_enable).  tools/uniform_share_emul.py derives the same share from the launch geometry alone — 128-column tiles, row residues, the
bands of cut_bands, the six-row ring window, the workgroup's reference normal (svgf_atrous_lds.h) — on the same synthetic scene.  The committed
bench line and the emulation must agree: if they do not, either the kernel's tiling changed and the profile set is stale, or the counters count
something else than the docs say."""
import json
import os
import sys

from tests.conftest import ROOT


def _recorded():
    path = os.path.join(ROOT, "profiles", "r06_bench_4k_f32.json")
    return json.loads([ln for ln in open(path) if ln.startswith("{")][-1])


def test_emulated_share_of_the_fast_path_equals_the_device_counters():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import uniform_share_emul as emul
    finally:
        sys.path.pop(0)
    d = _recorded()
    assert d["config"]["width"] == 3840 and d["config"]["height"] == 2160 and d["config"]["variant"] == "auto"
    want = d["uniform_normal_path_share"]
    got = emul.emulate(3840, 2160, steps=(8, 16))
    for S in (8, 16):
        assert abs(got[S]["A"] - want[str(S)]) <= 2e-4, (S, got[S]["A"], want[str(S)])
        assert got[S]["A"] <= got[S]["B"] <= got[S]["D"] and got[S]["A"] <= got[S]["C"] <= got[S]["D"]      # finer rules only ever add waves
    # a scene with a normal of its own in every texel never offers the fast path (also.curved_scene: share 0.0)
    assert d["also"]["curved_scene"]["uniform_normal_path_share"]["all"] == 0.0
    assert emul.emulate(1920, 1080, steps=(4,), scene="curved")[4]["D"] <= 0.01
