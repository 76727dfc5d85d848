"""Round 6 on the GPU: the tap-path statistics of the streaming a-trous kernel (svgf_path_stats_enable, include/svgf_ext.h) and stage parity on
the smooth-shaded scene (svgf_amd/synth.py, scene "curved": a normal of its own in every texel, the general tap path everywhere)."""
import numpy as np
import pytest

from svgf_amd import synth
from tests.helpers import CDT, gbuf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


def _run(G, d, fr, n, stats):
    outs = []
    gb = G.gb_dev(fr)
    rad = G.dev(fr["radiance"].astype(G.NPDT[d.params.storage]))
    if stats:
        d.path_stats_enable(True)
    for k in range(n):
        outs.append(G.host(d.Render(rad, gb, gb if k else None)).copy())
    return outs, (d.path_stats_read() if stats else None)


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_path_statistics_count_the_uniform_normal_path_and_change_nothing(G, storage):
    from svgf_amd import filter as F
    W, H, N = 640, 360, 5
    planar, curved = synth.make_frame(W, H, 0), synth.make_frame(W, H, 0, scene="curved")
    P = F.Params(storage=storage, steps=5)
    plain, _ = _run(G, F.Denoiser(W, H, P), planar, N, False)
    d = F.Denoiser(W, H, P)
    counted, st = _run(G, d, planar, N, True)
    for a, b in zip(plain, counted):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), "counting changed a result"
    nsurf = int(np.ceil((planar["region"] != synth.SKY).any(1).sum()))                 # rows that hold a surface pixel
    for i in range(5):
        total, uni = st[1 << i]
        assert 0 < uni <= total, (1 << i, total, uni)
        assert total <= N * H * ((W + 63) // 64) * 2 and total >= N * nsurf * (W // 128)   # a wave-step is 64 pixels of one row of a launch (bands overlap: <= 2x)
    assert st[1][1] / st[1][0] > 0.4 and st[1][1] / st[1][0] >= st[16][1] / st[16][0]     # piecewise planar: mostly uniform, less so at wide steps
    assert st[32] == (0, 0) and st[64] == (0, 0)
    assert d.path_stats_read() == {1 << i: (0, 0) for i in range(7)}                      # read zeroes the counters
    d.path_stats_enable(False)
    with pytest.raises(F.SvgfError, match="not enabled"):
        d.path_stats_read()
    # the general tap path only: the fast path switched off, and geometry that never offers it
    _, sg = _run(G, F.Denoiser(W, H, F.Params(storage=storage, steps=5, variant="lds-general")), planar, 2, True)
    _, sc = _run(G, F.Denoiser(W, H, P), curved, 2, True)
    for i in range(5):
        assert sg[1 << i][0] > 0 and sg[1 << i][1] == 0
        assert sc[1 << i][0] > 0 and sc[1 << i][1] <= 0.02 * sc[1 << i][0]


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_stages_on_the_smooth_shaded_scene_match_the_oracle(G, oracle, storage):
    """Temporal (bit for bit), moments and a-trous steps 1 / 4 / 16 (stage tolerances) on scene "curved", default variant = the LDS kernels' general
    tap path on every wave; and the frame driver on it equals the stage calls' sequence through the oracle's frame sequencing within the
    free-running bound."""
    from svgf_amd import filter as F
    W, H = 333, 207
    rng = np.random.default_rng(61)
    dt = CDT[storage]
    mv = (1.0, -2.0)
    f0, f1 = synth.make_frame(W, H, 3, mv=mv, scene="curved"), synth.make_frame(W, H, 4, mv=mv, scene="curved")
    d = F.Denoiser(W, H, F.Params(storage=storage))
    prev = rng.uniform(-0.1, 1.2, (H, W, 4)).astype(dt)
    mom_prev = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hist_prev = rng.integers(0, 40, (H, W)).astype(np.uint8)
    cur = (f1["radiance"] * 1.3 - 0.1).astype(dt)
    out = np.zeros_like(cur); hist = np.zeros((H, W), np.uint8); mom = np.zeros((H, W, 2), dt)
    oracle.temporal(W, H, storage, prev, cur, out, gbuf(f1), gbuf(f0), hist_prev, hist, mom, mom_prev, depth_threshold=0.8, normal_threshold=0.9,
                    history_base=24, mesh_id_test=1)
    o_col, o_hist, o_mom = d.new_colour(), d.new_history(), d.new_moments()
    d.TemporalFilter(G.dev(prev), G.dev(cur), o_col, G.gb_dev(f1), G.gb_dev(f0), G.dev(hist_prev), o_hist, o_mom, G.dev(mom_prev))
    assert np.array_equal(G.host(o_hist), hist) and 0.2 < (hist > 1).mean() < 0.98       # the curved surfaces do reproject (and some pixels do not)
    assert np.array_equal(G.host(o_col).view(np.uint8), out.view(np.uint8)) and np.array_equal(G.host(o_mom).view(np.uint8), mom.view(np.uint8))
    col = rng.uniform(0, 1, (H, W, 4)).astype(dt)
    momp = rng.uniform(0, 1, (H, W, 2)).astype(dt)
    hl = rng.integers(1, 8, (H, W)).astype(np.uint8)
    want = np.zeros_like(col)
    oracle.moments(W, H, storage, col, want, momp, gbuf(f1), hl, phi_colour=10.0, phi_normal=128.0, radius=3)
    o = d.new_colour()
    d.FilterMoments(G.dev(col), o, G.dev(momp), G.gb_dev(f1), G.dev(hl))
    got = G.host(o)
    assert np.array_equal(got[hl >= 4].view(np.uint8), col[hl >= 4].view(np.uint8))
    lim = 2e-5 if storage == "f32" else 1e-3
    assert np.abs(got[..., :3].astype(np.float64) - want[..., :3].astype(np.float64)).max() <= lim
    for step in (1, 4, 16):
        src = np.concatenate([f1["radiance"][..., :3] * 1.2 - 0.05, rng.uniform(-0.01, 0.05, (H, W, 1)).astype(np.float32)], -1).astype(dt)
        want = np.zeros_like(src); fbw = np.full_like(src, 7)
        oracle.atrous(W, H, storage, src, want, fbw, gbuf(f1), step=step, phi_colour=10.0, phi_normal=128.0, iteration=0)
        o, fb = d.new_colour(), G.dev(np.full_like(src, 7))
        d.FilterKernel(G.dev(src), o, fb, G.gb_dev(f1), step, 0)
        G.assert_colour_close(G.host(o), want, storage, f"curved scene, a-trous step {step}")
        G.assert_colour_close(G.host(fb), fbw, storage, f"curved scene, feedback step {step}")
        sky = f1["region"] == synth.SKY
        assert np.array_equal(G.host(o)[sky].view(np.uint8), want[sky].view(np.uint8))


@pytest.mark.parametrize("edge_first", [False, True])
def test_the_bench_check_of_the_strips_against_the_one_gpu_frame(G, edge_first):
    """What bench_strips does before it trusts a schedule (svgf_amd/strips.py: verify, _checksum), here with every rank of the partition in this
    process (mailbox transport: real peer addressing, real exchanges): VERIFY_FRAMES frames from a fresh start through four strips and through the
    whole-frame driver, every rank's owned rows against the same rows of the whole frame as 2 x 64-bit checksums — equal under both schedules;
    and a strip whose radiance differs in ONE texel of ONE frame is seen."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, storage = 640, 600, 4, "f32"
    P = F.Params(storage=storage, steps=5)
    fr = [synth.make_frame(W, H, k) for k in range(2)]
    gbw = [G.gb_dev(fr[0]), G.gb_dev(fr[0])]
    whole = F.Denoiser(W, H, P)
    for k in range(strips.VERIFY_FRAMES):
        out_w = whole.Render(G.dev(fr[k & 1]["radiance"]), gbw[k & 1], gbw[(k & 1) ^ 1] if k else None)
    torch.cuda.synchronize()
    parts = strips.partition(H, world)
    ref = torch.stack([strips._checksum(out_w[a:b]) for a, b in parts]).cpu()

    def run(spoil):
        drv = strips.NativeStrips(W, H, world, P, list(range(world)), [0] * world, plan="auto", motion_reach=0, transport="mailbox")
        drv.set_edge_first(edge_first)
        assert drv.plan == "grouped"
        try:
            gbs = [[F.GBuffer(*(G.dev(np.ascontiguousarray(fr[0][n][lay["y0"]:lay["y1"]])) for n in ("motion", "normal", "uv"))) for lay in drv.layouts] for _ in range(2)]
            for k in range(strips.VERIFY_FRAMES):
                rads = [np.ascontiguousarray(fr[k & 1]["radiance"][lay["y0"]:lay["y1"]]).copy() for lay in drv.layouts]
                if spoil and k == 2:
                    lay = drv.layouts[2]
                    rads[2][lay["own"][0] - lay["y0"] + 7, 100, 1] += np.float32(2.0 ** -20)       # one texel of rank 2's own rows, one frame
                outs = drv.frame([G.dev(r) for r in rads], gbs[k & 1], gbs[(k & 1) ^ 1] if k else None)
            drv.sync()
            return torch.stack([strips._checksum(drv.owned(r, o)) for r, o in enumerate(outs)]).cpu()
        finally:
            drv.close()
    assert torch.equal(run(False), ref), f"strips (edge_first={edge_first}) differ from the whole frame"
    bad = run(True)
    assert not torch.equal(bad[2], ref[2]), "a spoiled texel must show in its rank's checksum"
