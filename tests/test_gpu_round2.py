"""GPU tests of the pieces added after round 1: the C++ strip driver with RCCL groups (loop-back communicator), BASELINE config #4
(7680x4320 as 8 strips of 540 rows), the 1080p fp32 full-size run, the temporal strip guard turned into a reported error,
device handling of the ABI, svgf_resize, the debug-view sequences, and a per-stage error report."""
import json
import os

import numpy as np
import pytest

from svgf_amd import synth
from tests.conftest import ROOT
from tests.helpers import CDT, frames, free_running_envelope, gbuf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


@pytest.fixture(scope="module")
def loop_comm():
    """ONE RCCL communicator of world size 1: every send/recv of the virtual ranks has communicator rank 0 as its peer."""
    from svgf_amd import filter as F
    from svgf_amd import strips
    comm = strips.rccl_comm(1, 0, 0)
    yield comm
    F.load_library().svgf_rccl_comm_destroy(comm)


def _strip_inputs(G, fr, lay, storage):
    from svgf_amd import filter as F
    sl = slice(lay["y0"], lay["y1"])
    gb = F.GBuffer(*(G.dev(np.ascontiguousarray(fr[k][sl])) for k in ("motion", "normal", "uv")))
    rad = G.dev(np.ascontiguousarray(fr["radiance"][sl].astype(G.NPDT[storage])))
    return rad, gb


@pytest.mark.parametrize("plan", ["ghost", "grouped", "per-iteration"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_native_strip_driver_over_rccl_loopback(G, loop_comm, plan, storage):
    """svgf_strips_frame (C++: stage sequence, ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd on its own stream, HIP events, state
    exchange posted after iteration 0) with 3 virtual ranks on one device: every frame of a panning sequence equals the
    single-context result bit for bit, history included; no reprojection leaves a strip."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, N = 320, 420, 3, 4
    fr = frames(W, H, N, mv=(1.0, -2.5))
    params = F.Params(storage=storage, steps=5)
    whole = G.HipPipeline(W, H, storage, steps=5)
    side = torch.cuda.Stream(priority=-1)
    drv = strips.NativeStrips(W, H, world, params, list(range(world)), [0] * world, streams=[side.cuda_stream] * world, comms=[loop_comm],
                              plan=plan, motion_reach=3, loopback=True)
    assert drv.plan == plan
    gbs = [G.gb_dev(f) for f in fr]
    torch.cuda.synchronize()
    prev_in = None
    for k in range(N):
        want = whole.frame(fr[k]["radiance"], gbs[k], gbs[max(k - 1, 0)])
        torch.cuda.synchronize()
        cur_in = [_strip_inputs(G, fr[k], lay, storage) for lay in drv.layouts]
        outs = drv.frame([c[0] for c in cur_in], [c[1] for c in cur_in], [p[1] for p in prev_in] if prev_in else None)
        drv.sync()
        got = np.concatenate([G.host(drv.owned(r, o)) for r, o in enumerate(outs)], 0)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"plan {plan}: frame {k}"
        prev_in = cur_in
    hist = np.concatenate([G.host(drv.owned(r, drv.state_plane(r, F.PLANE_HISTORY, 1 - drv.pingpong(r)))) for r in range(world)], 0)
    assert np.array_equal(hist, whole.taps["hist"])
    drv.close()


def test_native_strip_driver_on_crowded_frames(G, loop_comm):
    """Every 8th column disoccluded in every frame from frame 3 on (12 % of the surface pixels young): each rank's sample says "crowded" two
    frames later and the rank serves its strip's young pixels with the LDS-streaming kernel (choose_moments_kernel: every rank for itself).
    Every frame still equals the single-context stage sequence bit for bit."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, N, storage = 512, 420, 3, 10, "f32"
    fr = frames(W, H, 2, mv=(0.0, 0.0))
    flipped = fr[0]["normal"].copy()
    flipped.view(np.int16)[:, ::8, 0:3] ^= np.int16(-32768)
    variants = [dict(fr[0]), dict(fr[0], normal=flipped)]                 # the two G-buffers the crowded frames alternate between
    frame_of = lambda k: dict(variants[k % 2] if k >= 3 else variants[0], radiance=fr[k % 2]["radiance"])      # noqa: E731
    params = F.Params(storage=storage, steps=5)
    whole = G.HipPipeline(W, H, storage, steps=5)
    side = torch.cuda.Stream(priority=-1)
    drv = strips.NativeStrips(W, H, world, params, list(range(world)), [0] * world, streams=[side.cuda_stream] * world, comms=[loop_comm],
                              plan="ghost", motion_reach=3, loopback=True)
    torch.cuda.synchronize()
    prev_in, prev_gb, crowded = None, None, []
    for k in range(N):
        f = frame_of(k)
        gb = G.gb_dev(f)
        want = whole.frame(f["radiance"], gb, prev_gb if prev_gb is not None else gb)
        torch.cuda.synchronize()
        cur_in = [_strip_inputs(G, f, lay, storage) for lay in drv.layouts]
        outs = drv.frame([c[0] for c in cur_in], [c[1] for c in cur_in], [p[1] for p in prev_in] if prev_in else None)
        drv.sync()
        got = np.concatenate([G.host(drv.owned(r, o)) for r, o in enumerate(outs)], 0)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"frame {k}"
        crowded.append([bool(drv.lib.svgf_adaptive_moments_state(drv.lib.svgf_strips_context(drv._h, r))) for r in range(world)])
        prev_in, prev_gb = cur_in, gb
    assert not any(crowded[3]) and all(crowded[-1]), crowded
    drv.close()


@pytest.mark.parametrize("plan", ["per-iteration", "grouped", "ghost"])
def test_native_strip_driver_with_two_frames_in_flight(G, loop_comm, plan):
    """svgf_strips_set_frames_in_flight(2): iterations 1.. of a frame — their halo exchanges included — on a side stream of every
    rank beside the next frame's temporal launch, frames alternating between two pairs of filter planes.  Six frames of a panning
    sequence WITHOUT a synchronisation in between (result f is read after call f + 1, or after the final sync): bitwise equal to
    the single-context frames, history included; then back to one frame in flight, and one frame more."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, N = 320, 420, 3, 7
    storage = "f32"
    fr = frames(W, H, N, mv=(1.0, -2.5))
    params = F.Params(storage=storage, steps=5)
    whole = G.HipPipeline(W, H, storage, steps=5)
    side = torch.cuda.Stream(priority=-1)
    drv = strips.NativeStrips(W, H, world, params, list(range(world)), [0] * world, streams=[side.cuda_stream] * world, comms=[loop_comm],
                              plan=plan, motion_reach=3, loopback=True)
    drv.set_frames_in_flight(2)
    gbs = [G.gb_dev(f) for f in fr]
    want = [whole.frame(fr[k]["radiance"], gbs[k], gbs[max(k - 1, 0)]) for k in range(N)]
    torch.cuda.synchronize()
    inputs = [[_strip_inputs(G, fr[k], lay, storage) for lay in drv.layouts] for k in range(N)]       # every frame's planes stay alive and untouched
    outs, got = {}, {}

    def collect(k):          # on the ranks' compute stream: ordered behind frame k by call k + 1
        with torch.cuda.stream(side):
            got[k] = [drv.owned(r, o).clone() for r, o in enumerate(outs[k])]
    for k in range(N - 1):
        outs[k] = drv.frame([c[0] for c in inputs[k]], [c[1] for c in inputs[k]], [p[1] for p in inputs[k - 1]] if k else None)
        if k >= 1:
            collect(k - 1)
    drv.sync()
    collect(N - 2)
    drv.set_frames_in_flight(1)
    k = N - 1
    outs[k] = drv.frame([c[0] for c in inputs[k]], [c[1] for c in inputs[k]], [p[1] for p in inputs[k - 1]])
    drv.sync()
    collect(k)
    torch.cuda.synchronize()
    for k in range(N):
        g = np.concatenate([G.host(t) for t in got[k]], 0)
        assert np.array_equal(g.view(np.uint8), want[k].view(np.uint8)), f"plan {plan}: frame {k}"
    hist = np.concatenate([G.host(drv.owned(r, drv.state_plane(r, F.PLANE_HISTORY, 1 - drv.pingpong(r)))) for r in range(world)], 0)
    assert np.array_equal(hist, whole.taps["hist"])
    drv.close()


def test_config4_8k_as_eight_strips_of_540_rows(G, loop_comm):
    """BASELINE.json configs[3]: 7680x4320 fp32 cut into 8 strips x 540 rows (plan auto = grouped: one exchange of filter rows between iterations 2 and 3,
    48-row halo), two frames through the C++ strip driver with its RCCL exchanges, bitwise against the whole frame on the same device."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, storage = 7680, 4320, 8, "f32"
    sc = synth.make_scene(W, H, 0)
    rads = [synth.make_radiance(sc["base"], W, k) for k in range(2)]
    params = F.Params(storage=storage, steps=5)
    drv = strips.NativeStrips(W, H, world, params, list(range(world)), [0] * world, comms=[loop_comm], plan="auto", motion_reach=4, loopback=True)
    assert drv.plan == "grouped" and [lay["own"][1] - lay["own"][0] for lay in drv.layouts] == [540] * 8
    assert drv.layouts[3]["y0"] == 1620 - 48 and drv.layouts[3]["y1"] == 2160 + 48
    whole = G.HipPipeline(W, H, storage, steps=5)
    gb = G.gb_dev(sc)
    strip_gb = [F.GBuffer(*(gb_t[lay["y0"]:lay["y1"]].contiguous() for gb_t in (gb.motion, gb.normal, gb.uv))) for lay in drv.layouts]
    for k in range(2):
        want = torch.from_numpy(whole.frame(rads[k], gb, gb))
        rad = G.dev(rads[k])
        outs = drv.frame([rad[lay["y0"]:lay["y1"]].contiguous() for lay in drv.layouts], strip_gb, strip_gb if k else None)
        drv.sync()
        got = torch.cat([drv.owned(r, o) for r, o in enumerate(outs)], 0).cpu()
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), f"8K / 8 strips: frame {k}"
    drv.close()


def test_1080p_fp32_full_size(G, loop_comm):
    """BASELINE.json configs[1] at full size: 1920x1080 fp32, temporal + 5 iterations.  Size-independent properties: history
    counts 1..HistoryLength on surfaces and 1 on sky; the filtered colour stays inside [0,1] (convex weights of clamped
    inputs); sky texels filter to exactly 0; two strips tile the whole frame bitwise."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, base = 1920, 1080, 6
    fr = synth.make_frame(W, H, 0)
    d = F.Denoiser(W, H, F.Params(storage="f32", steps=5, history_base=base))
    gb = G.gb_dev(fr)
    rad = G.dev(fr["radiance"])
    sky = torch.from_numpy(fr["region"] == synth.SKY).cuda()
    outs = []
    for k in range(8):
        out = d.Render(rad, gb, gb)
        h = d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())
        assert torch.all(h[~sky] == min(k + 1, base)) and torch.all(h[sky] == 1)
        assert torch.isfinite(out).all() and out[..., :3].min() >= 0 and out[..., :3].max() <= 1
        outs.append(out.clone())
    # sky texels never accumulate history (zero normal), stay on the spatial estimate, whose weights are all pow(0, phi_n) = 0:
    # exactly 0, copied through every iteration ("faithful but ugly")
    assert torch.all(outs[-1][sky] == 0)
    # two strips == the whole frame (stage calls), two frames
    whole = G.HipPipeline(W, H, "f32", steps=5)
    drv = strips.NativeStrips(W, H, 2, F.Params(storage="f32", steps=5), [0, 1], [0, 0], comms=[loop_comm], plan="auto", motion_reach=0, loopback=True)
    sgb = [F.GBuffer(*(t[lay["y0"]:lay["y1"]].contiguous() for t in (gb.motion, gb.normal, gb.uv))) for lay in drv.layouts]
    for k in range(2):
        r_np = synth.make_radiance(fr["base"], W, k)
        want = torch.from_numpy(whole.frame(r_np, gb, gb))
        r_dev = G.dev(r_np)
        o = drv.frame([r_dev[lay["y0"]:lay["y1"]].contiguous() for lay in drv.layouts], sgb, sgb if k else None)
        drv.sync()
        got = torch.cat([drv.owned(r, t) for r, t in enumerate(o)], 0).cpu()
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), k
    drv.close()


def test_motion_beyond_the_state_halo_is_reported(G, loop_comm):
    """A strip's temporal stage never reads outside its rows; a reprojection that would (|mv.y| = 6 rows with motion_reach = 4)
    used to turn silently into a rejection.  Now it is counted on the device and svgf_strips_sync / svgf_sync return
    SVGF_ERR_HALO; with a sufficient reach the same sequence is bit-identical and raises nothing."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, N, storage = 256, 420, 3, 3, "f32"
    fr = frames(W, H, N, mv=(0.0, 6.0))
    params = F.Params(storage=storage, steps=5)
    whole = G.HipPipeline(W, H, storage, steps=5)
    gbs = [G.gb_dev(f) for f in fr]
    wants = [whole.frame(fr[k]["radiance"], gbs[k], gbs[max(k - 1, 0)]) for k in range(N)]
    for reach, ok in ((4, False), (6, True)):
        drv = strips.NativeStrips(W, H, world, params, list(range(world)), [0] * world, comms=[loop_comm], plan="grouped", motion_reach=reach, loopback=True)
        prev_in, raised = None, False
        for k in range(N):
            cur_in = [_strip_inputs(G, fr[k], lay, storage) for lay in drv.layouts]
            outs = drv.frame([c[0] for c in cur_in], [c[1] for c in cur_in], [p[1] for p in prev_in] if prev_in else None)
            try:
                drv.sync()
            except F.SvgfError as e:
                assert "halo" in str(e) and "reprojection" in str(e)
                raised = True
            got = np.concatenate([G.host(drv.owned(r, o)) for r, o in enumerate(outs)], 0)
            if ok:
                assert np.array_equal(got.view(np.uint8), wants[k].view(np.uint8)), k
            prev_in = cur_in
        assert raised == (not ok), f"motion_reach {reach}"
        drv.close()
    # the same guard through the stage calls of a strip context
    lay = strips.strips_plan(W, H, 1, world, 5, "grouped", 3, 4)
    d = F.Denoiser(W, H, params, strip=(lay["y0"], lay["y1"] - lay["y0"], lay["own"][0], lay["own"][1]))
    rad, gb1 = _strip_inputs(G, fr[1], lay, storage)
    _, gb0 = _strip_inputs(G, fr[0], lay, storage)
    d.set_rows(lay["y0"], lay["y1"])                     # temporal on every local row: the outermost ones reproject 6 rows out
    d.TemporalFilter(d.new_colour(), rad, d.new_colour(), gb1, gb0, d.new_history(), d.new_history(), d.new_moments(), d.new_moments())
    assert d.halo_violations() > 0
    with pytest.raises(F.SvgfError, match="halo"):
        d.sync()
    d.sync()                                             # the counter was cleared by the failing call
    wd = F.Denoiser(W, H, params)                        # a whole-frame context holds every row: nothing to count
    wd.Render(G.dev(fr[1]["radiance"]), gbs[1], gbs[0])
    wd.sync()
    assert wd.halo_violations() == 0


def test_contexts_name_their_device(G):
    """Every entry point runs on the context's device and leaves the caller's current device alone; a device that does not exist
    is refused cleanly (on a 1-GPU box: device 1)."""
    import ctypes as C
    import torch
    from svgf_amd import filter as F
    n = torch.cuda.device_count()
    lib = F.load_library()
    h = C.c_void_p()
    p = F.Params(storage="f32").to_c()
    assert lib.svgf_create(C.byref(h), 64, 64, C.byref(p), n, None) == -3 and not h.value          # SVGF_ERR_NO_DEVICE
    assert lib.svgf_create(C.byref(h), 64, 64, C.byref(p), -1, None) == -3
    # two contexts side by side (same device on a 1-GPU box, two devices otherwise), interleaved frames, each equal to a lone run
    W, H = 200, 90
    fr = frames(W, H, 3, mv=(1.0, 0.0))
    devs = [0, 1 % n]
    ds = [F.Denoiser(W, H, F.Params(storage="f32", steps=3), device=dv) for dv in devs]
    lone = F.Denoiser(W, H, F.Params(storage="f32", steps=3), device=0)
    before = torch.cuda.current_device()
    for k in range(3):
        want = G.host(lone.Render(G.dev(fr[k]["radiance"]), G.gb_dev(fr[k]), G.gb_dev(fr[k - 1]) if k else None))
        for d, dv in zip(ds, devs):
            dev = f"cuda:{dv}"
            got = G.host(d.Render(G.dev(fr[k]["radiance"], dev), G.gb_dev(fr[k], dev), G.gb_dev(fr[k - 1], dev) if k else None))
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (k, dv)
        assert torch.cuda.current_device() == before
    # plane sizes the kernels' 32-bit offsets cannot address are refused at creation
    assert lib.svgf_create(C.byref(h), 16384, 16384, C.byref(p), 0, None) == -1


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_resize_reallocates_and_restarts(G, storage):
    """svgf_resize = application::ResizeRenderTextures (App.cu:742-778): after it the context behaves like a new one of the new
    size (state zeroed, ping-pong restarted), tunables kept."""
    from svgf_amd import filter as F
    d = F.Denoiser(200, 120, F.Params(storage=storage, steps=4, phi_colour=7.0))
    fr = frames(200, 120, 3, mv=(1.0, 0.0))
    for k in range(3):
        d.Render(G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), G.gb_dev(fr[k]), G.gb_dev(fr[k - 1]) if k else None)
    for (W, H) in ((331, 203), (64, 40)):
        d.Resize(W, H)
        assert d.size() == (W, H, (0, H, 0, H)) and d.pingpong() == 0 and d.state_plane(F.PLANE_COLOUR, 0) is None
        fresh = F.Denoiser(W, H, F.Params(storage=storage, steps=4, phi_colour=7.0))
        fr = frames(W, H, 4, mv=(-2.5, 1.5))
        for k in range(4):
            args = (G.dev(fr[k]["radiance"].astype(G.NPDT[storage])), G.gb_dev(fr[k]), G.gb_dev(fr[k - 1]) if k else None)
            a, b = G.host(d.Render(*args)), G.host(fresh.Render(*args))
            assert a.shape == (H, W, 4) and np.array_equal(a.view(np.uint8), b.view(np.uint8)), (W, H, k)
    with pytest.raises(F.SvgfError):
        d.Resize(0, 10)


@pytest.mark.parametrize("mesh_id_test", [1, 0])
def test_previous_guide_plane_stands_in_for_the_previous_gbuffer(G, mesh_id_test):
    """svgf_set_prev_guide (svgf.h): the frame driver keeps {depth, ddepth, normal, instance ID} of every current G-buffer and, when
    the next frame's `prev` is that G-buffer (same plane addresses), its reprojection test reads the kept plane instead of the three
    planes.  (a) bit-identical to the driver with the feature off and to a `prev` handed over as copies at other addresses, on a
    panning sequence with disocclusions and instance-ID rejections, history included; (b) the planes of `prev` are really not read:
    scrambling them after their frame changes nothing with the feature on, and does with it off."""
    import torch
    from svgf_amd import filter as F
    W, H, N = 333, 210, 6
    fr = frames(W, H, N, mv=(-2.5, 1.5))
    p = F.Params(storage="f32", steps=5, mesh_id_test=mesh_id_test)
    on, off, copies, scr_on, scr_off = (F.Denoiser(W, H, p) for _ in range(5))
    on.set_prev_guide(True)                                 # opt-in: the host vouches that it leaves the previous planes alone
    scr_on.set_prev_guide(True)
    copies.set_prev_guide(True)                             # ... and a `prev` at other addresses is read as it is anyway
    gbs = [G.gb_dev(f) for f in fr]
    gbs_scr = [G.gb_dev(f) for f in fr]                     # a second set, scrambled frame by frame
    differs = False
    for k in range(N):
        rad = G.dev(fr[k]["radiance"])
        prev = gbs[k - 1] if k else None
        a = G.host(on.Render(rad, gbs[k], prev))
        b = G.host(off.Render(rad, gbs[k], prev))
        c = G.host(copies.Render(rad, gbs[k], F.GBuffer(prev.motion.clone(), prev.normal.clone(), prev.uv.clone()) if k else None))
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)) and np.array_equal(a.view(np.uint8), c.view(np.uint8)), k
        for plane in (F.PLANE_HISTORY, F.PLANE_MOMENTS, F.PLANE_COLOUR):
            ha, hb = (G.host(d.state_plane(plane, 1 - d.pingpong())) for d in (on, off))
            assert np.array_equal(ha.view(np.uint8), hb.view(np.uint8)), (k, plane)
        if k:
            torch.cuda.synchronize()
            for t in (gbs_scr[k - 1].motion, gbs_scr[k - 1].normal, gbs_scr[k - 1].uv):
                t.zero_()                                    # the previous G-buffer is gone: depth 0 everywhere = sky, nothing reprojects
        s_on = G.host(scr_on.Render(rad, gbs_scr[k], gbs_scr[k - 1] if k else None))
        s_off = G.host(scr_off.Render(rad, gbs_scr[k], gbs_scr[k - 1] if k else None))
        assert np.array_equal(s_on.view(np.uint8), a.view(np.uint8)), f"frame {k}: the previous planes were read"
        differs = differs or not np.array_equal(s_off.view(np.uint8), a.view(np.uint8))
    assert differs, "scrambling the previous G-buffer must matter when its planes are read"
    # fp16 storage keeps a guide plane too (same bits with the switch on and off); without any iteration, and with the direct
    # variant, there is none: the switch is accepted and changes nothing
    for kw in (dict(storage="f16", steps=5), dict(storage="f32", steps=0), dict(storage="f32", steps=3, variant="direct")):
        x, y = F.Denoiser(W, H, F.Params(**kw)), F.Denoiser(W, H, F.Params(**kw))
        x.set_prev_guide(True)
        for k in range(3):
            rad = G.dev(fr[k]["radiance"].astype(G.NPDT[kw["storage"]]))
            ax, ay = (G.host(d.Render(rad, gbs[k], gbs[k - 1] if k else None)) for d in (x, y))
            assert np.array_equal(ax.view(np.uint8), ay.view(np.uint8))


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_debug_view_sequences(G, oracle, storage):
    """SVGFDebugOutput::TemporalFilter (App.cu:602-609): temporal only; ::ATrousWaveletFilter (App.cu:611-620): temporal, then the
    wavelet filter on what FilterBuffer[0] still holds — the previous frame's result — with iteration 0 feeding RenderBuffer
    back.  Against the oracle's stage functions sequenced the same way."""
    from svgf_amd import filter as F
    W, H, N, steps = 160, 96, 4, 3
    fr = frames(W, H, N, mv=(1.0, 0.0))
    dt = CDT[storage]
    P = dict(depth_threshold=0.8, normal_threshold=0.9, history_base=24, mesh_id_test=1)
    for mode in ("temporal", "atrous"):
        d = F.Denoiser(W, H, F.Params(storage=storage, steps=steps))
        d.set_debug_mode(mode)
        colour = [np.zeros((H, W, 4), dt) for _ in range(2)]
        mom = [np.zeros((H, W, 2), dt) for _ in range(2)]
        hist = [np.zeros((H, W), np.uint8) for _ in range(2)]
        filt = [np.zeros((H, W, 4), dt) for _ in range(2)]
        pp_res, Pi = 0, 0
        for k in range(N):
            kp = max(k - 1, 0)
            got = G.host(d.Render(G.dev(fr[k]["radiance"].astype(dt)), G.gb_dev(fr[k]), G.gb_dev(fr[kp]) if k else None))
            oracle.temporal(W, H, storage, colour[1 - Pi], fr[k]["radiance"].astype(dt), colour[Pi], gbuf(fr[k]), gbuf(fr[kp]), hist[1 - Pi], hist[Pi],
                            mom[Pi], mom[1 - Pi], **P)
            if mode == "temporal":
                want = colour[Pi]
                assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (mode, k)
            else:
                pp = pp_res
                for i in range(steps):
                    oracle.atrous(W, H, storage, filt[pp], filt[1 - pp], colour[Pi] if i == 0 else None, gbuf(fr[k]), step=1 << i, phi_colour=10.0,
                                  phi_normal=128.0, iteration=i)
                    pp ^= 1
                pp_res = pp
                if storage == "f32":
                    G.assert_colour_close(got, filt[pp], storage, f"debug atrous frame {k}")
                else:
                    assert np.abs(got.astype(np.float64) - filt[pp].astype(np.float64)).max() <= 2e-3
            assert np.array_equal(G.host(d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())), hist[Pi]), (mode, k)
            Pi ^= 1
    with pytest.raises(F.SvgfError):
        d._check(d.lib.svgf_set_debug_mode(d._h, 7), "svgf_set_debug_mode")


def test_parity_report(G, oracle):
    """Max / mean error of every stage against the oracle on identical inputs, and of the free-running sequence, written to
    gpurun_out/parity_report.json (copied into profiles/ per round) so that drift is visible from round to round.  Bounds
    as in tests/gpu_helpers.py:TOL; accept/reject masks must match exactly (mismatch counts are reported and must be 0)."""
    from svgf_amd import filter as F
    W, H, N = 256, 144, 8
    report = {"frame": f"{W}x{H}", "frames": N, "mv": [-2.5, 1.5], "stages": {}}
    for storage in ("f32", "f16"):
        fr = frames(W, H, N, mv=(-2.5, 1.5))
        ref = oracle.Pipeline(W, H, storage, steps=5, nthreads=8)
        free = G.HipPipeline(W, H, storage, steps=5)
        d = F.Denoiser(W, H, F.Params(storage=storage, steps=5))
        gbs = [G.gb_dev(f) for f in fr]
        acc = {}

        def note(name, got, want):
            e = np.abs(got.astype(np.float64) - want.astype(np.float64))
            a = acc.setdefault(name, {"max_abs": 0.0, "sum": 0.0, "n": 0})
            a["max_abs"] = max(a["max_abs"], float(e.max())); a["sum"] += float(e.sum()); a["n"] += e.size
        mask_mismatch = 0
        for k in range(N):
            kp = max(k - 1, 0)
            want_out = ref.frame(fr[k]["radiance"], gbuf(fr[k]), gbuf(fr[kp]))
            t = ref.taps
            col, hist, mom = d.new_colour(), d.new_history(), d.new_moments()
            d.TemporalFilter(G.dev(t["prev_colour"]), G.dev(t["radiance"]), col, gbs[k], gbs[kp], G.dev(t["prev_hist"]), hist, mom, G.dev(t["prev_mom"]))
            mask_mismatch += int((G.host(hist) != t["hist"]).sum())
            note("temporal_colour", G.host(col), t["temporal"]); note("temporal_moments", G.host(mom), t["mom"])
            out = d.new_colour()
            d.FilterMoments(G.dev(t["temporal"]), out, G.dev(t["mom"]), gbs[k], G.dev(t["hist"]))
            note("moments", G.host(out), t["moments"])
            for i in range(5):
                d.FilterKernel(G.dev(t["atrous_in"][i]), out, G.dev(t["temporal"]) if i == 0 else None, gbs[k], 1 << i, i)
                g_, w_ = G.host(out), t["atrous_out"][i]
                note(f"atrous_step{1 << i}_colour", g_[..., :3], w_[..., :3]); note(f"atrous_step{1 << i}_variance", g_[..., 3], w_[..., 3])
            got_free = free.frame(fr[k]["radiance"], gbs[k], gbs[kp])
            mask_mismatch += int((free.taps["hist"] != t["hist"]).sum())
            note("free_running_colour", got_free[..., :3], want_out[..., :3])
        rep = {n: {"max_abs": a["max_abs"], "mean_abs": a["sum"] / a["n"]} for n, a in acc.items()}
        rep["mask_mismatches"] = mask_mismatch
        report["stages"][storage] = rep
        # The envelope: the same 8 frames free-running through two CORRECT CPU builds of the reference's source — the oracle and its
        # all-fp32 + FMA-contraction build — against the device's distance from the oracle, for both cameras.
        for mv in ((-2.5, 1.5), (0.0, 0.0)):
            frs = fr if mv[0] else frames(W, H, N, mv=mv)
            env = {fl: free_running_envelope(oracle, frs, storage, flavour=fl) for fl in ("fp32", "fp32fma", "fused", "hwulp")}
            refm = oracle.Pipeline(W, H, storage, steps=5, nthreads=8)
            hipm = G.HipPipeline(W, H, storage, steps=5)
            gbm = gbs if mv[0] else [G.gb_dev(f) for f in frs]
            worst, tight = 0.0, (2e-5 if storage == "f32" else 1e-3)
            worst_frac = 0.0
            for k in range(N):
                kp = max(k - 1, 0)
                w_ = refm.frame(frs[k]["radiance"], gbuf(frs[k]), gbuf(frs[kp])).astype(np.float64)
                g_ = hipm.frame(frs[k]["radiance"], gbm[k], gbm[kp]).astype(np.float64)
                e_ = np.abs(g_ - w_)[..., :3]
                worst = max(worst, float(e_.max()))
                worst_frac = max(worst_frac, float((e_ > tight + 1e-5 * np.abs(w_[..., :3])).mean()))
            report.setdefault("envelope", {})[f"{storage} mv={list(mv)}"] = {
                "hip_vs_oracle": {"max_abs": worst, "frac_beyond_tight": worst_frac},
                "oracle_vs_build": env, "hip_inside_fp32fma_envelope": bool(worst <= env["fp32fma"]["max_abs"]),
                "hip_inside_hwulp_envelope": bool(worst <= env["hwulp"]["max_abs"])}
        assert mask_mismatch == 0
        assert rep["temporal_colour"]["max_abs"] == 0.0 and rep["temporal_moments"]["max_abs"] == 0.0          # bit-exact stage
        lim = 2e-5 + 1e-5 if storage == "f32" else 1e-3
        for i in range(5):
            assert rep[f"atrous_step{1 << i}_colour"]["max_abs"] <= lim, (storage, i)
        assert rep["free_running_colour"]["max_abs"] <= (5e-4 if storage == "f32" else 5e-3)
    # (written before the guard below: a run that trips it still leaves its numbers to be looked at — and, if the change is meant, committed)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    # drift guard: no stage's max error may grow beyond 4x what the last committed report recorded (identical inputs, deterministic
    # kernels: the numbers only move when a kernel changes)
    import glob
    committed = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_parity_report.json")))
    assert committed, "profiles/r0N_parity_report.json is missing"
    base = json.load(open(committed[-1]))["stages"]
    for storage, rep in report["stages"].items():
        for name, v in rep.items():
            if isinstance(v, dict) and name in base.get(storage, {}):
                old = base[storage][name]["max_abs"]
                assert v["max_abs"] <= 4.0 * old + 1e-12, f"{storage} {name}: max error {v['max_abs']:.3e} vs {old:.3e} in {os.path.basename(committed[-1])}"
    print(json.dumps(report))


def test_bench_ends_loudly_when_the_communicator_cannot_come_up():
    """Two rank processes on one device cannot bring an RCCL communicator up (RCCL refuses two ranks per device:
    profiles/r05_probe_rccl_two_ranks_one_gpu.txt): the bench must FAIL — non-zero exit, no JSON line — on every rank; there is no other
    driver it could quietly measure instead (the Python restatement of the schedule left the GPU path in round 5)."""
    import subprocess
    import sys
    env = dict(os.environ, SVGF_BENCH_SHARE_DEVICES="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "1080p", "--no-extra", "--no-one-gpu", "--prime-ms", "0", "--prime-frames", "0"],
                       env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert "RCCL communicator is unavailable" in p.stderr, p.stderr[-2000:]


def test_bench_strips_line_on_one_gpu(G):
    """`bench.py --gpus 1 --strips`: the N > 1 code path (C++ strip driver, world size 1) prints every field of the N > 1 line."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):      # an in-process group of this pytest run may own that port
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--strips", "--steps", "3", "--warmup", "1", "--workload", "4k", "--prime-ms", "0", "--prime-frames", "0"],
                       env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["driver"].startswith("C++") and d["one_gpu_ms"] > 0 and d["speedup_vs_one_gpu"] > 0
    assert set(d["halo_plans"]) == {"ghost", "grouped", "per-iteration"} and d["pan"]["ms_per_step"] > 0 and d["roofline"]["frac"] > 0


def test_bench_strips_leg_that_never_returns_leaves_the_measured_legs(G):
    """A leg of the N > 1 bench that hangs (SVGF_BENCH_HANG_AT: the test's stand-in for an exchange that never completes) does not take the
    legs before it along: after --leg-timeout the line is printed with the headline plan and the legs measured so far, `incomplete` names
    the leg, exit code 5 (a hang is never reported as success: ADVICE r05).  A hang BEFORE the headline has nothing to print: exit code 4, no line."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--strips", "--steps", "3", "--warmup", "1", "--workload", "1080p", "--prime-ms", "0", "--prime-frames", "0",
           "--leg-timeout", "15"]
    p = subprocess.run(cmd, env=dict(env, SVGF_BENCH_HANG_AT="plan ghost"), capture_output=True, text=True, timeout=500)
    assert p.returncode == 5, (p.returncode, p.stderr[-2000:])
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert "plan ghost" in d["incomplete"] and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["config"]["halo_plan"] == "grouped"
    assert set(d["halo_plans"]) == {"grouped", "per-iteration"} and d["pan"] is None
    assert d["verified"]["three_launches"] is True and d["verified"]["edge_first"] is True      # the strips reproduced the one-GPU frame under both schedules before the leg hung
    p = subprocess.run(cmd, env=dict(env, SVGF_BENCH_HANG_AT="headline"), capture_output=True, text=True, timeout=500)
    assert p.returncode == 4 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert "did not finish" in p.stderr


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_general_tap_path_equals_the_uniform_normal_fast_path(G, storage):
    """SVGF_VARIANT_LDS_GENERAL switches the uniform-normal fast path of the a-trous kernel off (every wave evaluates n.n' per tap):
    the frames of a panning sequence are bit-identical to the default — the fast path is an exact shortcut, whatever the geometry."""
    from svgf_amd import filter as F
    W, H, N = 389, 222, 4
    fr = frames(W, H, N, mv=(1.0, -2.5))
    a, b = F.Denoiser(W, H, F.Params(storage=storage, steps=5)), F.Denoiser(W, H, F.Params(storage=storage, steps=5, variant="lds-general"))
    gbs = [G.gb_dev(f) for f in fr]
    for k in range(N):
        rad = G.dev(fr[k]["radiance"].astype(G.NPDT[storage]))
        x, y = (G.host(d.Render(rad, gbs[k], gbs[k - 1] if k else None)) for d in (a, b))
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (storage, k)
