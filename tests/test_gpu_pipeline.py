"""GPU tests of two frames in flight (svgf_set_frames_in_flight, include/svgf.h): the frame driver puts iterations 1.. of a frame on a
stream of its own, beside the next frame's temporal launch.  Nothing about the results may change: every frame of a sequence — and the
state the sequence leaves (history length, moments, the fed-back colour of application::WaveletFilter's iteration 0, App.cu:504-505) —
equals the one-frame-at-a-time driver's, bit for bit, when each result is read after the call that orders it on the stream."""
import numpy as np
import pytest

from tests.helpers import frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


def _bits(t):
    import torch
    return t.contiguous().view(torch.uint8)


def _run(G, seq, storage, in_flight, *, steps=5, variant="auto", fusion=False, prev_guide=False, own_stream=False, reset_at=None,
         steps_at=None, timing=False):
    """The sequence through Render; -> (list of results as host arrays, state planes, the Denoiser's stage timing or None).
    With two frames in flight a result is copied only after the NEXT Render (or the final flush) — the contract of svgf.h."""
    import contextlib
    import torch
    from svgf_amd import filter as F
    H, W = seq[0]["radiance"].shape[:2]
    stream = torch.cuda.Stream() if own_stream else None
    d = F.Denoiser(W, H, F.Params(storage=storage, steps=steps, variant=variant), stream=stream.cuda_stream if stream else None)
    d.set_iteration_fusion(fusion)
    d.set_prev_guide(prev_guide)
    d.set_frames_in_flight(in_flight)
    if timing:
        d.timing_enable(True)
    gbs = [G.gb_dev(f) for f in seq]
    rads = [G.dev(f["radiance"].astype(G.NPDT[storage])) for f in seq]
    torch.cuda.synchronize()
    on = (lambda: torch.cuda.stream(stream)) if stream else contextlib.nullcontext
    outs, waiting = [], None
    for k in range(len(seq)):
        if reset_at == k:
            if waiting is not None:                     # svgf_reset_history zeroes the filter planes too: the frame in flight is read first
                d.flush()
                with on():
                    outs.append(waiting.clone())
                waiting = None
            d.reset_history()
        if steps_at and k in steps_at:
            d.set_params(F.Params(storage=storage, steps=steps_at[k], variant=variant))
        view = d.Render(rads[k], gbs[k], gbs[k - 1] if k else None)
        with on():
            if in_flight == 2:
                if waiting is not None:
                    outs.append(waiting.clone())        # frame k - 1: ordered on the stream by the call for frame k
                waiting = view
            else:
                outs.append(view.clone())
    if waiting is not None:
        d.flush()
        with on():
            outs.append(waiting.clone())
    d.sync()
    torch.cuda.synchronize()
    state = {"hist": G.host(d.state_plane(F.PLANE_HISTORY, 1 - d.pingpong())), "moments": G.host(d.state_plane(F.PLANE_MOMENTS, 1 - d.pingpong())),
             "colour": G.host(d.state_plane(F.PLANE_COLOUR, 1 - d.pingpong()))}
    t = d.timing_read() if timing else None
    return [G.host(o) for o in outs], state, t


def _assert_same(a, b):
    (oa, sa, _), (ob, sb, _) = a, b
    assert len(oa) == len(ob)
    for k, (x, y) in enumerate(zip(oa, ob)):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"frame {k}"
    for name in sa:
        assert np.array_equal(sa[name].view(np.uint8), sb[name].view(np.uint8)), name


@pytest.mark.parametrize("own_stream", [False, True])
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("mv", [(0.0, 0.0), (1.5, -2.5)])
def test_two_frames_in_flight_equal_one(G, storage, mv, own_stream):
    seq = frames(640, 360, 7, mv=mv)
    _assert_same(_run(G, seq, storage, 2, own_stream=own_stream), _run(G, seq, storage, 1, own_stream=own_stream))


@pytest.mark.parametrize("kw", [dict(fusion=True), dict(prev_guide=True), dict(variant="direct"), dict(variant="lds-general"), dict(steps=2),
                                dict(steps=1), dict(steps=0), dict(steps=10)], ids=lambda kw: "-".join(f"{k}={v}" for k, v in kw.items()))
def test_two_frames_in_flight_with_every_option(G, kw):
    """... with the pair launch (the tail then starts at iteration 2), the previous-guide read, the kernels that read the caller's
    G-buffer instead of the guide plane, and iteration counts that leave one / no launch for the side stream."""
    seq = frames(384, 216, 6, mv=(0.5, 1.0))
    _assert_same(_run(G, seq, "f32", 2, **kw), _run(G, seq, "f32", 1, **kw))


def test_reset_and_parameter_changes_between_frames_in_flight(G):
    """svgf_reset_history and svgf_set_params (another iteration count) while a frame's tail is still on the side stream."""
    seq = frames(512, 288, 8, mv=(1.0, 0.0))
    kw = dict(reset_at=4, steps_at={2: 3, 5: 1, 6: 5})
    _assert_same(_run(G, seq, "f32", 2, **kw), _run(G, seq, "f32", 1, **kw))


def test_1080p_sequence(G):
    """BASELINE.json configs[1] (1920x1080 fp32, 5 iterations): launches long enough for the two streams to really overlap."""
    seq = frames(1920, 1080, 6, mv=(2.5, -1.5))
    _assert_same(_run(G, seq, "f32", 2, own_stream=True), _run(G, seq, "f32", 1, own_stream=True))


def test_switching_the_mode_mid_sequence(G):
    """1 -> 2 -> 1 frames in flight inside one sequence: switching back orders the frame in flight at once; results unchanged."""
    import torch
    from svgf_amd import filter as F
    seq = frames(384, 216, 9)
    want, _, _ = _run(G, seq, "f32", 1)
    d = F.Denoiser(384, 216, F.Params(storage="f32", steps=5))
    gbs = [G.gb_dev(f) for f in seq]
    got, waiting = [], None
    for k, f in enumerate(seq):
        if k == 3:
            d.set_frames_in_flight(2)
        if k == 7:
            d.set_frames_in_flight(1)                    # frame 6 is ordered on the stream here
            got.append(waiting.clone())
            waiting = None
        v = d.Render(G.dev(f["radiance"]), gbs[k], gbs[k - 1] if k else None)
        if 3 <= k < 7:
            if waiting is not None:
                got.append(waiting.clone())
            waiting = v
        else:
            got.append(v.clone())
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for k, (x, y) in enumerate(zip(got, want)):
        assert np.array_equal(G.host(x).view(np.uint8), y.view(np.uint8)), f"frame {k}"


def test_filter_plane_and_debug_view_after_switching_back_to_one_frame(G):
    """2 -> 1 frames in flight after an ODD number of frames in flight: the last result sits in the second pair of filter planes.
    svgf_state_plane(FILTER, ..) and the SVGF_DEBUG_ATROUS view (which filters what the previous frame left in FilterBuffer,
    App.cu:611-620) must see THAT plane: both are compared with a context that never left one frame in flight."""
    from svgf_amd import filter as F
    seq = frames(320, 180, 8)
    gbs = [G.gb_dev(f) for f in seq]
    one = F.Denoiser(320, 180, F.Params(storage="f32", steps=4))
    two = F.Denoiser(320, 180, F.Params(storage="f32", steps=4))
    for nflight in (3, 4):                               # odd and even counts of frames in flight
        one.reset_history(); two.reset_history()
        two.set_frames_in_flight(2)
        for k in range(nflight):
            r1 = one.Render(G.dev(seq[k]["radiance"]), gbs[k], gbs[k - 1] if k else None).clone()
            two.Render(G.dev(seq[k]["radiance"]), gbs[k], gbs[k - 1] if k else None)
        two.set_frames_in_flight(1)
        idx = 4 & 1                                      # steps = 4: the result is in FilterBuffer[0] (no odd-N copy, App. B #12)
        assert np.array_equal(G.host(two.state_plane(F.PLANE_FILTER, idx)).view(np.uint8), G.host(r1).view(np.uint8)), nflight
        # ... and a second 2 -> 1 switch WITHOUT a frame in between must not rename the pairs again (the switch used to infer
        # "the last frame used the second pair" from the toggle, which also reads that way when no frame ran since the last switch)
        two.set_frames_in_flight(2)
        two.set_frames_in_flight(1)
        assert np.array_equal(G.host(two.state_plane(F.PLANE_FILTER, idx)).view(np.uint8), G.host(r1).view(np.uint8)), (nflight, "2 -> 1 -> 2 -> 1")
        one.set_debug_mode("atrous"); two.set_debug_mode("atrous")
        k = nflight
        a = one.Render(G.dev(seq[k]["radiance"]), gbs[k], gbs[k - 1])
        b = two.Render(G.dev(seq[k]["radiance"]), gbs[k], gbs[k - 1])
        assert np.array_equal(G.host(a).view(np.uint8), G.host(b).view(np.uint8)), f"debug view after {nflight} frames in flight"
        one.set_debug_mode("final"); two.set_debug_mode("final")


def test_two_frames_in_flight_with_iterations_the_direct_kernel_runs(G):
    """Seven iterations (steps 1..64: all LDS launches on the guide plane) and nine (steps 128, 256 go to the direct kernel, which reads the
    caller's G-buffer planes: such a frame's tail stays on the caller's stream) — bitwise equal to one frame at a time, and the
    caller may overwrite `cur` as soon as the NEXT call has returned."""
    for steps in (7, 9):
        seq = frames(384, 216, 6, mv=(1.0, 0.0))
        _assert_same(_run(G, seq, "f32", 2, steps=steps), _run(G, seq, "f32", 1, steps=steps))


def test_resize_with_a_frame_in_flight(G):
    from svgf_amd import filter as F
    a, b = frames(320, 200, 3), frames(448, 256, 4)
    d = F.Denoiser(320, 200, F.Params(storage="f32", steps=5))
    d.set_frames_in_flight(2)
    for k, f in enumerate(a):
        d.Render(G.dev(f["radiance"]), G.gb_dev(f), None)
    d.Resize(448, 256)                                   # waits for both streams, frees both pairs of filter planes
    outs, waiting = [], None
    gbs = [G.gb_dev(f) for f in b]
    for k, f in enumerate(b):
        v = d.Render(G.dev(f["radiance"]), gbs[k], gbs[k - 1] if k else None)
        if waiting is not None:
            outs.append(waiting.clone())
        waiting = v
    d.flush()
    outs.append(waiting.clone())
    want, _, _ = _run(G, b, "f32", 1)
    for k, (x, y) in enumerate(zip(outs, want)):
        assert np.array_equal(G.host(x).view(np.uint8), y.view(np.uint8)), f"frame {k}"


def test_refusals(G):
    from svgf_amd import filter as F
    d = F.Denoiser(64, 64, F.Params(storage="f32", steps=3))
    with pytest.raises(F.SvgfError, match="1 or 2"):
        d.set_frames_in_flight(3)
    d.set_frames_in_flight(2)
    with pytest.raises(F.SvgfError, match="frames in flight"):
        d.set_debug_mode("atrous")
    d.set_frames_in_flight(1)
    d.set_debug_mode("temporal")
    with pytest.raises(F.SvgfError, match="debug view"):
        d.set_frames_in_flight(2)


def test_stage_timing_with_two_frames_in_flight(G):
    """The per-stage HIP events still bracket every launch (the tail's on the side stream, from its own start)."""
    seq = frames(640, 360, 6)
    _, _, t = _run(G, seq, "f32", 2, timing=True)
    ms, n = t
    assert n == 6 and all(0.0 < v < 50.0 for v in ms[:7]), (ms, n)


def test_stream_change_and_destruction_with_a_frame_in_flight(G):
    """svgf_set_stream while a frame's tail is on the side stream (the frame is ordered on the NEW stream), and svgf_destroy with one
    in flight (waits for both streams)."""
    import torch
    from svgf_amd import filter as F
    seq = frames(384, 216, 6, mv=(1.0, 0.5))
    want, _, _ = _run(G, seq, "f32", 1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d = F.Denoiser(384, 216, F.Params(storage="f32", steps=5), stream=s1.cuda_stream)
    d.set_frames_in_flight(2)
    gbs = [G.gb_dev(f) for f in seq]
    rads = [G.dev(f["radiance"]) for f in seq]
    torch.cuda.synchronize()
    got, waiting, cur = [], None, s1
    for k in range(len(seq)):
        if k == 3:
            s2.wait_stream(s1)                           # what the caller itself enqueued on the old stream ...
            d.set_stream(s2.cuda_stream)                 # ... and the frame in flight is ordered on the new one by the library
            cur = s2
            with torch.cuda.stream(cur):
                got.append(waiting.clone())
            waiting = None
        v = d.Render(rads[k], gbs[k], gbs[k - 1] if k else None)
        with torch.cuda.stream(cur):
            if waiting is not None:
                got.append(waiting.clone())
        waiting = v
    d.flush()
    with torch.cuda.stream(cur):
        got.append(waiting.clone())
    torch.cuda.synchronize()
    for k, (x, y) in enumerate(zip(got, want)):
        assert np.array_equal(G.host(x).view(np.uint8), y.view(np.uint8)), f"frame {k}"
    d.Render(rads[0], gbs[0], gbs[1])                    # a frame in flight ...
    d.close()                                            # ... at destruction
    torch.cuda.synchronize()
