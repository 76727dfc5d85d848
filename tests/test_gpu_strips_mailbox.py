"""The C++ strip driver (svgf_amd/csrc/svgf_strip.hip) with REAL peer addressing on one device.

Every other single-device test of the driver runs it over a loop-back RCCL communicator: one communicator of size 1, every send and
receive addressed to rank 0, all virtual ranks on one shared communication stream — so the branch a multi-GPU run takes (a
communication stream per rank, sends to `rank +- 1`, receives from `rank +- 1`, the ranks' groups matched against each other) never ran
before the first 8-GPU job.  SVGF_TRANSPORT_MAILBOX runs exactly that branch: every rank of the partition lives in this process with
its own streams and events, posts the messages svgf_strips_messages lists with its neighbours' real rank numbers, and the library matches
each send to the receive the peer posted for it — in posting order per (source, destination) pair, inside the same group: RCCL's rule —
and fails the frame when a message has no partner.  Results must equal the whole frame BIT FOR BIT (history and moments included)."""
import ctypes as C

import numpy as np
import pytest

from svgf_amd import synth
from tests.helpers import frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


def _strip_inputs(G, fr, lay, storage):
    from svgf_amd import filter as F
    sl = slice(lay["y0"], lay["y1"])
    gb = F.GBuffer(*(G.dev(np.ascontiguousarray(fr[k][sl])) for k in ("motion", "normal", "uv")))
    rad = G.dev(np.ascontiguousarray(fr["radiance"][sl].astype(G.NPDT[storage])))
    return rad, gb


def _expected_traffic(W, H, world, steps, plan, reach, storage, radius=3):
    from svgf_amd import strips
    msgs = [m for r in range(world) for m in strips.strip_messages(W, H, r, world, steps, plan, radius, reach, storage)]
    sends = [m for m in msgs if m["send"]]
    return len({m["exchange"] for m in msgs}), len(sends), sum(m["bytes"] for m in sends)


def _run_sequence(G, W, H, world, plan, storage, fr, reach, params=None, own_streams=True, edge_first=True):
    """N frames of `fr` through the mailbox driver and through the single-context stage calls; asserts bitwise equality of every frame,
    of the final history and moments, and that the transport matched exactly the messages svgf_strips_messages lists."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    kw = dict(steps=5)
    kw.update(params or {})
    P = F.Params(storage=storage, **kw)
    whole = G.HipPipeline(W, H, storage, **kw)
    streams = [torch.cuda.Stream(priority=-1) for _ in range(world)] if own_streams else None     # a compute stream per rank, as in a real run
    drv = strips.NativeStrips(W, H, world, P, list(range(world)), [0] * world, streams=[s.cuda_stream for s in streams] if streams else None,
                              plan=plan, motion_reach=reach, transport="mailbox")
    drv.set_edge_first(edge_first)
    gbs = [G.gb_dev(f) for f in fr]
    torch.cuda.synchronize()
    prev_in = None
    for k in range(len(fr)):
        want = whole.frame(fr[k]["radiance"], gbs[k], gbs[max(k - 1, 0)])
        torch.cuda.synchronize()
        cur_in = [_strip_inputs(G, fr[k], lay, storage) for lay in drv.layouts]
        torch.cuda.synchronize()
        outs = drv.frame([c[0] for c in cur_in], [c[1] for c in cur_in], [p[1] for p in prev_in] if prev_in else None)
        drv.sync()
        got = np.concatenate([G.host(drv.owned(r, o)) for r, o in enumerate(outs)], 0)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"world {world}, plan {plan}, {storage}: frame {k}"
        prev_in = cur_in
    hist = np.concatenate([G.host(drv.owned(r, drv.state_plane(r, F.PLANE_HISTORY, 1 - drv.pingpong(r)))) for r in range(world)], 0)
    assert np.array_equal(hist, whole.taps["hist"])
    mom = np.concatenate([G.host(drv.owned(r, drv.state_plane(r, F.PLANE_MOMENTS, 1 - drv.pingpong(r)))) for r in range(world)], 0)
    assert np.array_equal(mom.view(np.uint8), whole.taps["mom"].view(np.uint8))
    nex, nsend, nbytes = _expected_traffic(W, H, world, P.steps, drv.plan, reach, storage, P.moments_radius)
    groups, copies, moved = drv.transport_stats()
    assert (groups, copies, moved) == (nex * len(fr), nsend * len(fr), nbytes * len(fr)), ((groups, copies, moved), (nex, nsend, nbytes))
    drv.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("plan", ["ghost", "grouped", "per-iteration"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_real_peer_addressing_small_worlds(G, world, plan, storage):
    """World 2 (every rank has ONE neighbour) and 3 (the middle rank has two): a pan of (1.0, -3.5) px per frame, motion reach 4, so that the
    state exchange carries colour, moments AND history rows that the next frame's reprojection really reads."""
    W, H = 320, 420
    _run_sequence(G, W, H, world, plan, storage, frames(W, H, 4, mv=(1.0, -3.5)), reach=4)


@pytest.mark.parametrize("plan", ["grouped", "per-iteration"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_three_launches_per_exchanging_iteration_still_give_the_same_bits(G, plan, storage):
    """svgf_strips_set_edge_first(0): round 4's schedule — two edge launches, the exchange behind an event, an interior launch — stays behind the
    ABI (devices without stream memory operations, iterations the direct kernel runs).  The default — ONE launch whose first workgroups
    produce the edge rows and write the word the communication stream waits for — is what every other test of this file runs."""
    W, H = 320, 420
    _run_sequence(G, W, H, 3, plan, storage, frames(W, H, 4, mv=(1.0, -3.5)), reach=4, edge_first=False)


def test_strips_no_taller_than_their_two_edges(G):
    """64-row strips under the per-iteration plan: at step 16 the two 32-row edge ranges ARE the strip — no interior to split off: that iteration is one
    plain launch with the exchange behind an event (the `hi <= lo` branch of the schedule), the others run edge rows first."""
    W, H = 256, 192
    _run_sequence(G, W, H, 3, "per-iteration", "f32", frames(W, H, 3, mv=(1.0, -1.5)), reach=2)


def test_edge_rows_first_on_odd_sizes_and_seven_iterations(G):
    """W % 128 != 0 (a last x tile that is mostly outside the frame), strips of unequal height (H % world != 0), seven iterations (steps 32 and 64:
    edge ranges of 128 / 256 rows... as far as the strips allow: the plan keeps them per iteration) — every exchanging iteration a single launch
    over {top edge, bottom edge, interior} with bands cut per range."""
    W, H = 333, 1621
    _run_sequence(G, W, H, 3, "per-iteration", "f32", frames(W, H, 3, mv=(-1.0, 2.5)), reach=3, params=dict(steps=7))


@pytest.mark.parametrize("plan,storage", [("ghost", "f32"), ("grouped", "f16"), ("per-iteration", "f32"), ("per-iteration", "f16"), ("grouped", "f32"), ("ghost", "f16")])
def test_config4_row_geometry_eight_ranks(G, plan, storage):
    """BASELINE.json configs[3]'s partition — 4320 rows as 8 strips of 540 — on a narrow frame, every halo plan, both storages, under a pan with
    motion reach 4, three frames (the state exchange of frames 0 and 1 is consumed)."""
    W, H = 384, 4320
    _run_sequence(G, W, H, 8, plan, storage, frames(W, H, 3, mv=(-1.5, 3.5)), reach=4)


def test_config4_full_size_per_iteration_plan(G):
    """configs[3] itself: 7680x4320 fp32 as 8 strips of 540 rows with "RCCL halo exchange per a-trous iter" (the per-iteration plan), each rank
    addressing its real neighbours; two frames of a pan, bitwise against the whole frame on the same device.  (The ghost plan at full size
    over the loop-back communicator: test_gpu_round2.py::test_config4_8k_as_eight_strips_of_540_rows.)"""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, storage, mv = 7680, 4320, 8, "f32", (2.0, -3.0)
    drv = strips.NativeStrips(W, H, world, F.Params(storage=storage, steps=5), list(range(world)), [0] * world, plan="per-iteration", motion_reach=3, transport="mailbox")
    assert [lay["own"][1] - lay["own"][0] for lay in drv.layouts] == [540] * 8 and drv.layouts[3]["y0"] == 1620 - 32
    whole = G.HipPipeline(W, H, storage, steps=5)
    prev_gb, prev_strip = None, None
    for k in range(2):
        sc = synth.make_scene(W, H, k, mv=mv)
        rad_np = synth.make_radiance(sc["base"], W, k)
        gb = G.gb_dev(sc)
        want = torch.from_numpy(whole.frame(rad_np, gb, prev_gb if prev_gb is not None else gb))
        rad = G.dev(rad_np)
        strip_gb = [F.GBuffer(*(t[lay["y0"]:lay["y1"]].contiguous() for t in (gb.motion, gb.normal, gb.uv))) for lay in drv.layouts]
        outs = drv.frame([rad[lay["y0"]:lay["y1"]].contiguous() for lay in drv.layouts], strip_gb, prev_strip)
        drv.sync()
        got = torch.cat([drv.owned(r, o) for r, o in enumerate(outs)], 0).cpu()
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), f"8K / 8 ranks: frame {k}"
        prev_gb, prev_strip = gb, strip_gb
    groups, copies, moved = drv.transport_stats()
    assert groups == 2 * 5 and copies == 2 * (7 * 2) * (3 + 4)            # 7 boundaries x 2 directions x (3 state planes + 4 filter exchanges)
    drv.close()


@pytest.mark.parametrize("params", [dict(moments_radius=0), dict(moments_radius=2), dict(phi_normal=0.0), dict(moments_radius=1), dict(steps=1), dict(steps=0)])
def test_real_peer_addressing_with_other_tunables(G, params):
    """The moments radius changes every halo of the plan, PhiNormal = 0 sends the iterations to the direct kernel, and under the
    default variants a radius other than 3 / 1 or PhiNormal = 0 has no every-pixel moments kernel for the first three frames — the strip driver
    shares choose_moments_kernel with the frame driver.  W % 64 != 0: the right-most 64-column segment is partly outside the frame."""
    W, H = 203, 420
    _run_sequence(G, W, H, 3, "grouped" if params.get("steps", 5) > 3 else "per-iteration", "f32", frames(W, H, 5, mv=(-2.5, 1.5)), reach=2, params=params, own_streams=False)


def test_two_frames_in_flight_with_real_peer_addressing(G):
    """svgf_strips_set_frames_in_flight(2) moves iterations 1.. — and their exchanges — to a side stream of every rank: the exchanges' ready / done
    events then tie the communication stream to TWO streams per rank.  Six frames without a sync in between, read as svgf.h prescribes."""
    import torch
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, N, storage, plan = 320, 420, 3, 6, "f32", "per-iteration"
    fr = frames(W, H, N, mv=(1.0, -2.5))
    whole = G.HipPipeline(W, H, storage, steps=5)
    side = [torch.cuda.Stream(priority=-1) for _ in range(world)]
    drv = strips.NativeStrips(W, H, world, F.Params(storage=storage, steps=5), list(range(world)), [0] * world, streams=[s.cuda_stream for s in side], plan=plan,
                              motion_reach=3, transport="mailbox")
    drv.set_frames_in_flight(2)
    gbs = [G.gb_dev(f) for f in fr]
    want = [whole.frame(fr[k]["radiance"], gbs[k], gbs[max(k - 1, 0)]) for k in range(N)]
    inputs = [[_strip_inputs(G, fr[k], lay, storage) for lay in drv.layouts] for k in range(N)]
    torch.cuda.synchronize()
    outs, got = {}, {}

    def collect(k):          # on each rank's compute stream: ordered behind frame k by call k + 1
        got[k] = []
        for r, o in enumerate(outs[k]):
            with torch.cuda.stream(side[r]):
                got[k].append(drv.owned(r, o).clone())
    for k in range(N):
        outs[k] = drv.frame([c[0] for c in inputs[k]], [c[1] for c in inputs[k]], [p[1] for p in inputs[k - 1]] if k else None)
        if k >= 1:
            collect(k - 1)
    drv.sync()
    collect(N - 1)
    torch.cuda.synchronize()
    for k in range(N):
        g = np.concatenate([G.host(t) for t in got[k]], 0)
        assert np.array_equal(g.view(np.uint8), want[k].view(np.uint8)), f"frame {k}"
    drv.close()


@pytest.mark.parametrize("fault,text", [(1, "waits for"), (2, "posts no receive"), (3, "expects")])
def test_the_mailbox_refuses_what_would_deadlock_a_real_run(G, fault, text):
    """The matching itself: drop one send / one receive of rank 1, or post one of its receives with half the size.  On a node the frame would
    hang in ncclGroupEnd's kernels; here the frame that meets the defect fails with SVGF_ERR_COMM naming the ranks, and the driver refuses
    every later frame.  (Without this the bitwise tests above could pass over a mailbox that matches anything with anything.)"""
    from svgf_amd import filter as F
    from svgf_amd import strips
    W, H, world, storage = 256, 420, 3, "f32"
    fr = frames(W, H, 2)
    # (motion reach 0: the state exchange carries the colour plane only — one message per pair and direction, so that a dropped message is a
    # MISSING message and not a shifted pairing of the planes behind it)
    drv = strips.NativeStrips(W, H, world, F.Params(storage=storage, steps=5), list(range(world)), [0] * world, plan="grouped", motion_reach=0, transport="mailbox")
    cur = [_strip_inputs(G, fr[0], lay, storage) for lay in drv.layouts]
    drv.frame([c[0] for c in cur], [c[1] for c in cur], None)
    drv.sync()
    assert drv.lib.svgf_strips_mailbox_fault(drv._h, 1, fault) == 0
    with pytest.raises(F.SvgfError, match=text) as e:
        drv.frame([c[0] for c in cur], [c[1] for c in cur], [c[1] for c in cur])
    assert "RCCL error" in str(e.value) and "rank 1" in str(e.value)
    with pytest.raises(F.SvgfError, match="earlier exchange failed"):
        drv.frame([c[0] for c in cur], [c[1] for c in cur], [c[1] for c in cur])
    drv.close()
    # and the transports that are not the mailbox refuse the hook, as a world that is not fully local refuses the mailbox
    h = C.c_void_p()
    p = F.Params(storage=storage, steps=5).to_c()
    ranks, devs = (C.c_int * 2)(0, 1), (C.c_int * 2)(0, 0)
    assert drv.lib.svgf_strips_create(C.byref(h), W, H, 3, C.byref(p), 0, 0, 2, ranks, devs, None, None, F.TRANSPORT["mailbox"]) == -1
