"""The halo plan's arithmetic on CPU: geometry, lock-step virtual ranks, and a real world_size-2 torch.distributed run over gloo.

What runs here is the PYTHON restatement of the strip schedule (svgf_amd/strips.py: Geometry, StripRunner, LocalComm / DistComm) with the CPU
oracle as its stage backend (tests/oracle_stages.py) — NOT the product driver.  The product's strip driver is C++ (svgf_amd/csrc/svgf_strip.hip)
and is tested on the GPU, with real peer addressing, in tests/test_gpu_strips_mailbox.py; its geometry and its message lists are checked against
this restatement without a GPU in tests/test_abi.py.  What these tests establish is that the plan itself is right: which rows must travel when
for N strips to equal the whole frame bit for bit — over a real process group (world size 2), for every plan."""
import os
import socket

import numpy as np
import pytest
import torch

from svgf_amd import strips, synth
from tests.helpers import CDT, gbuf
from tests.oracle_stages import OracleStages

PARAMS = dict(steps=5, depth_threshold=0.8, normal_threshold=0.9, history_base=24, phi_colour=10.0, phi_normal=128.0,
              moments_radius=3, mesh_id_test=1)


def test_partition_and_geometry():
    assert strips.partition(4320, 8) == [(540 * r, 540 * (r + 1)) for r in range(8)]
    assert strips.partition(10, 3) == [(0, 3), (3, 6), (6, 10)]
    g = strips.Geometry.make(7680, 4320, 3, 8, 5, plan="per-iteration")
    assert g.groups == [[0], [1], [2], [3], [4]] and g.halo_group == [2, 4, 8, 16, 32] and g.ext_atrous == [0] * 5
    assert (g.ext_moments, g.ext_temporal, g.halo_state) == (2, 5, 9) and g.halo_max == 32
    g = strips.Geometry.make(7680, 4320, 3, 8, 5, plan="grouped")
    assert g.groups == [[0, 1, 2], [3, 4]] and g.halo_group == [14, 48] and g.ext_atrous == [12, 8, 0, 32, 0]
    assert (g.ext_moments, g.ext_temporal, g.halo_state, g.halo_max) == (14, 17, 21, 48)
    assert (g.y0, g.y1, g.own) == (1620 - 48, 2160 + 48, (1620, 2160))
    g = strips.Geometry.make(7680, 4320, 0, 8, 5, plan="ghost")
    assert g.halo_group == [62] and g.ext_atrous == [60, 56, 48, 32, 0] and g.halo_state == 62 + 3 + 4 and g.y0 == 0
    with pytest.raises(ValueError, match="shorter than"):
        strips.Geometry.make(640, 64, 0, 4, 5, plan="ghost")
    assert strips._subtract((0, 10), [(2, 4), (6, 10)]) == [(0, 2), (4, 6)]


def _local_inputs(fr, g):
    sl = slice(g.y0, g.y1)
    return {k: torch.from_numpy(np.ascontiguousarray(fr[k][sl])) for k in ("motion", "normal", "uv")}


def _reference(oracle, W, H, storage, frs):
    ref = oracle.Pipeline(W, H, storage, nthreads=4, **PARAMS)
    outs = []
    for k, fr in enumerate(frs):
        outs.append(ref.frame(fr["radiance"], gbuf(fr), gbuf(frs[max(k - 1, 0)])).copy())
    return outs, ref


@pytest.mark.parametrize("plan", ["per-iteration", "grouped", "ghost"])
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_virtual_ranks_bit_identical(oracle, plan, storage):
    W, H, world, N = 96, 312, 3, 4
    mv = (1.0, -2.5)
    frs = [synth.make_frame(W, H, f, mv=mv) for f in range(N)]
    want, ref = _reference(oracle, W, H, storage, frs)
    lc = strips.LocalComm()
    runners, geos = [], []
    for r in range(world):
        g = strips.Geometry.make(W, H, r, world, 5, plan=plan, motion_reach=3)
        geos.append(g)
        runners.append(strips.StripRunner(g, OracleStages(g, PARAMS, storage), lc.for_rank(r), storage=storage))
    for k in range(N):
        inputs = []
        for g in geos:
            rad = torch.from_numpy(np.ascontiguousarray(frs[k]["radiance"][g.y0:g.y1].astype(CDT[storage])))
            inputs.append((rad, _local_inputs(frs[k], g), _local_inputs(frs[max(k - 1, 0)], g)))
        outs = strips.run_virtual(runners, inputs)
        got = np.concatenate([r.owned(o).numpy() for r, o in zip(runners, outs)], 0)
        assert np.array_equal(got.view(np.uint8), want[k].view(np.uint8)), f"plan {plan}: frame {k} differs from the whole frame"
    hist = np.concatenate([r.owned(r.hist[r.P ^ 1]).numpy() for r in runners], 0)
    assert np.array_equal(hist, ref.hist[ref.P ^ 1])


def _free_port():
    import bench
    return bench.free_port()             # (below the ephemeral range: see there)


def _gloo_worker(rank, world, port, W, H, N, plan, storage, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = strips.Geometry.make(W, H, rank, world, 5, plan=plan, motion_reach=3)
        runner = strips.StripRunner(g, OracleStages(g, PARAMS, storage), strips.DistComm(), storage=storage)
        outs = []
        for k in range(N):
            fr = synth.make_frame(W, H, k, mv=(1.0, -2.5), row_begin=g.y0, row_end=g.y1)      # each rank makes only its rows
            fp = synth.make_frame(W, H, max(k - 1, 0), mv=(1.0, -2.5), row_begin=g.y0, row_end=g.y1)
            tz = lambda f: {n: torch.from_numpy(f[n]) for n in ("motion", "normal", "uv")}      # noqa: E731
            out = runner.frame(torch.from_numpy(fr["radiance"].astype(CDT[storage])), tz(fr), tz(fp))
            outs.append(runner.owned(out).numpy().copy())
        runner.flush()
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("plan", ["per-iteration", "grouped", "ghost"])
def test_gloo_world2_bit_identical(oracle, plan):
    """Two processes, torch.distributed over gloo, the Python schedule with oracle stages (see the module docstring: not the product driver)."""
    import torch.multiprocessing as mp
    W, H, N, world, storage = 64, 200, 3, 2, "f32"
    frs = [synth.make_frame(W, H, f, mv=(1.0, -2.5)) for f in range(N)]
    want, _ = _reference(oracle, W, H, storage, frs)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, W, H, N, plan, storage, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in range(N):
        got = np.concatenate([res[r][k] for r in range(world)], 0)
        assert np.array_equal(got.view(np.uint8), want[k].view(np.uint8)), f"frame {k}"


def test_auto_plan_exchanges_between_iterations():
    """SVGF_PLAN_AUTO (both restatements): grouped where its halo fits the strips — BASELINE.json configs[3] names a halo exchange between
    a-trous iterations, and ghost has none — else per-iteration; ghost only where neither applies (steps <= 1: all three coincide)."""
    from svgf_amd import filter as F
    assert strips.Geometry.make(7680, 4320, 3, 8, 5, plan="auto").plan == "grouped"
    assert strips.strips_plan(7680, 4320, 3, 8, 5, plan="auto")["plan"] == F.HALO_PLAN_NAME[F.HALO_PLAN["grouped"]]
    # strips of 40 rows: grouped's 48-row halo does not fit, per-iteration's 32 does
    assert strips.Geometry.make(640, 320, 3, 8, 5, plan="auto", motion_reach=0).plan == "per-iteration"
    assert strips.strips_plan(640, 320, 3, 8, 5, plan="auto", motion_reach=0)["plan"] == "per-iteration"
    with pytest.raises(ValueError):
        strips.Geometry.make(640, 160, 3, 8, 5, plan="auto")


def test_bench_line_of_an_eight_rank_run_from_canned_timings():
    """bench.py's rank-0 assembly of the N > 1 line (strips_line) on canned per-rank results for world 8 — no GPU, no communicator: the first
    real 8-GPU run must not die in Python glue after the measurement.  Every plan carries its speed-up over one GPU, its share of the
    aggregate roofline and the north_star's >= 6x target; the headline is the plan `auto` resolves to (grouped), never ghost."""
    import argparse
    import json
    import bench
    W, H, world = 7680, 4320, 8
    args = argparse.Namespace(steps=20, warmup=5)
    res = dict(ms_per_step=0.444, plan="grouped", rows_per_rank=540, rows_held=636, host_ms=0.31, motion_reach=0, edge_first=True, ms_per_step_three_launches=0.466,
               verified={"frames": 6, "plan": "grouped", "three_launches": True, "edge_first": True},
               atrous_timing=lambda b_it, b_fb: (25, 25 * 0.052, (5 * 560 * W) * 5 * b_it + (5 * 560 * W) * b_fb),
               driver="C++ (svgf_strips_frame)", one_gpu_ms=2.75, rccl_ranks=8, _comm=object(), pan={"mv": [1.5, -3.5], "motion_reach": 4, "plan": "grouped", "ms_per_step": 0.47, "rows_held": 644},
               other_plans={"per-iteration": dict(ms_per_step=0.4816, rows_held=604, host_ms=0.33, ms_per_step_three_launches=0.53),
                            "ghost": dict(ms_per_step=0.4685, rows_held=678, host_ms=0.29, edge_first=True)})
    line = bench.strips_line(res, args, W, H, "f32", 5, world)
    d = json.loads(json.dumps(line))                       # serialisable as it stands
    assert d["metric"] == bench.METRIC and d["n_gpus"] == 8 and d["unit"] == "Mpixels/s" and d["scaling"] == "strong"
    assert d["config"]["halo_plan"] == "grouped" and d["config"]["world_size"] == 8 and "7680x4320" in d["config"]["workload"]
    assert abs(d["value"] - W * H / 0.444e-3 / 1e6) < 1 and d["ms_per_step"] == 0.444
    assert set(d["halo_plans"]) == {"grouped", "per-iteration", "ghost"}
    for name, p in d["halo_plans"].items():
        assert p["target_speedup"] == 6.0 and p["target_met"] == (2.75 / p["ms_per_step"] >= 6.0), name
        assert abs(p["speedup_vs_one_gpu"] - 2.75 / p["ms_per_step"]) < 1e-3
        assert abs(p["frac_of_aggregate_8TBps"] - 459 * W * H / (p["ms_per_step"] * 1e-3) / 1e9 / 64000) < 1e-3
    assert d["halo_plans"]["grouped"]["target_met"] and not d["halo_plans"]["per-iteration"]["target_met"] and not d["halo_plans"]["ghost"]["target_met"]
    assert d["target_met"] is True and d["speedup_vs_one_gpu"] == d["halo_plans"]["grouped"]["speedup_vs_one_gpu"]
    assert d["fastest_plan"] == "grouped" and d["fastest_plan_that_exchanges_between_iterations"] == "grouped"
    assert d["config"]["edge_first"] is True and d["verified"]["edge_first"] is True and d["halo_plans"]["grouped"]["ms_per_step_three_launches"] == 0.466
    assert d["halo_plans"]["per-iteration"]["edge_first"] is False and d["halo_plans"]["ghost"]["edge_first"] is True
    assert d["roofline"]["frac"] > 0 and d["roofline"]["unit"] == "GB/s" and d["pan"]["Mpixels/s"] > 0 and d["rccl_ranks"] == 8
    # a leg that hung after the headline: the same line with what was measured, and `incomplete`
    part = dict(res, other_plans={}, pan=None)
    d2 = json.loads(json.dumps(bench.strips_line(part, args, W, H, "f32", 5, world, incomplete="leg 'plan ghost' did not finish")))
    assert set(d2["halo_plans"]) == {"grouped"} and d2["pan"] is None and "plan ghost" in d2["incomplete"]
    # other world sizes carry no target; no one-GPU reference: no speed-up
    d4 = bench.strips_line(dict(res, one_gpu_ms=None), args, W, H, "f32", 5, 4)
    assert d4["target_speedup"] is None and d4["speedup_vs_one_gpu"] is None and all(p["target_met"] is None for p in d4["halo_plans"].values())


def test_strip_verification_checksum_sees_a_flipped_bit_and_a_misplaced_row():
    """bench_strips checks every rank's owned rows against the one-GPU frame through strips._checksum (two 64-bit sums over the raw words, one of
    them position-weighted): equal planes give equal sums; one flipped bit, a -0.0 for a +0.0, or two rows swapped do not."""
    rng = np.random.default_rng(9)
    a = torch.from_numpy(rng.uniform(0, 1, (37, 129, 4)).astype(np.float32))
    s = strips._checksum(a)
    assert torch.equal(s, strips._checksum(a.clone()))
    b = a.clone(); b.view(torch.int32)[5, 7, 2] ^= 1
    assert not torch.equal(s, strips._checksum(b))
    c = a.clone(); c[3, 3, 1] = 0.0
    d = c.clone(); d[3, 3, 1] = -0.0
    assert not torch.equal(strips._checksum(c), strips._checksum(d))
    e = a.clone(); e[[4, 9]] = e[[9, 4]]
    assert torch.equal(strips._checksum(e)[0], s[0]) and not torch.equal(strips._checksum(e)[1], s[1])      # same words, other places
    h = torch.from_numpy(rng.uniform(0, 1, (8, 64, 4)).astype(np.float16))                                       # fp16 storage: 8 B per pixel = two words
    assert torch.equal(strips._checksum(h), strips._checksum(h.clone())) and strips._checksum(h).shape == (2,)


def _verify_worker(rank, world, port, q):
    """One rank of the bench's verification over gloo: a frame cut into `world` strips, every rank checksums ITS rows; rank 0 holds the reference."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W = 60, 40
        frame = torch.from_numpy(np.random.default_rng(3).uniform(0, 1, (H, W, 4)).astype(np.float32))
        parts = strips.partition(H, world)
        ref = torch.stack([strips._checksum(frame[a:b]) for a, b in parts]) if rank == 0 else None
        a, b = parts[rank]
        verdicts = [strips._all_ranks_match(strips._checksum(frame[a:b]), ref, rank, world)]            # every rank holds the right rows
        bad = frame[a:b].clone()
        if rank == world - 1:
            bad.view(torch.int32)[1, 2, 3] ^= 1                                                          # ... the last rank one flipped bit
        verdicts.append(strips._all_ranks_match(strips._checksum(bad), ref, rank, world))
        q.put((rank, verdicts))
    finally:
        dist.destroy_process_group()


def test_strip_verification_over_a_real_process_group():
    """strips._all_ranks_match (the collective half of bench_strips' check against the one-GPU frame) over torch.distributed with three gloo ranks and
    uneven strips: every rank learns "equal" when all strips are right, and every rank learns "not equal" when ONE rank's rows differ by one bit."""
    import torch.multiprocessing as mp
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r] == [True, False] for r in range(world)), res
