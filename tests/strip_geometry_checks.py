"""The checks of the strip driver's GEOMETRY (no device): svgf_strips_plan against the Python restatement of the partition, and the pairing of the
messages svgf_strips_messages lists (what svgf_strips_frame hands to its transport).  Used by tests/test_abi.py on a grid and on seeded random
partitions; `python -m tests.strip_geometry_checks --minutes 2 [--seed S]` sweeps more of the latter."""
from __future__ import annotations

import numpy as np
import pytest

PLANE_BYTES = {0: 16, 1: 8, 2: 16, 3: 1}          # colour, moments, filter (fp32 storage), history


def check_plan_against_python(W, H, world, steps, plan, mr, reach, ranks=None):
    """-> the number of ranks compared (0: both sides refuse the partition)."""
    from svgf_amd import strips
    n = 0
    for rank in (range(world) if ranks is None else ranks):
        try:
            g = strips.Geometry.make(W, H, rank, world, steps, plan=plan, moments_radius=mr, motion_reach=reach)
        except ValueError:
            with pytest.raises(ValueError):
                strips.strips_plan(W, H, rank, world, steps, plan, mr, reach)
            continue
        lay = strips.strips_plan(W, H, rank, world, steps, plan, mr, reach)
        assert (lay["y0"], lay["y1"], lay["own"]) == (g.y0, g.y1, g.own), (W, H, world, steps, plan, rank)
        assert lay["ext_atrous"] == g.ext_atrous and lay["halo_group"] == g.halo_group
        assert (lay["ext_moments"], lay["ext_temporal"], lay["halo_state"], lay["halo_max"]) == (g.ext_moments, g.ext_temporal, g.halo_state, g.halo_max)
        assert lay["plan"] == (g.plan if isinstance(g.plan, str) else plan)
        assert 0 <= lay["y0"] <= lay["own"][0] < lay["own"][1] <= lay["y1"] <= H
        n += 1
    return n


def check_messages(W, H, world, steps, plan, mr, reach, storage):
    """-> 1 if the partition exists and its messages pair up, 0 if it is refused (by the plan AND by svgf_strips_messages)."""
    from svgf_amd import strips
    try:
        lays = [strips.strips_plan(W, H, r, world, steps, plan, mr, reach) for r in range(world)]
    except ValueError:
        with pytest.raises(ValueError):
            strips.strip_messages(W, H, 0, world, steps, plan, mr, reach, storage)
        return 0
    # the owned rows tile the frame, in rank order
    assert lays[0]["own"][0] == 0 and lays[-1]["own"][1] == H and all(lays[r]["own"][1] == lays[r + 1]["own"][0] for r in range(world - 1)), (W, H, world)
    msgs = [strips.strip_messages(W, H, r, world, steps, plan, mr, reach, storage) for r in range(world)]
    nex = 1 + max(0, len(lays[0]["halo_group"]) - 1)
    # (no iteration and no motion reach: every state row a rank needs it has computed itself — nothing travels)
    state = steps > 0 or reach > 0
    if world > 1:
        assert {m["exchange"] for r in range(world) for m in msgs[r]} == set(range(0 if state else 1, nex)), (W, H, world, steps, plan)
    for ex in range(nex):
        for a in range(world):
            for b in (a - 1, a + 1):
                if not 0 <= b < world:
                    assert not [m for m in msgs[a] if m["peer"] == b]
                    continue
                sent = [(m["plane"], m["rows"], m["bytes"]) for m in msgs[a] if m["exchange"] == ex and m["send"] and m["peer"] == b]
                recv = [(m["plane"], m["rows"], m["bytes"]) for m in msgs[b] if m["exchange"] == ex and not m["send"] and m["peer"] == a]
                assert sent == recv and (sent or (ex == 0 and not state)), (W, H, world, steps, plan, ex, a, b, sent, recv)
    for r in range(world):
        own, y0, y1 = lays[r]["own"], lays[r]["y0"], lays[r]["y1"]
        for m in msgs[r]:
            assert abs(m["peer"] - r) == 1
            lo, hi = m["rows"]
            assert lo < hi and m["bytes"] == (hi - lo) * W * PLANE_BYTES[m["plane"]] // ((2 if m["plane"] != 3 else 1) if storage == "f16" else 1)
            if m["send"]:
                assert own[0] <= lo and hi <= own[1], (r, m, own)
            else:
                assert y0 <= lo and hi <= y1 and (hi <= own[0] or lo >= own[1]), (r, m, own, y0, y1)
    return 1


def random_case(seed):
    rng = np.random.default_rng(seed)
    world = int(rng.integers(1, 17))
    W = int(rng.choice([int(rng.integers(1, 9000)), 64, 128, 7680]))
    H = int(rng.choice([int(rng.integers(world, 6000)), 4320, 2160, 1080, world, world * 70 + int(rng.integers(0, world))]))
    steps = int(rng.integers(0, 11))
    plan = str(rng.choice(["ghost", "grouped", "per-iteration", "auto"]))
    mr, reach = int(rng.integers(0, 4)), int(rng.integers(0, 25))
    storage = ("f32", "f16")[int(rng.integers(0, 2))]
    n = check_plan_against_python(W, H, world, steps, plan, mr, reach)
    ok = check_messages(W, H, world, steps, plan, mr, reach, storage)
    assert (n > 0) == bool(ok)
    return ok


if __name__ == "__main__":
    import argparse
    import time
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=10000)
    a = ap.parse_args()
    t_end, seed, ok = time.monotonic() + a.minutes * 60, a.seed, 0
    while time.monotonic() < t_end:
        ok += random_case(seed)
        seed += 1
    print(f"strip_geometry_checks: seeds {a.seed}..{seed - 1}: {seed - a.seed} partitions, {ok} accepted by the plan, all consistent")
