"""bench.py's input pools on the CPU (torch CPU tensors stand in for device memory): the pan pool must be a consistent camera
pan in both walking directions, and current / previous G-buffers must be distinct planes."""
import importlib
import sys

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def bench():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        return importlib.import_module("bench")
    finally:
        sys.argv = argv


def test_pan_pool_is_a_consistent_pan_forth_and_back(bench):
    W, H, P = 96, 64, 6
    scene = bench.Scene(W, H, torch.device("cpu"), pool=P, mv=(-2.5, 1.5), nmasks=2)
    pool = bench.FramePool(scene, "f32", "pan")
    depth = lambda gb: gb.motion[..., 2].numpy()          # noqa: E731
    seen = []
    prev_cur = None
    for n in range(3 * (P - 1) + 2):
        rad, cur, prev = pool.frame(n)
        assert rad.shape == (H, W, 4) and rad.dtype == torch.float32
        if n:
            assert prev is prev_cur, "the previous G-buffer of a frame is the current one of the frame before"
            assert cur.motion.data_ptr() != prev.motion.data_ptr() and cur.normal.data_ptr() != prev.normal.data_ptr()
            # the surface seen at p in the current frame was at p + trunc(mv) (+ the half pixel) in the previous one: depths agree there
            mvx, mvy = float(cur.motion[0, 0, 0]), float(cur.motion[0, 0, 1])
            assert (abs(mvx), abs(mvy)) == (2.5, 1.5)
            dx, dy = int(mvx), int(mvy)
            zc, zp = depth(cur), depth(prev)
            ys, xs = np.mgrid[8:H - 8, 8:W - 8]
            a, b = zc[ys, xs], zp[ys + dy, xs + dx]
            ok = (a > 0) & (b > 0)
            # ... inside DepthThreshold everywhere but at silhouettes (the reprojection truncates the half pixel: edges move by < 1 px)
            assert ok.mean() > 0.5 and (np.abs(a[ok] - b[ok]) < 0.8 * 0.9).mean() > 0.93, (n, (np.abs(a[ok] - b[ok]) < 0.72).mean())
        prev_cur = cur
        seen.append(float(cur.motion[0, 0, 0]))
    assert min(seen) == -2.5 and max(seen) == 2.5             # walked in both directions


def test_static_pool_ping_pongs_two_sets_of_planes(bench):
    W, H = 64, 48
    scene = bench.Scene(W, H, torch.device("cpu"), pool=2, nmasks=2)
    pool = bench.FramePool(scene, "f16", "static")
    r0, c0, p0 = pool.frame(0)
    r1, c1, p1 = pool.frame(1)
    assert r0.dtype == torch.float16 and c0 is p1 and c1 is p0 and c0 is not c1
    for a, b in ((c0.motion, c1.motion), (c0.normal, c1.normal), (c0.uv, c1.uv)):
        assert a.data_ptr() != b.data_ptr() and torch.equal(a, b)
    assert float(c0.motion[..., :2].abs().max()) == 0.0
    # the headline is the ABI's default (svgf_set_prev_guide off): 32 B/px of previous G-buffer at the reprojected address instead of the 16 of the kept guide plane
    assert bench.moved_bytes_full("f32", 5) == 146 + 5 * 48 + 16 and bench.moved_bytes_full("f32", 5, prev_guide=True) == 130 + 5 * 48 + 16
    assert bench.alg_bytes_full("f32", 5) == 459 and bench.alg_bytes_full("f16", 5) == 323


def test_rendezvous_port_is_below_the_ephemeral_range():
    """bench.free_port: a port for the N > 1 rendezvous that no outgoing connection of another process can take in the meantime."""
    import socket
    import bench
    for _ in range(5):
        p = bench.free_port()
        assert 20000 <= p < 32768
        with socket.socket() as s:
            s.bind(("127.0.0.1", p))
