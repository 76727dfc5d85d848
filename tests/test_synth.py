"""The synthetic input generator: strips and canvases tile exactly (what lets every rank, and bench.py's pan pool, cut its
frames out of independently generated pieces)."""
import numpy as np

from svgf_amd import synth


import pytest


@pytest.mark.parametrize("scene", synth.SCENES)
def test_rows_and_columns_tile_exactly(scene):
    W, H, mv = 200, 120, (-2.5, 1.5)
    whole = synth.make_frame(W, H, 3, mv=mv, scene=scene)
    parts = [synth.make_frame(W, H, 3, mv=mv, row_begin=a, row_end=b, scene=scene) for a, b in ((0, 37), (37, 90), (90, 120))]
    for k in ("motion", "normal", "uv", "radiance", "region", "base"):
        assert np.array_equal(np.concatenate([p[k] for p in parts], 0), whole[k]), k
    big = synth.make_scene(W, H, 3, mv=mv, row_begin=-7, row_end=130, col_begin=-11, col_end=220, scene=scene)
    for k in ("motion", "normal", "region", "base"):
        assert np.array_equal(big[k][7:127, 11:211], whole[k]), k


def test_curved_scene_has_a_normal_of_its_own_in_nearly_every_texel():
    """Scene "curved" (bench.py: also.curved_scene): smooth-shaded geometry, normalize(FragNormal) per texel (GBuffer.frag:65) — the planar scene
    SURVEY 8(d) prescribes keeps the a-trous kernel on its uniform-normal fast path, this one never does.  Unit normals, analytic depth derivative
    within the planar scene's range, the same sky band."""
    W, H = 640, 360
    c, p = synth.make_frame(W, H, 0, scene="curved"), synth.make_frame(W, H, 0)
    surf = c["region"] != synth.SKY
    assert np.array_equal(surf, p["region"] != synth.SKY) and 0.04 < (~surf).mean() < 0.2
    n = c["normal"][..., :3]
    both = surf[:, 1:] & surf[:, :-1]
    same_c = ((n[:, 1:] == n[:, :-1]).all(-1) & both).sum() / both.sum()
    pn = p["normal"][..., :3]
    same_p = ((pn[:, 1:] == pn[:, :-1]).all(-1) & both).sum() / both.sum()
    assert same_c < 0.02 and same_p > 0.9, (same_c, same_p)
    ln = np.linalg.norm(n.view(np.float16).astype(np.float32), axis=-1)[surf]
    assert 0.999 < ln.min() and ln.max() < 1.001
    dz = c["motion"][..., 3][surf]
    assert dz.min() >= 0 and dz.max() < 0.05 and np.all(c["motion"][..., 2][surf] > 1.0)
    assert np.all(c["normal"][~surf] == 0) and np.all(c["motion"][~surf][:, 2] == 0)
    # the planar scene is what it was (the golden fixtures and every pinned fuzz seed are made of it)
    assert np.array_equal(synth.make_frame(64, 48, 2)["normal"], synth.make_frame(64, 48, 2, scene="planar")["normal"])


def test_pan_frames_are_windows_of_two_canvases():
    """bench.py's Scene: frame f of a pan by mv (2*mv integral) is the canvas of parity f & 1 shifted by (f // 2) * 2 * mv."""
    W, H, mv = 160, 96, (-2.5, 1.5)
    sx, sy = int(2 * mv[0]), int(2 * mv[1])
    for f in range(6):
        k = f // 2
        canvas = synth.make_scene(W, H, f & 1, mv=mv, row_begin=min(0, sy * 3), row_end=H + max(0, sy * 3), col_begin=min(0, sx * 3), col_end=W + max(0, sx * 3))
        xa, ya = k * sx - min(0, sx * 3), k * sy - min(0, sy * 3)
        direct = synth.make_scene(W, H, f, mv=mv)
        for n in ("motion", "normal", "region", "base"):
            assert np.array_equal(canvas[n][ya:ya + H, xa:xa + W], direct[n]), (f, n)
        assert np.array_equal(canvas["uv"][ya:ya + H, xa:xa + W, 3], direct["uv"][..., 3])       # instance ids (the barycentrics are noise)


def test_sky_and_margins():
    f = synth.make_frame(320, 180, 0)
    sky = f["region"] == synth.SKY
    assert 0.04 < sky.mean() < 0.2
    assert np.all(f["motion"][sky][:, 2] == 0) and np.all(f["normal"][sky] == 0) and np.all(f["uv"][sky] == 0)
    assert np.all(f["radiance"][..., :3] >= 0) and np.all(f["radiance"][..., :3] <= 1) and np.all(f["radiance"][..., 3] == 1)
