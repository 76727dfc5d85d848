"""Error behaviour of the C ABI (include/svgf.h: "Errors are returned (0 = ok, negative = SVGF_ERR_*), never asserted" — the reference asserts,
App.cu:41-48): every entry point called the wrong way returns a negative status, leaves a text in svgf_last_error where there is a context
to keep it, launches nothing — and the context goes on working: the frame after a series of refused calls equals a fresh context's, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import frames

pytestmark = pytest.mark.gpu

W, H = 96, 40


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tests import gpu_helpers
    return gpu_helpers


@pytest.fixture()
def env(G):
    import torch
    from svgf_amd import filter as F
    lib = F.load_library()
    d = F.Denoiser(W, H, F.Params(storage="f32", steps=3))
    fr = frames(W, H, 3, mv=(1.0, -0.5))
    gbs = [G.gb_dev(f) for f in fr]
    rads = [G.dev(f["radiance"]) for f in fr]
    planes = dict(col=d.new_colour(), col2=d.new_colour(), col3=d.new_colour(), mom=d.new_moments(), mom2=d.new_moments(), hist=d.new_history(), hist2=d.new_history())
    torch.cuda.synchronize()
    yield dict(F=F, lib=lib, d=d, h=d._h, fr=fr, gbs=gbs, rads=rads, p=planes)
    d.close()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _runtime_is_clean():
    """Nothing is left pending in the HIP runtime for the host: a refused call must not make the NEXT runtime call of the host — here torch's, which
    checks hipGetLastError after its launches — report an error it did not cause (a refused device number used to: "invalid device ordinal")."""
    import torch
    t = torch.from_numpy(np.arange(8, dtype=np.float32)).to("cuda:0")
    assert float((t + 1).sum().item()) == 36.0
    torch.cuda.synchronize()


def test_null_context_everywhere(G):
    """Every entry point that takes a context or a strip driver refuses NULL (no crash): a negative status, NULL, 0 — or, for the two
    destructors, nothing."""
    from svgf_amd import filter as F
    lib = F.load_library()
    N = None
    gb = F.GBufferC(None, None, None)
    pc = F.ParamsC()
    lib.svgf_default_params(C.byref(pc))
    ip, dp, up, ull = C.c_int(), C.c_double(), C.c_uint(), C.c_ulonglong()
    st, lay, out = F.StripC(0, 4, 0, 4), F.StripLayoutC(), C.c_void_p()
    calls = {
        "svgf_set_params": (N, C.byref(pc)), "svgf_set_stream": (N, N), "svgf_set_rows": (N, 0, 1), "svgf_resize": (N, 8, 8), "svgf_resize_strip": (N, 8, 8, C.byref(st)),
        "svgf_get_size": (N, C.byref(ip), C.byref(ip), N), "svgf_sync": (N,), "svgf_halo_violations": (N, C.byref(ull), 0), "svgf_set_valid_rows": (N, 0, 1),
        "svgf_temporal": (N, N, N, N, C.byref(gb), C.byref(gb), N, N, N, N), "svgf_moments": (N, N, N, N, C.byref(gb), N),
        "svgf_temporal_moments": (N, N, N, N, N, C.byref(gb), C.byref(gb), N, N, N, N, 0, 0, 0), "svgf_atrous": (N, N, N, N, C.byref(gb), 1, 0),
        "svgf_atrous_pair": (N, N, N, N, C.byref(gb)), "svgf_set_iteration_fusion": (N, 1), "svgf_taa": (N, N, N, N), "svgf_demodulate": (N, N, N, N), "svgf_modulate": (N, N, N, N),
        "svgf_denoise_frame": (N, N, C.byref(gb), N, C.byref(out)), "svgf_reset_history": (N,), "svgf_set_frames_in_flight": (N, 2), "svgf_flush": (N,),
        "svgf_set_debug_mode": (N, 0), "svgf_set_prev_guide": (N, 1), "svgf_set_adaptive_moments": (N, 1), "svgf_adaptive_moments_sample": (N, C.byref(up), C.byref(up)),
        "svgf_import_gbuffer_pitched": (N, 0, N, 64, N), "svgf_import_gbuffer_array": (N, 0, N, N), "svgf_export_to_array": (N, N, N),
        "svgf_timing_enable": (N, 1), "svgf_timing_read": (N, C.byref(dp), C.byref(ip), 1),
        "svgf_path_stats_enable": (N, 1), "svgf_path_stats_read": (N, C.byref(ull), 1),
        "svgf_strips_layout": (N, 0, C.byref(lay)), "svgf_strips_frame": (N, N, N, N, N), "svgf_strips_sync": (N,), "svgf_strips_set_frames_in_flight": (N, 2),
        "svgf_strips_set_edge_first": (N, 1), "svgf_strips_timing_enable": (N, 1), "svgf_strips_timing_read": (N, C.byref(ip), C.byref(dp), C.byref(dp), C.byref(dp)),
        "svgf_strips_mailbox_fault": (N, 0, 0), "svgf_strips_transport_stats": (N, C.byref(ull), C.byref(ull), C.byref(ull)),
    }
    for name, args in calls.items():
        rc = getattr(lib, name)(*args)
        assert rc < 0, f"{name}(NULL, ...) returned {rc}"
        assert lib.svgf_status_string(rc), name
    assert lib.svgf_state_plane(N, 0, 0) is None and lib.svgf_plane_bytes(N, 0) == 0 and lib.svgf_strips_context(N, 0) is None
    assert lib.svgf_last_error(N) and lib.svgf_strips_last_error(N) is not None
    assert lib.svgf_adaptive_moments_state(N) <= 0 and lib.svgf_state_pingpong(N) <= 0
    lib.svgf_destroy(N)
    lib.svgf_strips_destroy(N)
    lib.svgf_default_params(N)
    _runtime_is_clean()


def test_create_refuses_bad_arguments(G):
    from svgf_amd import filter as F
    lib = F.load_library()
    pc = F.ParamsC()
    lib.svgf_default_params(C.byref(pc))

    def create(w=W, h=H, params=pc, device=0, out=True, **over):
        p = F.ParamsC()
        C.memmove(C.byref(p), C.byref(params), C.sizeof(p))
        for k, v in over.items():
            setattr(p, k, v)
        hdl = C.c_void_p(0x1234)
        rc = lib.svgf_create(C.byref(hdl) if out else None, w, h, C.byref(p), device, None)
        if rc == 0:
            lib.svgf_destroy(hdl)
        else:
            assert hdl.value in (None, 0x1234), "a refused svgf_create must not hand out a context"
        return rc
    assert create() == 0
    for kw in (dict(w=0), dict(w=-5), dict(h=0), dict(h=-1), dict(w=1 << 20, h=1 << 20), dict(out=False), dict(device=99), dict(device=-2),
               dict(storage=7), dict(storage=-1), dict(variant=42), dict(steps=-1), dict(steps=1000), dict(moments_radius=-1), dict(moments_radius=9), dict(nan_policy=5)):
        rc = create(**kw)
        assert rc < 0, (kw, rc)
    rc = lib.svgf_create(C.byref(C.c_void_p()), W, H, None, 0, None)      # NULL params: the defaults, or refused — never a crash
    assert rc <= 0
    # strips of a frame: row ranges that do not fit
    for strip in ((0, 0, 0, 0), (-1, 10, 0, 10), (0, H + 1, 0, H), (10, 10, 5, 15), (10, 10, 12, 11), (10, 10, 10, 25)):
        hdl = C.c_void_p()
        st = F.StripC(*strip)
        rc = lib.svgf_create_strip(C.byref(hdl), W, H, C.byref(st), C.byref(pc), 0, None)
        assert rc < 0, (strip, rc)
    assert lib.svgf_create_strip(C.byref(C.c_void_p()), W, H, None, C.byref(pc), 0, None) < 0
    _runtime_is_clean()


def test_stage_calls_refuse_what_they_cannot_run_and_the_context_goes_on(G, env):
    """Each required plane NULL in turn, planes aliased where the stage reads neighbours, steps < 1, row ranges outside the strip, enum values
    that do not exist: a negative status and a message; then three frames through the same context equal a fresh context's."""
    import torch
    F, lib, d, h, p, gbs, rads = env["F"], env["lib"], env["d"], env["h"], env["p"], env["gbs"], env["rads"]
    gc, gp = gbs[1].c, gbs[0].c                 # (GBuffer.c is already a byref)
    refused = []

    def no(rc, what):
        assert rc < 0, f"{what}: returned {rc}"
        msg = lib.svgf_last_error(h)
        assert msg and len(msg) > 3, what
        refused.append(what)
    good_t = [_p(p["col"]), _p(rads[1]), _p(p["col2"]), gc, gp, _p(p["hist"]), _p(p["hist2"]), _p(p["mom"]), _p(p["mom2"])]
    assert lib.svgf_temporal(h, *good_t) == 0
    for i in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        a = list(good_t)
        a[i] = None
        no(lib.svgf_temporal(h, *a), f"svgf_temporal arg {i} NULL")
    no(lib.svgf_temporal(h, good_t[0], good_t[1], good_t[0], *good_t[3:]), "svgf_temporal colour_out aliases prev_colour")
    empty = F.GBufferC(None, None, None)
    no(lib.svgf_temporal(h, good_t[0], good_t[1], good_t[2], C.byref(empty), gp, *good_t[5:]), "svgf_temporal empty G-buffer")
    good_m = [_p(p["col"]), _p(p["col2"]), _p(p["mom"]), gc, _p(p["hist"])]
    assert lib.svgf_moments(h, *good_m) == 0
    for i in range(5):
        a = list(good_m)
        a[i] = None
        no(lib.svgf_moments(h, *a), f"svgf_moments arg {i} NULL")
    no(lib.svgf_moments(h, good_m[0], good_m[0], *good_m[2:]), "svgf_moments in place")
    good_a = [_p(p["col"]), _p(p["col2"]), _p(p["col3"]), gc]
    assert lib.svgf_atrous(h, *good_a, 1, 0) == 0
    assert lib.svgf_atrous(h, good_a[0], good_a[1], None, gc, 2, 1) == 0          # no feedback plane: allowed
    for i in (0, 1, 3):
        a = list(good_a)
        a[i] = None
        no(lib.svgf_atrous(h, *a, 1, 0), f"svgf_atrous arg {i} NULL")
    for step in (0, -1, -(1 << 30)):
        no(lib.svgf_atrous(h, *good_a, step, 0), f"svgf_atrous step {step}")
    no(lib.svgf_atrous(h, good_a[0], good_a[0], good_a[2], gc, 1, 0), "svgf_atrous in place")
    no(lib.svgf_atrous_pair(h, good_a[0], good_a[0], good_a[2], gc), "svgf_atrous_pair in place")
    no(lib.svgf_atrous_pair(h, None, good_a[1], good_a[2], gc), "svgf_atrous_pair NULL input")
    no(lib.svgf_taa(h, _p(p["col"]), _p(p["col"]), _p(p["col"])), "svgf_taa in place")
    no(lib.svgf_taa(h, None, _p(p["col2"]), _p(p["col3"])), "svgf_taa NULL input")
    no(lib.svgf_demodulate(h, None, _p(p["col2"]), _p(p["col3"])), "svgf_demodulate NULL")
    no(lib.svgf_modulate(h, _p(p["col"]), None, _p(p["col3"])), "svgf_modulate NULL")
    out = C.c_void_p()
    no(lib.svgf_denoise_frame(h, None, gc, gp, C.byref(out)), "svgf_denoise_frame NULL radiance")
    no(lib.svgf_denoise_frame(h, _p(rads[0]), None, gp, C.byref(out)), "svgf_denoise_frame NULL G-buffer")
    no(lib.svgf_denoise_frame(h, _p(rads[0]), C.byref(empty), gp, C.byref(out)), "svgf_denoise_frame empty G-buffer")
    for rows in ((-1, 5), (7, 3), (0, H + 1), (H, H + 4)):
        no(lib.svgf_set_rows(h, *rows), f"svgf_set_rows {rows}")
    # an EMPTY range is a range: every stage call succeeds, launches nothing and writes nothing
    assert lib.svgf_set_rows(h, 5, 5) == 0
    sentinel = {k: torch.full_like(v, 7) for k, v in p.items()}
    st = [_p(p["col"]), _p(rads[1]), _p(sentinel["col2"]), gc, gp, _p(p["hist"]), _p(sentinel["hist2"]), _p(sentinel["mom"]), _p(p["mom2"])]
    assert lib.svgf_temporal(h, *st) == 0
    assert lib.svgf_moments(h, _p(p["col"]), _p(sentinel["col3"]), _p(p["mom"]), gc, _p(p["hist"])) == 0
    assert lib.svgf_atrous(h, _p(p["col"]), _p(sentinel["col"]), None, gc, 4, 1) == 0
    assert lib.svgf_atrous_pair(h, _p(p["col"]), _p(sentinel["col"]), _p(sentinel["col3"]), gc) == 0
    assert lib.svgf_taa(h, _p(p["col"]), _p(p["col2"]), _p(sentinel["col"])) == 0
    assert lib.svgf_modulate(h, _p(p["col"]), _p(p["col2"]), _p(sentinel["col"])) == 0
    torch.cuda.synchronize()
    assert all(bool((sentinel[k] == 7).all()) for k in ("col", "col2", "col3", "hist2", "mom")), "a stage call on an empty row range wrote something"
    assert lib.svgf_set_rows(h, -1, -1) == 0                                      # (-1, -1): the whole strip again
    for n in (0, 3, -1):
        no(lib.svgf_set_frames_in_flight(h, n), f"svgf_set_frames_in_flight {n}")
    no(lib.svgf_set_debug_mode(h, 99), "svgf_set_debug_mode 99")
    no(lib.svgf_set_debug_mode(h, -1), "svgf_set_debug_mode -1")
    no(lib.svgf_resize(h, 0, 10), "svgf_resize 0")
    no(lib.svgf_resize(h, 10, -3), "svgf_resize -3")
    pc = d.params.to_c()
    pc.storage = 1 - pc.storage
    no(lib.svgf_set_params(h, C.byref(pc)), "svgf_set_params changes the storage")
    no(lib.svgf_set_params(h, None), "svgf_set_params NULL")
    no(lib.svgf_timing_read(h, None, None, 4), "svgf_timing_read NULL")
    no(lib.svgf_import_gbuffer_pitched(h, 99, _p(p["col"]), 4096, _p(p["col2"])), "svgf_import_gbuffer_pitched plane 99")
    no(lib.svgf_import_gbuffer_pitched(h, 0, _p(p["col"]), 3, _p(p["col2"])), "svgf_import_gbuffer_pitched pitch 3")
    assert lib.svgf_state_plane(h, 99, 0) is None and lib.svgf_state_plane(h, 0, 2) is None and lib.svgf_plane_bytes(h, 99) == 0
    assert len(refused) >= 44
    _runtime_is_clean()
    # ... and the context is what it was: three frames equal a fresh context's
    fresh = F.Denoiser(W, H, F.Params(storage="f32", steps=3))
    for k in range(3):
        a = d.Render(rads[k], gbs[k], gbs[k - 1] if k else None)
        b = fresh.Render(rads[k], gbs[k], gbs[k - 1] if k else None)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), k
    fresh.close()


def test_strip_plans_and_drivers_refuse_bad_geometry(G):
    from svgf_amd import filter as F
    lib = F.load_library()
    lay = F.StripLayoutC()
    assert lib.svgf_strips_plan(320, 400, 0, 2, 5, 0, 3, 0, C.byref(lay)) == 0
    for args in ((0, 400, 0, 2, 5, 0, 3, 0), (320, 0, 0, 2, 5, 0, 3, 0), (320, 400, 2, 2, 5, 0, 3, 0), (320, 400, -1, 2, 5, 0, 3, 0), (320, 400, 0, 0, 5, 0, 3, 0),
                 (320, 400, 0, 2, -1, 0, 3, 0), (320, 400, 0, 2, 5, 99, 3, 0), (320, 400, 0, 2, 5, 0, -1, 0), (320, 400, 0, 2, 5, 0, 3, -1), (320, 3, 0, 8, 5, 0, 3, 0)):
        assert lib.svgf_strips_plan(*args, C.byref(lay)) < 0, args
    assert lib.svgf_strips_plan(320, 400, 0, 2, 5, 0, 3, 0, None) < 0
    n = C.c_int()
    assert lib.svgf_strips_messages(320, 400, 5, 2, 5, 0, 3, 0, 0, None, 0, C.byref(n)) < 0
    assert lib.svgf_strips_messages(320, 400, 0, 2, 5, 0, 3, 0, 9, None, 0, C.byref(n)) < 0
    pc = F.ParamsC()
    lib.svgf_default_params(C.byref(pc))
    ranks, devs = (C.c_int * 2)(0, 1), (C.c_int * 2)(0, 0)
    hdl = C.c_void_p()
    for kw in (dict(world=0), dict(world=2, nlocal=0), dict(world=2, nlocal=3), dict(world=2, ranks=(C.c_int * 2)(0, 0)), dict(world=2, ranks=(C.c_int * 2)(0, 5)),
               dict(world=2, devs=(C.c_int * 2)(0, 77)), dict(world=2, plan=42), dict(world=2, reach=-1), dict(world=2, transport=9), dict(world=2, w=0)):
        world, nlocal = kw.get("world", 2), kw.get("nlocal", 2)
        rc = lib.svgf_strips_create(C.byref(hdl), kw.get("w", 320), 400, world, C.byref(pc), kw.get("plan", 0), kw.get("reach", 0), nlocal, kw.get("ranks", ranks),
                                    kw.get("devs", devs), None, None, kw.get("transport", F.TRANSPORT["mailbox"]))
        assert rc < 0, (kw, rc)
    # two ranks, RCCL transport, no communicators: refused (the mailbox is the only transport that needs none)
    assert lib.svgf_strips_create(C.byref(hdl), 320, 400, 2, C.byref(pc), 0, 0, 2, ranks, devs, None, None, F.TRANSPORT["rccl"]) < 0
    _runtime_is_clean()
