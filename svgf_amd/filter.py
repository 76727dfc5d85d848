"""Python host-side mirror of the reference's filter entry points, bound to the C ABI of
include/svgf.h, svgf_ext.h and svgf_test.h (libsvgf_mi355x.so) with ctypes.

`Denoiser.TemporalFilter / FilterMoments / WaveletFilter / Render` keep the names, argument meaning
and sequencing of `application::TemporalFilter/FilterMoments/WaveletFilter` (src/App.cu:469-514)
and the `Render` stage order (src/App.cu:552-556).  PyTorch only supplies device memory and
streams; every computation happens in the HIP library.  There is no CPU fallback: if the library
or a GPU is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

from . import build as _build

SVGF_F32, SVGF_F16 = 0, 1
STORAGE = {"f32": SVGF_F32, "f16": SVGF_F16}
VARIANT = {"auto": 0, "direct": 1, "lds": 2, "lds-general": 3}
NAN_POLICY = {"reference": 0, "zero": 1}
PLANE_COLOUR, PLANE_MOMENTS, PLANE_FILTER, PLANE_HISTORY = 0, 1, 2, 3
MAX_STEPS = 10

EXPORTS = [
    "svgf_default_params", "svgf_status_string", "svgf_last_error", "svgf_abi_version", "svgf_create",
    "svgf_create_strip", "svgf_destroy", "svgf_set_params", "svgf_set_stream", "svgf_set_rows", "svgf_temporal", "svgf_temporal_moments", "svgf_demodulate", "svgf_modulate",
    "svgf_moments", "svgf_atrous", "svgf_atrous_pair", "svgf_set_iteration_fusion", "svgf_taa", "svgf_pack_gbuffer", "svgf_denoise_frame", "svgf_reset_history", "svgf_set_frames_in_flight", "svgf_flush", "svgf_state_plane",
    "svgf_state_pingpong", "svgf_plane_bytes", "svgf_timing_enable", "svgf_timing_read", "svgf_path_stats_enable", "svgf_path_stats_read",
    "svgf_resize", "svgf_resize_strip", "svgf_get_size", "svgf_sync", "svgf_halo_violations", "svgf_set_valid_rows", "svgf_set_debug_mode", "svgf_set_prev_guide", "svgf_set_adaptive_moments", "svgf_adaptive_moments_state", "svgf_adaptive_moments_sample",
    "svgf_import_gbuffer_pitched", "svgf_import_gbuffer_array", "svgf_export_to_array",
    "svgf_strips_plan", "svgf_rccl_unique_id", "svgf_rccl_comm_init", "svgf_rccl_comm_destroy", "svgf_rccl_comm_count", "svgf_strips_create", "svgf_strips_destroy",
    "svgf_strips_last_error", "svgf_strips_context", "svgf_strips_layout", "svgf_strips_frame", "svgf_strips_sync",
    "svgf_strips_timing_enable", "svgf_strips_timing_read", "svgf_strips_set_frames_in_flight", "svgf_strips_messages", "svgf_strips_transport_stats", "svgf_strips_mailbox_fault", "svgf_strips_set_edge_first",
]
ABI_VERSION = 8
PATH_STAT_STEPS = 7
TRANSPORT = {"rccl": 0, "rccl-loopback": 1, "mailbox": 2}
DEBUG_MODE = {"final": 0, "temporal": 1, "atrous": 2}
HALO_PLAN = {"auto": 0, "ghost": 1, "grouped": 2, "per-iteration": 3}
HALO_PLAN_NAME = {v: k for k, v in HALO_PLAN.items()}
GBUF_MOTION, GBUF_NORMAL, GBUF_UV = 0, 1, 2


class SvgfError(RuntimeError):
    pass


class GBufferC(C.Structure):
    _fields_ = [("motion", C.c_void_p), ("normal", C.c_void_p), ("uv", C.c_void_p)]


class ParamsC(C.Structure):
    _fields_ = [("steps", C.c_int), ("depth_threshold", C.c_float), ("normal_threshold", C.c_float),
                ("history_base", C.c_int), ("phi_colour", C.c_float), ("phi_normal", C.c_float),
                ("moments_radius", C.c_int), ("storage", C.c_int), ("mesh_id_test", C.c_int), ("variant", C.c_int), ("nan_policy", C.c_int)]


class StripMessageC(C.Structure):
    _fields_ = [("exchange", C.c_int), ("send", C.c_int), ("peer", C.c_int), ("plane", C.c_int), ("row_begin", C.c_int), ("row_end", C.c_int), ("bytes", C.c_size_t)]


class CameraC(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("prev_view_proj", C.c_float * 16), ("position", C.c_float * 3)]


class StripC(C.Structure):
    _fields_ = [("y0", C.c_int), ("rows", C.c_int), ("own_begin", C.c_int), ("own_end", C.c_int)]


class StripLayoutC(C.Structure):
    _fields_ = [("plan", C.c_int), ("strip", StripC), ("ext_atrous", C.c_int * MAX_STEPS), ("ngroups", C.c_int),
                ("group_first", C.c_int * MAX_STEPS), ("halo_group", C.c_int * MAX_STEPS), ("ext_moments", C.c_int),
                ("ext_temporal", C.c_int), ("halo_state", C.c_int), ("halo_max", C.c_int)]


@dataclass
class Params:
    """Tunables of src/App.h:109-114 (+ the build's storage / radius / variant switches)."""
    steps: int = 3
    depth_threshold: float = 0.8
    normal_threshold: float = 0.9
    history_base: int = 24
    phi_colour: float = 10.0
    phi_normal: float = 128.0
    moments_radius: int = 3
    storage: str = "f16"
    mesh_id_test: int = 1
    variant: str = "auto"
    nan_policy: str = "reference"         # "reference": a NaN texel stays NaN, as in Filter.cuh; "zero": the temporal stage reads it as 0 (svgf.h)

    def to_c(self) -> ParamsC:
        return ParamsC(self.steps, self.depth_threshold, self.normal_threshold, self.history_base, self.phi_colour,
                       self.phi_normal, self.moments_radius, STORAGE[self.storage], self.mesh_id_test,
                       VARIANT[self.variant], NAN_POLICY[self.nan_policy])


_lib = None


def library_path() -> str:
    # SVGF_LIBRARY lets the diagnostic tools load a -DSVGF_DIAG twin of the library; tests and bench never set it.
    return os.environ.get("SVGF_LIBRARY") or _build.LIB


def load_library():
    """dlopen the product library (building it first if hipcc is around and it is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if path == _build.LIB and _build.stale():
        try:
            _build.build_library()
        except Exception as e:  # noqa: BLE001
            # never fall back to a library older than the sources: tests or the bench would silently run outdated kernels
            if not os.path.exists(path):
                raise SvgfError(f"libsvgf_mi355x.so is missing and could not be built: {e}") from e
            if _build.have_hipcc():
                raise SvgfError(f"libsvgf_mi355x.so is older than its sources and the rebuild failed: {e}") from e
            import warnings
            warnings.warn(f"libsvgf_mi355x.so is older than its sources and there is no hipcc here to rebuild it ({e}); using it as shipped", stacklevel=2)
    if not os.path.exists(path):
        raise SvgfError("libsvgf_mi355x.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # torch ships a HIP runtime of its own (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's): it has to be the one
    # already in the process when this library is opened, otherwise the process holds two runtimes and the one torch's
    # tensors live in is not the one svgf_create talks to (hipGetDeviceCount fails -> "no usable gfx950 device").
    import torch  # noqa: F401
    lib = C.CDLL(path)
    vp, ip = C.c_void_p, C.c_int
    lib.svgf_default_params.argtypes = [C.POINTER(ParamsC)]
    lib.svgf_default_params.restype = None
    lib.svgf_status_string.argtypes = [ip]
    lib.svgf_status_string.restype = C.c_char_p
    lib.svgf_last_error.argtypes = [vp]
    lib.svgf_last_error.restype = C.c_char_p
    lib.svgf_abi_version.restype = ip
    lib.svgf_create.argtypes = [C.POINTER(vp), ip, ip, C.POINTER(ParamsC), ip, vp]
    lib.svgf_create_strip.argtypes = [C.POINTER(vp), ip, ip, C.POINTER(StripC), C.POINTER(ParamsC), ip, vp]
    lib.svgf_destroy.argtypes = [vp]
    lib.svgf_destroy.restype = None
    lib.svgf_set_params.argtypes = [vp, C.POINTER(ParamsC)]
    lib.svgf_set_stream.argtypes = [vp, vp]
    lib.svgf_set_rows.argtypes = [vp, ip, ip]
    lib.svgf_temporal.argtypes = [vp, vp, vp, vp, C.POINTER(GBufferC), C.POINTER(GBufferC), vp, vp, vp, vp]
    lib.svgf_demodulate.argtypes = [vp, vp, vp, vp]
    lib.svgf_modulate.argtypes = [vp, vp, vp, vp]
    lib.svgf_temporal_moments.argtypes = [vp, vp, vp, vp, vp, C.POINTER(GBufferC), C.POINTER(GBufferC), vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
    lib.svgf_moments.argtypes = [vp, vp, vp, vp, C.POINTER(GBufferC), vp]
    lib.svgf_atrous.argtypes = [vp, vp, vp, vp, C.POINTER(GBufferC), ip, ip]
    lib.svgf_atrous_pair.argtypes = [vp, vp, vp, vp, C.POINTER(GBufferC)]
    lib.svgf_set_iteration_fusion.argtypes = [vp, ip]
    lib.svgf_taa.argtypes = [vp, vp, vp, vp]
    lib.svgf_pack_gbuffer.argtypes = [vp, vp, vp, vp, C.POINTER(CameraC), vp, vp, vp]
    lib.svgf_denoise_frame.argtypes = [vp, vp, C.POINTER(GBufferC), C.POINTER(GBufferC), C.POINTER(vp)]
    lib.svgf_reset_history.argtypes = [vp]
    lib.svgf_set_frames_in_flight.argtypes = [vp, ip]
    lib.svgf_flush.argtypes = [vp]
    lib.svgf_state_plane.argtypes = [vp, ip, ip]
    lib.svgf_state_plane.restype = vp
    lib.svgf_state_pingpong.argtypes = [vp]
    lib.svgf_plane_bytes.argtypes = [vp, ip]
    lib.svgf_plane_bytes.restype = C.c_size_t
    lib.svgf_timing_enable.argtypes = [vp, ip]
    lib.svgf_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(ip), ip]
    lib.svgf_path_stats_enable.argtypes = [vp, ip]
    lib.svgf_path_stats_read.argtypes = [vp, C.POINTER(C.c_ulonglong), ip]
    lib.svgf_resize.argtypes = [vp, ip, ip]
    lib.svgf_resize_strip.argtypes = [vp, ip, ip, C.POINTER(StripC)]
    lib.svgf_get_size.argtypes = [vp, C.POINTER(ip), C.POINTER(ip), C.POINTER(StripC)]
    lib.svgf_sync.argtypes = [vp]
    lib.svgf_halo_violations.argtypes = [vp, C.POINTER(C.c_ulonglong), ip]
    lib.svgf_set_debug_mode.argtypes = [vp, ip]
    lib.svgf_set_prev_guide.argtypes = [vp, ip]
    lib.svgf_set_adaptive_moments.argtypes = [vp, ip]
    lib.svgf_adaptive_moments_state.argtypes = [vp]
    lib.svgf_adaptive_moments_sample.argtypes = [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.svgf_set_valid_rows.argtypes = [vp, ip, ip]
    lib.svgf_import_gbuffer_pitched.argtypes = [vp, ip, vp, C.c_size_t, vp]
    lib.svgf_import_gbuffer_array.argtypes = [vp, ip, vp, vp]
    lib.svgf_export_to_array.argtypes = [vp, vp, vp]
    lib.svgf_strips_plan.argtypes = [ip, ip, ip, ip, ip, ip, ip, ip, C.POINTER(StripLayoutC)]
    lib.svgf_rccl_unique_id.argtypes = [vp]
    lib.svgf_rccl_comm_init.argtypes = [C.POINTER(vp), ip, ip, vp, ip]
    lib.svgf_rccl_comm_destroy.argtypes = [vp]
    lib.svgf_rccl_comm_count.argtypes = [vp, C.POINTER(ip)]
    lib.svgf_strips_create.argtypes = [C.POINTER(vp), ip, ip, ip, C.POINTER(ParamsC), ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(vp), C.POINTER(vp), ip]
    lib.svgf_strips_messages.argtypes = [ip, ip, ip, ip, ip, ip, ip, ip, ip, C.POINTER(StripMessageC), ip, C.POINTER(ip)]
    lib.svgf_strips_transport_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.svgf_strips_mailbox_fault.argtypes = [vp, ip, ip]
    lib.svgf_strips_set_edge_first.argtypes = [vp, ip]
    lib.svgf_strips_destroy.argtypes = [vp]
    lib.svgf_strips_destroy.restype = None
    lib.svgf_strips_last_error.argtypes = [vp]
    lib.svgf_strips_last_error.restype = C.c_char_p
    lib.svgf_strips_context.argtypes = [vp, ip]
    lib.svgf_strips_context.restype = vp
    lib.svgf_strips_layout.argtypes = [vp, ip, C.POINTER(StripLayoutC)]
    lib.svgf_strips_frame.argtypes = [vp, C.POINTER(vp), C.POINTER(GBufferC), C.POINTER(GBufferC), C.POINTER(vp)]
    lib.svgf_strips_sync.argtypes = [vp]
    lib.svgf_strips_set_frames_in_flight.argtypes = [vp, ip]
    lib.svgf_strips_timing_enable.argtypes = [vp, ip]
    lib.svgf_strips_timing_read.argtypes = [vp, C.POINTER(ip), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _lib = lib
    return lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class GBuffer:
    """Device planes of one G-buffer = `cudaFramebuffer` (src/App.h:41-44) minus Position."""

    def __init__(self, motion, normal, uv):
        self.motion, self.normal, self.uv = motion, normal, uv
        self._c = GBufferC(motion.data_ptr(), normal.data_ptr(), uv.data_ptr() if uv is not None else None)

    @property
    def c(self):
        return C.byref(self._c)


class Denoiser:
    """One SVGF context on one device (`svgf_ctx`).  Method names follow src/App.cu:469-514."""

    def __init__(self, width, height, params: Params | None = None, *, device=0, stream=None, strip=None):
        import torch
        if not torch.cuda.is_available():
            raise SvgfError("svgf_amd needs an MI355X: no HIP device is visible and there is no CPU fallback")
        self.lib = load_library()
        self.params = params or Params()
        self.W, self.H = width, height
        self.device = torch.device("cuda", device)
        self.storage = self.params.storage
        self._torch = torch
        h = C.c_void_p()
        pc = self.params.to_c()
        s = C.c_void_p(stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream)
        if strip is None:
            self.strip = (0, height, 0, height)
            rc = self.lib.svgf_create(C.byref(h), width, height, C.byref(pc), device, s)
        else:
            self.strip = tuple(int(v) for v in strip)
            sc = StripC(*self.strip)
            rc = self.lib.svgf_create_strip(C.byref(h), width, height, C.byref(sc), C.byref(pc), device, s)
        if rc != 0:
            raise SvgfError(f"svgf_create failed: {self.lib.svgf_status_string(rc).decode()}")
        self._h = h

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise SvgfError(f"{what}: {self.lib.svgf_status_string(rc).decode()}: {self.lib.svgf_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.svgf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def rows(self):
        return self.strip[1]

    def colour_dtype(self):
        return self._torch.float32 if self.storage == "f32" else self._torch.float16

    def new_colour(self):
        return self._torch.zeros((self.rows, self.W, 4), dtype=self.colour_dtype(), device=self.device)

    def new_moments(self):
        return self._torch.zeros((self.rows, self.W, 2), dtype=self.colour_dtype(), device=self.device)

    def new_history(self):
        return self._torch.zeros((self.rows, self.W), dtype=self._torch.uint8, device=self.device)

    def set_params(self, params: Params):
        pc = params.to_c()
        self._check(self.lib.svgf_set_params(self._h, C.byref(pc)), "svgf_set_params")
        self.params = params

    def set_stream(self, stream_handle: int):
        self._check(self.lib.svgf_set_stream(self._h, C.c_void_p(stream_handle)), "svgf_set_stream")

    def set_rows(self, row_begin=-1, row_end=-1):
        self._check(self.lib.svgf_set_rows(self._h, row_begin, row_end), "svgf_set_rows")

    def set_valid_rows(self, row_begin=-1, row_end=-1):
        """Rows of the previous-frame planes that hold valid state (strips: own rows +- state halo)."""
        self._check(self.lib.svgf_set_valid_rows(self._h, row_begin, row_end), "svgf_set_valid_rows")

    def sync(self):
        """Wait for the context's stream; raises (SVGF_ERR_HALO) if a strip's temporal stage reprojected into rows it does not hold."""
        self._check(self.lib.svgf_sync(self._h), "svgf_sync")

    def halo_violations(self, clear=False) -> int:
        n = C.c_ulonglong()
        self._check(self.lib.svgf_halo_violations(self._h, C.byref(n), int(clear)), "svgf_halo_violations")
        return n.value

    def Resize(self, width, height, strip=None):
        """application::ResizeRenderTextures (src/App.cu:742-778): new render size, state reallocated and zeroed."""
        if strip is None:
            self._check(self.lib.svgf_resize(self._h, width, height), "svgf_resize")
            self.strip = (0, height, 0, height)
        else:
            sc = StripC(*[int(v) for v in strip])
            self._check(self.lib.svgf_resize_strip(self._h, width, height, C.byref(sc)), "svgf_resize_strip")
            self.strip = tuple(int(v) for v in strip)
        self.W, self.H = width, height

    def size(self):
        w, h, st = C.c_int(), C.c_int(), StripC()
        self._check(self.lib.svgf_get_size(self._h, C.byref(w), C.byref(h), C.byref(st)), "svgf_get_size")
        return w.value, h.value, (st.y0, st.rows, st.own_begin, st.own_end)

    def set_debug_mode(self, mode="final"):
        """SVGFDebugOutput sequences of application::Render (src/App.cu:545-649): 'final', 'temporal', 'atrous'."""
        self._check(self.lib.svgf_set_debug_mode(self._h, DEBUG_MODE[mode]), "svgf_set_debug_mode")

    def set_prev_guide(self, enable=True):
        """Whether the reprojection test may read the guide plane kept from the previous frame instead of `prev` (svgf.h)."""
        self._check(self.lib.svgf_set_prev_guide(self._h, 1 if enable else 0), "svgf_set_prev_guide")

    def set_adaptive_moments(self, enable=True):
        """The frame driver's choice between the young-pixel launch and the LDS-streaming kernel by a sample of recent frames' young pixels
        (include/svgf.h; default on; same results either way)."""
        self._check(self.lib.svgf_set_adaptive_moments(self._h, 1 if enable else 0), "svgf_set_adaptive_moments")

    def adaptive_moments_state(self) -> bool:
        return bool(self.lib.svgf_adaptive_moments_state(self._h))

    def adaptive_moments_sample(self):
        """-> (estimated young pixels, estimated waves that hold some) of a recent frame, as the frame driver reads them."""
        px, wv = C.c_uint(), C.c_uint()
        self._check(self.lib.svgf_adaptive_moments_sample(self._h, C.byref(px), C.byref(wv)), "svgf_adaptive_moments_sample")
        return px.value, wv.value

    def ImportPitched(self, plane, src_ptr, pitch_bytes, dst):
        self._check(self.lib.svgf_import_gbuffer_pitched(self._h, plane, C.c_void_p(src_ptr), pitch_bytes, _ptr(dst)), "svgf_import_gbuffer_pitched")

    def ImportArray(self, plane, hip_array, dst):
        self._check(self.lib.svgf_import_gbuffer_array(self._h, plane, C.c_void_p(hip_array), _ptr(dst)), "svgf_import_gbuffer_array")

    def ExportToArray(self, plane_tensor, hip_array):
        self._check(self.lib.svgf_export_to_array(self._h, _ptr(plane_tensor), C.c_void_p(hip_array)), "svgf_export_to_array")

    # -- the three stages (kernel-level API) -------------------------------------------------
    def TemporalFilter(self, prev_colour, radiance, colour_out, gb_cur: GBuffer, gb_prev: GBuffer, hist_prev, hist_cur,
                       moments_cur, moments_prev):
        """application::TemporalFilter, src/App.cu:469-478."""
        self._check(self.lib.svgf_temporal(self._h, _ptr(prev_colour), _ptr(radiance), _ptr(colour_out), gb_cur.c,
                                           gb_prev.c, _ptr(hist_prev), _ptr(hist_cur), _ptr(moments_cur),
                                           _ptr(moments_prev)), "svgf_temporal")

    def TemporalMoments(self, prev_colour, radiance, colour_out, filter_out, gb_cur: GBuffer, gb_prev: GBuffer, hist_prev, hist_cur,
                        moments_cur, moments_prev, moments_rows=(-1, -1), feedback_follows=False):
        """TemporalFilter + FilterMoments fused as in svgf_denoise_frame (src/App.cu:552-554), on caller-owned planes."""
        self._check(self.lib.svgf_temporal_moments(self._h, _ptr(prev_colour), _ptr(radiance), _ptr(colour_out), _ptr(filter_out),
                                                   gb_cur.c, gb_prev.c, _ptr(hist_prev), _ptr(hist_cur), _ptr(moments_cur),
                                                   _ptr(moments_prev), int(moments_rows[0]), int(moments_rows[1]), int(bool(feedback_follows))), "svgf_temporal_moments")

    def FilterMoments(self, colour, out, moments, gb: GBuffer, hist):
        """application::FilterMoments, src/App.cu:480-489."""
        self._check(self.lib.svgf_moments(self._h, _ptr(colour), _ptr(out), _ptr(moments), gb.c, _ptr(hist)), "svgf_moments")

    def FilterKernel(self, src, dst, feedback, gb: GBuffer, step: int, iteration: int):
        """One filter::FilterKernel launch (src/App.cu:504-505)."""
        self._check(self.lib.svgf_atrous(self._h, _ptr(src), _ptr(dst), _ptr(feedback), gb.c, step, iteration), "svgf_atrous")

    def FilterKernelPair(self, src, dst, feedback, gb: GBuffer):
        """The first two filter::FilterKernel launches of application::WaveletFilter (steps 1 and 2, src/App.cu:497-507) as one."""
        self._check(self.lib.svgf_atrous_pair(self._h, _ptr(src), _ptr(dst), _ptr(feedback), gb.c), "svgf_atrous_pair")

    def set_iteration_fusion(self, enable=True):
        """Whether Render / the strip driver run iterations 0 and 1 as one launch (svgf_atrous_pair: bit-identical, measured ~10 % slower)
        or one launch per iteration (the default)."""
        self._check(self.lib.svgf_set_iteration_fusion(self._h, 1 if enable else 0), "svgf_set_iteration_fusion")

    def WaveletFilter(self, filter_buffers, render_buffer, gb: GBuffer, steps=None):
        """application::WaveletFilter, src/App.cu:491-514: ping-pongs filter_buffers[0/1], iteration 0 feeds
        render_buffer.  Returns the buffer holding the result (no odd-N copy)."""
        steps = self.params.steps if steps is None else steps
        pp = 0
        for i in range(steps):
            self.FilterKernel(filter_buffers[pp], filter_buffers[1 - pp], render_buffer, gb, 1 << i, i)
            pp ^= 1
        return filter_buffers[pp]

    def TAA(self, filtered, history, out):
        """application::TAA, src/App.cu:516-522 (history = the previous call's out)."""
        self._check(self.lib.svgf_taa(self._h, _ptr(filtered), _ptr(history), _ptr(out)), "svgf_taa")

    def Demodulate(self, radiance, albedo, out):
        """radiance / max(albedo, 1e-3) — the SVGF paper's albedo demodulation, absent from the reference (README.md:14)."""
        self._check(self.lib.svgf_demodulate(self._h, _ptr(radiance), _ptr(albedo), _ptr(out)), "svgf_demodulate")

    def Modulate(self, filtered, albedo, out):
        """filtered * max(albedo, 1e-3)."""
        self._check(self.lib.svgf_modulate(self._h, _ptr(filtered), _ptr(albedo), _ptr(out)), "svgf_modulate")

    def PackGBuffer(self, position, normal, bary, view_proj, prev_view_proj, camera_position):
        """The G-buffer texels of resources/shaders/GBuffer.frag:62-88 from linear attribute planes; matrices are
        16 floats column-major (glm).  Returns a GBuffer of new device planes."""
        torch = self._torch
        cam = CameraC((C.c_float * 16)(*[float(v) for v in view_proj]), (C.c_float * 16)(*[float(v) for v in prev_view_proj]),
                      (C.c_float * 3)(*[float(v) for v in camera_position]))
        motion = torch.empty((self.rows, self.W, 4), dtype=torch.float32, device=self.device)
        nout = torch.empty((self.rows, self.W, 4), dtype=torch.int16, device=self.device)
        uvout = torch.empty((self.rows, self.W, 4), dtype=torch.int16, device=self.device)
        self._check(self.lib.svgf_pack_gbuffer(self._h, _ptr(position), _ptr(normal), _ptr(bary), C.byref(cam), _ptr(motion),
                                               _ptr(nout), _ptr(uvout)), "svgf_pack_gbuffer")
        return GBuffer(motion, nout, uvout)

    # -- whole frame on context-owned state --------------------------------------------------
    def Render(self, radiance, gb_cur: GBuffer, gb_prev: GBuffer | None = None):
        """The filter share of application::Render (src/App.cu:552-556).  Returns a tensor VIEW of the
        context-owned result plane (valid until the next call)."""
        out = C.c_void_p()
        self._check(self.lib.svgf_denoise_frame(self._h, _ptr(radiance), gb_cur.c, gb_prev.c if gb_prev else None,
                                                C.byref(out)), "svgf_denoise_frame")
        return self._wrap(out.value, (self.rows, self.W, 4), self.colour_dtype())

    def set_frames_in_flight(self, frames=2):
        """2: iterations 1.. of a frame run on a stream of the context's own beside the next frame's temporal launch; the view Render
        returned is ordered on the context's stream only by the next Render / flush / sync (include/svgf.h)."""
        self._check(self.lib.svgf_set_frames_in_flight(self._h, int(frames)), "svgf_set_frames_in_flight")

    def flush(self):
        """Order the frame in flight on the context's stream (no host wait)."""
        self._check(self.lib.svgf_flush(self._h), "svgf_flush")

    def reset_history(self):
        self._check(self.lib.svgf_reset_history(self._h), "svgf_reset_history")

    def pingpong(self) -> int:
        return self.lib.svgf_state_pingpong(self._h)

    def state_plane(self, plane: int, index: int):
        p = self.lib.svgf_state_plane(self._h, plane, index)
        if not p:
            return None
        if plane == PLANE_HISTORY:
            return self._wrap(p, (self.rows, self.W), self._torch.uint8)
        ch = 2 if plane == PLANE_MOMENTS else 4
        return self._wrap(p, (self.rows, self.W, ch), self.colour_dtype())

    def _wrap(self, ptr, shape, dtype):
        """Zero-copy tensor over library-owned device memory (through __cuda_array_interface__)."""
        torch = self._torch
        typestr = {torch.float32: "<f4", torch.float16: "<f2", torch.uint8: "|u1"}[dtype]

        class _Holder:
            pass
        hld = _Holder()
        hld.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3}
        hld.owner = self
        return torch.as_tensor(hld, device=self.device)

    # -- timing -------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        """on: False/0 = off, True/1 = stage events on every frame, n = on every n-th frame."""
        self._check(self.lib.svgf_timing_enable(self._h, int(on)), "svgf_timing_enable")

    def timing_read(self):
        """-> (list of per-stage summed ms [temporal, moments, atrous0..], frames)."""
        n = 2 + MAX_STEPS
        arr = (C.c_double * n)()
        fr = C.c_int()
        self._check(self.lib.svgf_timing_read(self._h, arr, C.byref(fr), n), "svgf_timing_read")
        return list(arr)[: 2 + self.params.steps], fr.value

    def path_stats_enable(self, on=True):
        """Diagnostics (include/svgf_ext.h): count, per a-trous step, the wave-steps that filtered a surface pixel and those on the uniform-normal path."""
        self._check(self.lib.svgf_path_stats_enable(self._h, int(bool(on))), "svgf_path_stats_enable")

    def path_stats_read(self):
        """-> {step: (wave-steps with a surface pixel, those of them on the uniform-normal tap path)} since the last read; synchronises."""
        arr = (C.c_ulonglong * (2 * PATH_STAT_STEPS))()
        self._check(self.lib.svgf_path_stats_read(self._h, arr, 2 * PATH_STAT_STEPS), "svgf_path_stats_read")
        return {1 << i: (int(arr[2 * i]), int(arr[2 * i + 1])) for i in range(PATH_STAT_STEPS)}
