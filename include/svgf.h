/* svgf.h — C ABI of libsvgf_mi355x.so: the SVGF denoiser hot path of jacquespillet/SVGF,
 * rebuilt as hand-written HIP kernels for AMD Instinct MI355X (gfx950).
 *
 * The reference has no plugin/FFI layer: the filter is three CUDA kernels in src/Filter.cuh launched
 * by three host methods of `application` (src/App.cu:469-514).  This header is the boundary those
 * call sites bind to instead; every entry point cites the reference interface it replaces.
 * INTEGRATION.md shows the reference-side patch.  Three headers, one library:
 *     svgf.h       what a host of the reference binds: lifecycle, the three stages, the frame driver, the stages either side
 *                  of the path (TAA, G-buffer adapters, albedo), the multi-GPU strip driver's product calls;
 *     svgf_ext.h   opt-ins and diagnostics (strip contexts and row ranges, fused stage calls, frames in flight, stream capture,
 *                  svgf_set_prev_guide, adaptive moments, per-stage timing, tap-path statistics, the strip driver's switches);
 *     svgf_test.h  test transports and fault injection of the strip driver.
 *
 * Conventions
 *  - All plane pointers are DEVICE pointers to tight row-major planes (index = y*W + x,
 *    Filter.cuh:60,439,536).  G-buffer planes replace the reference's CUDA texture objects
 *    (App.h:41-44, CudaUtil.h:68-99) with the same texel formats:
 *        motion  float[4]   {mv.x, mv.y, depth, ddepth}      (GBuffer.frag:67-71,81-82; App.cu:751)
 *        normal  uint16[4]  IEEE-half bits {nx, ny, nz, matID}       (GBuffer.frag:65,78,85; App.cu:749)
 *        uv      uint16[4]  IEEE-half bits {b0, b1, b2, instanceID}  (GBuffer.frag:64,77,86; App.cu:750)
 *    Colour planes are {r,g,b,variance}, moments planes {E[L],E[L^2]}, in the storage type chosen at
 *    creation: SVGF_F16 = the reference's half4/half2 (Filter.cuh:15-16), SVGF_F32 = float4/float2.
 *    History planes are uint8 (Filter.cuh:359,400).
 *  - Calls enqueue work on the context's HIP stream and return without synchronising, like the
 *    reference's launches on the default stream (App.cu:471-505).  Errors are returned (0 = ok,
 *    negative = SVGF_ERR_*), never asserted (the reference asserts: App.cu:41-48).  A refused call launches nothing, leaves the
 *    context as it was and nothing pending in the HIP runtime (tests/test_gpu_errors.py).
 *  - A context is not thread-safe (like the reference's single render thread, App.cu:692-734): one host thread at a
 *    time per context; different contexts are independent, may share a device or live on different devices of one
 *    process (every entry point makes the context's device current and restores the caller's).
 *  - Results: the temporal stage, the accept / reject masks, the history and every copied texel are the reference's bits (NaN, +-inf
 *    and -0.0 texels included); the filtered values are within the stated fp32 tolerance of the reference's fp64-island arithmetic
 *    (tests/gpu_helpers.py:TOL).  What the reference's binary does with non-finite and out-of-range input, the sky sentinel,
 *    mesh_id_test and the sign of zero is written out in docs/behaviour.md.
 */
#ifndef SVGF_MI355X_H
#define SVGF_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVGF_ABI_VERSION 8

enum svgf_status {
    SVGF_OK = 0,
    SVGF_ERR_INVALID = -1,    /* bad argument (null plane, bad size, bad row range, step < 1 ...)   */
    SVGF_ERR_HIP = -2,        /* a HIP runtime call failed; see svgf_last_error()                   */
    SVGF_ERR_NO_DEVICE = -3,  /* no usable gfx950 device                                            */
    SVGF_ERR_HALO = -4,       /* requested rows need taps outside the rows this strip holds, or (svgf_sync / svgf_strips_sync)
                                 a strip's temporal stage reprojected into rows it does not hold                         */
    SVGF_ERR_ALLOC = -5,
    SVGF_ERR_COMM = -6        /* an RCCL call failed, or librccl could not be opened                */
};

enum svgf_storage { SVGF_F32 = 0, SVGF_F16 = 1 };

/* Which a-trous kernel to run (all give the same results; for A/B measurement). */
enum svgf_variant { SVGF_VARIANT_AUTO = 0, SVGF_VARIANT_DIRECT = 1, SVGF_VARIANT_LDS = 2,
                    SVGF_VARIANT_LDS_GENERAL = 3 };   /* LDS kernels without the uniform-normal fast path (same results) */

typedef struct svgf_ctx svgf_ctx;

/* One G-buffer = the four texture objects of `cudaFramebuffer` (App.h:41-44) minus Position,
 * which the filter never reads. */
typedef struct svgf_gbuffer {
    const void* motion;
    const void* normal;
    const void* uv;
} svgf_gbuffer;

/* Tunables = application members src/App.h:109-114 (GUI ranges src/GUI.cpp:988-993). */
typedef struct svgf_params {
    int   steps;             /* SpatialFilterSteps, default 3; à-trous step of iteration i is 1<<i (App.cu:502) */
    float depth_threshold;   /* DepthThreshold  0.8  */
    float normal_threshold;  /* NormalThreshold 0.9  */
    int   history_base;      /* HistoryLength   24, clamped to [1,255] (uint8 history)        */
    float phi_colour;        /* PhiColour       10   */
    float phi_normal;        /* PhiNormal       128  */
    int   moments_radius;    /* 3 = reference (Filter.cuh:465); 1 = 3x3 variant                                */
    int   storage;           /* svgf_storage                                                                   */
    int   mesh_id_test;      /* 1 = compare instance IDs as Filter.cuh:245-247 intends, 0 = the de-facto no-op */
    int   variant;           /* svgf_variant                                                                   */
    int   nan_policy;        /* svgf_nan_policy: SVGF_NAN_REFERENCE (default) or SVGF_NAN_ZERO                 */
} svgf_params;
/* SVGF_NAN_REFERENCE: a NaN texel stays NaN, as in the reference (Filter.cuh:63-83).  SVGF_NAN_ZERO (an extension): the temporal stage
 * reads a NaN channel of the radiance / previous colour / previous moments as 0.  Same bits for finite input.  docs/behaviour.md. */
enum svgf_nan_policy { SVGF_NAN_REFERENCE = 0, SVGF_NAN_ZERO = 1 };

/* Rows of the global frame a context's planes hold, and the rows it owns (multi-GPU row strips; a whole-frame context: 0, H, 0, H). */
typedef struct svgf_strip {
    int y0;          /* global row stored at local row 0         */
    int rows;        /* local rows in every plane                */
    int own_begin;   /* first global row this context computes   */
    int own_end;     /* one past the last                        */
} svgf_strip;

void        svgf_default_params(svgf_params* p);                      /* App.h:109-114 defaults, SVGF_F16 */
const char* svgf_status_string(int status);
const char* svgf_last_error(const svgf_ctx* ctx);                      /* text of the last failure on ctx */
int         svgf_abi_version(void);

/* Lifecycle = application::ResizeRenderTextures (App.cu:742-778) for the filter's share of it.
 * `hip_stream` is a hipStream_t (NULL = the null stream).  State planes are allocated and zeroed
 * lazily by the first svgf_denoise_frame (App. B #9, #10: zero-init, exact size). */
int  svgf_create(svgf_ctx** out, int width, int height, const svgf_params* params, int device, void* hip_stream);
void svgf_destroy(svgf_ctx* ctx);
int  svgf_set_params(svgf_ctx* ctx, const svgf_params* params);       /* storage must not change */
int  svgf_set_stream(svgf_ctx* ctx, void* hip_stream);
/* New render size: synchronises the stream, frees every state plane; the next svgf_denoise_frame allocates them again (exact size,
 * zeroed: ResetRender, App.cu:777).  Tunables, stream, device, debug mode and timing settings stay. */
int  svgf_resize(svgf_ctx* ctx, int width, int height);
int  svgf_get_size(const svgf_ctx* ctx, int* width, int* height, svgf_strip* strip);     /* any pointer may be NULL */
int  svgf_sync(svgf_ctx* ctx);                                         /* wait for the context's stream */

/* Stage 1 — replaces application::TemporalFilter (App.cu:469-478) launching filter::TemporalFilter
 * (Filter.cuh:359-404, LoadPreviousData :225-258).  `radiance` (1-spp input, clamped on load) and
 * `colour_out` may alias, which is the reference's in-place CurrentImage.  History is ping-ponged
 * (hist_prev read at the reprojected pixel, hist_cur written): the reference's single buffer is a
 * data race for non-zero motion (App. B #1). */
int svgf_temporal(svgf_ctx* ctx, const void* prev_colour, const void* radiance, void* colour_out,
                  const svgf_gbuffer* cur, const svgf_gbuffer* prev,
                  const uint8_t* hist_prev, uint8_t* hist_cur, void* moments_cur, const void* moments_prev);

/* Stage 2 — replaces application::FilterMoments (App.cu:480-489) launching filter::FilterMoments
 * (Filter.cuh:430-525). */
int svgf_moments(svgf_ctx* ctx, const void* colour, void* out, const void* moments,
                 const svgf_gbuffer* gbuf, const uint8_t* hist);

/* Stage 3, one iteration — replaces one trip of the loop in application::WaveletFilter
 * (App.cu:497-507) launching filter::FilterKernel (Filter.cuh:527-624).  `feedback` is RenderOutput:
 * written (non-sky pixels only) iff iteration == 0 and it is non-null. */
int svgf_atrous(svgf_ctx* ctx, const void* in, void* out, void* feedback, const svgf_gbuffer* gbuf,
                int step, int iteration);

/* Whole frame — replaces the sequence application::Render runs (App.cu:552-556) on context-owned
 * state (RenderBuffer[2], MomentsBuffer[2], FilterBuffer[2], history; App.h:138-141).
 * `prev` may be NULL on the first frame.  *result receives the device pointer of the final
 * colour+variance plane (valid until the next call; no odd-N copy, App. B #12). */
int svgf_denoise_frame(svgf_ctx* ctx, const void* radiance, const svgf_gbuffer* cur,
                       const svgf_gbuffer* prev, const void** result);
int svgf_reset_history(svgf_ctx* ctx);                                 /* zero all state planes (ResetRender) */
/* The sequences application::Render runs in its debug views (SVGFDebugOutput, App.cu:545-649) on the same state:
 *   SVGF_DEBUG_FINAL     TemporalFilter, FilterMoments, WaveletFilter (App.cu:552-556)                  — the default
 *   SVGF_DEBUG_TEMPORAL  TemporalFilter only; *result = the temporally accumulated colour (App.cu:602-609)
 *   SVGF_DEBUG_ATROUS    TemporalFilter, then WaveletFilter WITHOUT FilterMoments (App.cu:611-620; also the Depth view,
 *                        :632-638): the filter's input is whatever FilterBuffer[0] holds — the previous frame's result
 *                         — and iteration 0 still feeds RenderBuffer back. */
enum svgf_debug_mode { SVGF_DEBUG_FINAL = 0, SVGF_DEBUG_TEMPORAL = 1, SVGF_DEBUG_ATROUS = 2 };
int svgf_set_debug_mode(svgf_ctx* ctx, int mode);

/* Debug taps / state access (the reference's SVGFDebugOutput modes read these, App.cu:567-649). */
enum svgf_plane { SVGF_PLANE_COLOUR = 0, SVGF_PLANE_MOMENTS = 1, SVGF_PLANE_FILTER = 2, SVGF_PLANE_HISTORY = 3 };
void* svgf_state_plane(svgf_ctx* ctx, int plane, int index);           /* index 0/1; NULL before first frame */
int   svgf_state_pingpong(const svgf_ctx* ctx);                        /* PingPongInx (App.cu:374)           */
size_t svgf_plane_bytes(const svgf_ctx* ctx, int plane);

/* The stage after the path — replaces application::TAA (App.cu:516-522) launching filter::TAAFilterKernel
 * (Filter.cuh:288-357): neighbourhood-clamped temporal anti-aliasing in PAL-YUV + linear->sRGB.  `history` is the
 * previous call's `out` (a separate plane: the reference reads it from the buffer it is writing, a race). */
int svgf_taa(svgf_ctx* ctx, const void* filtered, const void* history, void* out);

/* The stage in front of the path — the G-buffer texels resources/shaders/GBuffer.frag:62-88 writes, computed from
 * linear attribute planes (for producers that are not the reference's OpenGL rasteriser):
 *   position float[4] {world x,y,z, primitive id}   = OutPosition   (GBuffer.frag:63,80)
 *   normal   float[4] {world normal (any length), material id}      (GBuffer.frag:62,79)
 *   bary     float[4] {b0,b1,b2, instance id}                       (GBuffer.frag:61,78)
 * and the camera of application::Rasterize (App.cu:396-398); geometry static between the two frames.  motion = (prev - cur) NDC *
 * 0.5 * (W,H), depth = |camera - position|, ddepth = max(|dFdx|,|dFdy|) by 2x2-quad differences of depth.  A texel whose normal is
 * (0,0,0) has no geometry and is written as the cleared texel (all zero = sky for the filter). */
typedef struct svgf_camera {
    float view_proj[16];       /* column-major, Projection * inverse(Frame)          */
    float prev_view_proj[16];  /* column-major, Projection * inverse(PreviousFrame)  */
    float position[3];         /* Frame * (0,0,0,1)                                  */
} svgf_camera;
int svgf_pack_gbuffer(svgf_ctx* ctx, const void* position, const void* normal, const void* bary, const svgf_camera* camera,
                      void* motion_out, void* normal_out, void* uv_out);

/* Texture / pitched adapters — what the reference gets from its CUDA <-> OpenGL mappings (CreateMapping, CudaUtil.h:68-99;
 * Framebuffer.cpp:7-49): device-to-device copies on the context's stream between array-backed / pitched surfaces and the tight
 * planes the filter reads; the filtered plane goes back into the display texture's array (cudaMemcpyToArray, App.cu:561). */
enum svgf_gbuffer_plane { SVGF_GBUF_MOTION = 0, SVGF_GBUF_NORMAL = 1, SVGF_GBUF_UV = 2 };
int svgf_import_gbuffer_pitched(svgf_ctx* ctx, int plane, const void* src, size_t src_pitch_bytes, void* dst);
int svgf_import_gbuffer_array(svgf_ctx* ctx, int plane, const void* hip_array /* hipArray_const_t */, void* dst);
int svgf_export_to_array(svgf_ctx* ctx, const void* plane_data, void* hip_array /* hipArray_t */);

/* Albedo demodulation / re-modulation around the filter (absent in the reference, README.md:14,172-174):
 *   svgf_demodulate: out.rgb = radiance.rgb / max(albedo.rgb, 1e-3)   (before svgf_temporal)
 *   svgf_modulate:   out.rgb = filtered.rgb * max(albedo.rgb, 1e-3)   (after the last svgf_atrous);   out.w = in.w; `out` may alias `in`. */
int svgf_demodulate(svgf_ctx* ctx, const void* radiance, const void* albedo, void* out);
int svgf_modulate(svgf_ctx* ctx, const void* filtered, const void* albedo, void* out);

/* ---- Multi-GPU: row strips with RCCL halo exchange (the reference is single-GPU) ------------------------
 * The frame is cut into `world` contiguous row strips, one per GPU, and application::Render's filter sequence
 * (App.cu:552-556) runs on every strip; rows a strip needs from its neighbours travel as RCCL send/recv groups over xGMI,
 * posted from a communication stream of the driver's own and tied to the filter stream by HIP events.  Results are
 * bit-identical to the single-GPU frame.  A driver holds the strips of the ranks of THIS process (one process per GPU, or
 * several devices per process).  librccl is opened at run time (the one already in the process, else ROCm's; SVGF_RCCL_LIBRARY
 * overrides).  svgf_amd/csrc/svgf_strip.hip. */
#define SVGF_MAX_STEPS 10                                              /* GUI range 0-10, GUI.cpp:988 */
/* Halo plans: PER_ITERATION = an exchange in front of every a-trous iteration (BASELINE.json configs[3]); GROUPED = iterations in groups,
 * one exchange per group, the group's later iterations recomputed on ghost rows; GHOST = no exchange between iterations (all ghost rows).
 * AUTO = GROUPED where its halo fits the strips (the fewest exchanges that still exchange between iterations; measured fastest on an
 * 8K/8 strip), else PER_ITERATION. */
enum svgf_halo_plan { SVGF_PLAN_AUTO = 0, SVGF_PLAN_GHOST = 1, SVGF_PLAN_GROUPED = 2, SVGF_PLAN_PER_ITERATION = 3 };
typedef struct svgf_strips svgf_strips;
typedef struct svgf_strip_layout {
    int plan;                             /* the plan in force (AUTO resolved, see above)                                     */
    svgf_strip strip;                     /* rows the rank's planes hold / own                                               */
    int ext_atrous[SVGF_MAX_STEPS];       /* rows beyond the owned ones iteration i is computed on                           */
    int ngroups, group_first[SVGF_MAX_STEPS], halo_group[SVGF_MAX_STEPS];   /* iteration groups and their input halos          */
    int ext_moments, ext_temporal;        /* the same for the moments and temporal stages                                    */
    int halo_state;                       /* previous-frame state rows needed beyond the owned ones (ext_temporal + motion_reach) */
    int halo_max;
} svgf_strip_layout;
/* Pure geometry: what rank `rank` of `world` holds and computes.  SVGF_ERR_HALO if the strips are shorter than the halo. */
int svgf_strips_plan(int width, int height, int rank, int world, int steps, int plan, int moments_radius, int motion_reach,
                     svgf_strip_layout* out);
/* Bootstrap helpers around ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy (id128: 128 bytes, produced on one rank and
 * handed to the others by whatever the host has — MPI, a socket, torch.distributed). */
int svgf_rccl_unique_id(void* id128);
int svgf_rccl_comm_init(void** comm, int world, int rank, const void* id128, int device);
int svgf_rccl_comm_destroy(void* comm);
int svgf_rccl_comm_count(void* comm, int* count);              /* ncclCommCount: the ranks RCCL itself reports for the communicator */
/* How the ranks of a driver reach their neighbours: SVGF_TRANSPORT_RCCL = ncclSend / ncclRecv to the neighbour's rank of comms[k], the
 * product transport.  (Values 1 and 2 are the test transports of svgf_test.h.) */
enum svgf_strip_transport { SVGF_TRANSPORT_RCCL = 0 };
/* ranks / devices / compute_streams (hipStream_t, NULL entries = the null stream) / comms (ncclComm_t; loop-back: comms[0] only;
 * may be NULL when world == 1 or with the mailbox) describe the nlocal ranks of this process.  motion_reach = the largest |mv.y| (rows) the
 * temporal reprojection may need beyond what a strip computes itself; exceeding it is reported by svgf_strips_sync. */
int svgf_strips_create(svgf_strips** out, int width, int height, int world, const svgf_params* params, int plan, int motion_reach,
                       int nlocal, const int* ranks, const int* devices, void* const* compute_streams, void* const* comms, int transport);
void svgf_strips_destroy(svgf_strips* s);
const char* svgf_strips_last_error(const svgf_strips* s);
svgf_ctx* svgf_strips_context(svgf_strips* s, int local_index);            /* the strip's context (state planes, svgf_get_size ...);
                                                                             owned by the driver: svgf_destroy on it does nothing        */
int svgf_strips_layout(const svgf_strips* s, int local_index, svgf_strip_layout* out);
/* One frame on every local strip.  radiance[k], cur[k], prev[k] are device planes of local rank k holding ITS rows
 * [strip.y0, strip.y0 + strip.rows) (prev may be NULL, or prev[k].motion NULL, on the first frame); results[k] receives the
 * plane whose OWNED rows hold the result.  Enqueues and returns. */
int svgf_strips_frame(svgf_strips* s, const void* const* radiance, const svgf_gbuffer* cur, const svgf_gbuffer* prev, const void** results);
/* Wait for the last frame's state exchange and the streams; SVGF_ERR_HALO if a reprojection left a strip (see svgf_sync). */
int svgf_strips_sync(svgf_strips* s);

#ifdef __cplusplus
}
#endif
#endif /* SVGF_MI355X_H */
