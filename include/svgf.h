/* svgf.h — C ABI of libsvgf_mi355x.so: the SVGF denoiser hot path of jacquespillet/SVGF,
 * rebuilt as hand-written HIP kernels for AMD Instinct MI355X (gfx950).
 *
 * The reference has no plugin/FFI layer: the filter is three CUDA kernels in src/Filter.cuh launched
 * by three host methods of `application` (src/App.cu:469-514).  This header is the boundary those
 * call sites bind to instead; every entry point cites the reference interface it replaces.
 * INTEGRATION.md shows the reference-side patch.
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
 *    negative = SVGF_ERR_*), never asserted (the reference asserts: App.cu:41-48).
 *  - A context is not thread-safe (like the reference's single render thread, App.cu:692-734): one host thread at a
 *    time per context; different contexts are independent and may share a device.  Resizing = a new context
 *    (ResizeRenderTextures frees and reallocates everything, App.cu:742-778).
 *  - Strips: a context may hold only rows [y0, y0+rows) of a WxH frame (multi-GPU row strips);
 *    "inside the frame" tests always use the global frame, so strip results are bit-identical
 *    to the whole-frame result as long as the halo rows hold valid data.
 */
#ifndef SVGF_MI355X_H
#define SVGF_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVGF_ABI_VERSION 1

enum svgf_status {
    SVGF_OK = 0,
    SVGF_ERR_INVALID = -1,    /* bad argument (null plane, bad size, bad row range, step < 1 ...)   */
    SVGF_ERR_HIP = -2,        /* a HIP runtime call failed; see svgf_last_error()                   */
    SVGF_ERR_NO_DEVICE = -3,  /* no usable gfx950 device                                            */
    SVGF_ERR_HALO = -4,       /* requested rows need taps outside the rows this strip holds         */
    SVGF_ERR_ALLOC = -5
};

enum svgf_storage { SVGF_F32 = 0, SVGF_F16 = 1 };

/* Which à-trous kernel to run (all give the same results; for A/B measurement). */
enum svgf_variant { SVGF_VARIANT_AUTO = 0, SVGF_VARIANT_DIRECT = 1, SVGF_VARIANT_LDS = 2 };

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
} svgf_params;

/* Rows of the global frame this context's planes hold, and the rows it owns (computes by default). */
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
int  svgf_create_strip(svgf_ctx** out, int width, int height, const svgf_strip* strip,
                       const svgf_params* params, int device, void* hip_stream);
void svgf_destroy(svgf_ctx* ctx);
int  svgf_set_params(svgf_ctx* ctx, const svgf_params* params);       /* storage must not change */
int  svgf_set_stream(svgf_ctx* ctx, void* hip_stream);
/* Restrict the following stage calls to global rows [row_begin,row_end) (interior/boundary split of
 * a strip); (-1,-1) restores the owned rows. */
int  svgf_set_rows(svgf_ctx* ctx, int row_begin, int row_end);

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

/* Stages 1 + 2 fused, for hosts that own their planes (the strip runner): what svgf_denoise_frame does internally.
 * The temporal launch also stores its result into `filter_out` — where history >= 4 FilterMoments is a copy
 * (Filter.cuh:521) — and the moments launch then only re-filters the young pixels (history < 4) of global rows
 * [moments_row_begin, moments_row_end) (a sub-range of the rows set by svgf_set_rows; -1,-1 = those rows).  Same
 * results as svgf_temporal + svgf_moments on those rows, 32 B/px (fp32) less traffic in steady state.
 * feedback_follows != 0: the caller will run svgf_atrous iteration 0 with feedback = colour_out over every row of
 * colour_out it goes on to use; then a texel that feedback overwrites (history >= 4, depth != 0: Filter.cuh:619-622) is
 * not stored into colour_out at all by this call (another 16 B/px), only into filter_out. */
int svgf_temporal_moments(svgf_ctx* ctx, const void* prev_colour, const void* radiance, void* colour_out, void* filter_out,
                          const svgf_gbuffer* cur, const svgf_gbuffer* prev, const uint8_t* hist_prev, uint8_t* hist_cur,
                          void* moments_cur, const void* moments_prev, int moments_row_begin, int moments_row_end,
                          int feedback_follows);

/* Stage 3, one iteration — replaces one trip of the loop in application::WaveletFilter
 * (App.cu:497-507) launching filter::FilterKernel (Filter.cuh:527-624).  `feedback` is RenderOutput:
 * written (non-sky pixels only) iff iteration == 0 and it is non-null. */
int svgf_atrous(svgf_ctx* ctx, const void* in, void* out, void* feedback, const svgf_gbuffer* gbuf,
                int step, int iteration);

/* The stage after the path — replaces application::TAA (App.cu:516-522) launching filter::TAAFilterKernel
 * (Filter.cuh:288-357): neighbourhood-clamped temporal anti-aliasing in PAL-YUV + linear->sRGB.  `history` is the
 * previous call's `out` (a separate plane: the reference reads it from the buffer it is writing, a race). */
int svgf_taa(svgf_ctx* ctx, const void* filtered, const void* history, void* out);

/* The stage in front of the path — the G-buffer texels resources/shaders/GBuffer.frag:62-88 writes, computed from
 * linear attribute planes (for producers that are not the reference's OpenGL rasteriser):
 *   position float[4] {world x,y,z, primitive id}   = OutPosition   (GBuffer.frag:63,80)
 *   normal   float[4] {world normal (any length), material id}      (GBuffer.frag:62,79)
 *   bary     float[4] {b0,b1,b2, instance id}                       (GBuffer.frag:61,78)
 * and the camera of application::Rasterize (App.cu:396-398).  Geometry is taken to be static between the two frames
 * (PreviousMVP * vertex == prev_view_proj * world position).  motion = (prev - cur) NDC * 0.5 * (W,H), depth =
 * |camera - position|, ddepth = max(|dFdx|,|dFdy|) by 2x2-quad differences of depth (0 towards a texel without
 * geometry; OpenGL extrapolates the triangle there, which no image-space adapter can).  A texel whose normal is
 * (0,0,0) has no geometry and is written as the cleared texel (all zero = sky for the filter). */
typedef struct svgf_camera {
    float view_proj[16];       /* column-major, Projection * inverse(Frame)          */
    float prev_view_proj[16];  /* column-major, Projection * inverse(PreviousFrame)  */
    float position[3];         /* Frame * (0,0,0,1)                                  */
} svgf_camera;
int svgf_pack_gbuffer(svgf_ctx* ctx, const void* position, const void* normal, const void* bary, const svgf_camera* camera,
                      void* motion_out, void* normal_out, void* uv_out);

/* Albedo demodulation / re-modulation around the filter.  The reference does not have it — "it's not doing
 * albedo demodulation as described in the paper, so it doesn't really work with textured meshes" (README.md:14,172-174) —
 * so there is no reference call site; the definition is the SVGF paper's:
 *   svgf_demodulate: out.rgb = radiance.rgb / max(albedo.rgb, 1e-3), out.w = radiance.w       (before svgf_temporal)
 *   svgf_modulate:   out.rgb = filtered.rgb * max(albedo.rgb, 1e-3), out.w = filtered.w       (after the last svgf_atrous)
 * `albedo` is a {r,g,b,-} plane in the context's storage type; `out` may alias the first argument.  Note that the
 * reference's imageLoad clamps colour to [0,1] (Filter.cuh:78-83): illumination above 1 is clipped by the temporal stage,
 * so a caller with bright lights over dark albedo should pre-scale its radiance. */
int svgf_demodulate(svgf_ctx* ctx, const void* radiance, const void* albedo, void* out);
int svgf_modulate(svgf_ctx* ctx, const void* filtered, const void* albedo, void* out);

/* Whole frame — replaces the sequence application::Render runs (App.cu:552-556) on context-owned
 * state (RenderBuffer[2], MomentsBuffer[2], FilterBuffer[2], history; App.h:138-141).
 * `prev` may be NULL on the first frame.  *result receives the device pointer of the final
 * colour+variance plane (valid until the next call; no odd-N copy, App. B #12). */
int svgf_denoise_frame(svgf_ctx* ctx, const void* radiance, const svgf_gbuffer* cur,
                       const svgf_gbuffer* prev, const void** result);
int svgf_reset_history(svgf_ctx* ctx);                                 /* zero all state planes (ResetRender) */

/* Debug taps / state access (the reference's SVGFDebugOutput modes read these, App.cu:567-649). */
enum svgf_plane { SVGF_PLANE_COLOUR = 0, SVGF_PLANE_MOMENTS = 1, SVGF_PLANE_FILTER = 2, SVGF_PLANE_HISTORY = 3 };
void* svgf_state_plane(svgf_ctx* ctx, int plane, int index);           /* index 0/1; NULL before first frame */
int   svgf_state_pingpong(const svgf_ctx* ctx);                        /* PingPongInx (App.cu:374)           */
size_t svgf_plane_bytes(const svgf_ctx* ctx, int plane);

/* Per-stage device timing with HIP events on the context's stream (the reference only prints whole
 * frame time, App.cu:727-731).  Slots: 0 temporal, 1 moments, 2+i à-trous iteration i. */
#define SVGF_MAX_STEPS 10                                              /* GUI range 0-10, GUI.cpp:988 */
int svgf_timing_enable(svgf_ctx* ctx, int on);                                  /* 0 = off, n = time every n-th frame (events cost ~1 us each) */
int svgf_timing_read(svgf_ctx* ctx, double* ms_sum, int* frames, int slots);   /* synchronises; resets sums */

#ifdef __cplusplus
}
#endif
#endif /* SVGF_MI355X_H */
