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
 *    negative = SVGF_ERR_*), never asserted (the reference asserts: App.cu:41-48).  A refused call launches nothing, leaves the
 *    context as it was and nothing pending in the HIP runtime (hipGetLastError is clean afterwards: a host or framework that checks it
 *    after its own launches does not trip over this library's refusals — tests/test_gpu_errors.py).
 *  - A context is not thread-safe (like the reference's single render thread, App.cu:692-734): one host thread at a
 *    time per context; different contexts are independent, may share a device or live on different devices of one
 *    process (every entry point makes the context's device current and restores the caller's).  svgf_resize is
 *    ResizeRenderTextures (App.cu:742-778): everything is freed and reallocated, the accumulation restarts.
 *  - mesh_id_test.  Filter.cuh:245-247 fetches the RGBA16UI barycentric/instance texture through tex2D<float4>: the
 *    half bits come back as denormal floats and int(...) of both sides is 0, so in the reference's BINARY the
 *    instance-ID test never rejects.  svgf_default_params sets mesh_id_test = 1, the comparison the
 *    source intends (decode the half, compare the IDs); a host that wants the reference's de-facto accept/reject mask sets
 *    mesh_id_test = 0.  Both are covered by the bit-exact temporal tests.
 *  - Sky.  A texel whose GetDepth() is the sentinel (depth 0, Filter.cuh:199-207 — a depth of literally 1e30f reads the
 *    same) is "sky": the wavelet filter copies it and skips its feedback store.  The kernels rely on a sky TAP having
 *    weight exactly 0, which holds while ddepth * step < ~1e22 (|1e30 - z| / phi_z overflows the exponent of exp to -inf
 *    or beyond -150): any real depth derivative.  The LDS kernels also copy a sky CENTRE through the taps (its depth enters
 *    them as -1e30, so that every one of its weights is exactly 0 and the normalisation multiplies its own colour by rcp(1)):
 *    the same bound, on the sky texel's own ddepth.
 *  - Non-finite input.  The reference's imageLoad / imageStore clamp is glm::clamp = min(max(x, 0), 1) built from `(x < y) ? y : x`
 *    (Filter.cuh:63-69,78-83): +inf clamps to 1, -inf to 0, and a NaN texel STAYS NaN.  It then poisons the history through mix (:398) for
 *    as long as the pixel keeps reprojecting onto it, reaches the wavelet sums channel by channel (the weight itself stays finite:
 *    `max(weightLillum, 0.0)` in :424 is CUDA's fmax, which drops the NaN; a sky centre is copied whatever its neighbours hold) and turns
 *    the zero-weight sums FilterMoments forms for zero-normal (sky) texels into NaN (0 x NaN, :498-499).  svgf_temporal, svgf_moments,
 *    svgf_atrous, svgf_atrous_pair, svgf_denoise_frame and the strip driver reproduce exactly that (tests/test_gpu_nonfinite.py: NaN
 *    masks identical to the oracle's, finite values within the stage tolerances): a host that wants its NaNs healed must clean the
 *    radiance before the temporal stage — the reference does not, and neither does this library.  svgf_taa likewise: a NaN texel goes
 *    through glm's min / max position by position (Filter.cuh:330-338) and the NaN test of :351 writes the pixel black.
 *  - Sign of zero.  The reference's value clamp, built from `(x < y) ? y : x`, passes -0.0 through (Filter.cuh:57-82); so do the kernels: the
 *    temporal stage, the copied sky texels of the wavelet filter and svgf_taa store the reference's bits, a filtered texel that comes out
 *    zero carries the reference's sign (tests: test_temporal_bit_exact, test_atrous, tests/fuzz_parity.py compare raw bits).  One exception:
 *    svgf_atrous_pair (opt-in) rounds the texels of a band that holds a -0.0 texel as it does next to a NaN — within the stage tolerance.
 *  - Non-finite / out-of-range G-buffer texels: what the reference's binary does, reproduced (tests/test_gpu_gbuffer_nonfinite.py):
 *        motion   `Coord + ivec2(MotionVector)` (Filter.cuh:232) is a float -> int conversion toward zero that SATURATES and turns a NaN
 *                 into 0, added to the pixel coordinate with wrap-around: a NaN motion reprojects the pixel onto itself; +-inf and anything
 *                 beyond +-2^31 pixels lands outside the frame and is rejected (:235).  The instance ID of mesh_id_test = 1 converts the same way.
 *        depth    only 0 is the sentinel (:204; -0 too): a negative or denormal depth is a number, a depth of exactly 1e30 reads like the
 *                 sentinel.  A NaN depth fails no test — `abs(dz) > DepthThreshold` is false, the reprojection is ACCEPTED (:242) — and drops
 *                 out of every weight it enters: `max(weightZ, 0.0)` is CUDA's fmax (:424).
 *        ddepth   `max(ddepth, 1e-6f)` / `max(ddepth, 1e-8)` are fmaxf / fmax (:563,461): a NaN or negative derivative gives the floor.
 *        normal   a NaN normal fails no test either (`dot < NormalThreshold` is false: ACCEPTED, :252) and has weight 0 in the filters
 *                 (saturate(NaN) = 0, :419); a zero-length normal is rejected and has weight 0.
 *    Not reproduced: a depth derivative of +inf (or beyond ~1e22 / step) on a sky texel — see "Sky" above.
 *  - Bit-identity next to a NaN.  The streaming a-trous kernel redoes, the reference's way, exactly the pixels whose fast result held a NaN;
 *    every other pixel keeps its bits, so strips and row ranges stay bit-identical to the whole frame with NaN texels present (colour or
 *    G-buffer).  Both moments kernels — the LDS-streaming one (the first three frames after a reset, crowded frames, svgf_moments) and the
 *    young-pixel launch of the drivers — follow the same rule, so which of them serves a frame, svgf_set_adaptive_moments and the stage
 *    calls against the frame driver change no bit either.  A texel WITHOUT depth that holds a normal (not the all-zero normal of a cleared
 *    texel) counts for the uniform-normal shortcut like a surface texel; a workgroup whose reference normal holds a NaN takes no shortcut.
 *    (tests/fuzz_parity.py sweeps sizes, tunables, partitions and poisoned texels for exactly these claims; tests/test_gpu_fuzz.py pins
 *    what it found.)  What still depends on the path taken, within the stated tolerance: the pair launch svgf_atrous_pair (an opt-in)
 *    takes the exact form for every pixel of a band that holds a NaN.
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

#define SVGF_ABI_VERSION 7

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

/* Which à-trous kernel to run (all give the same results; for A/B measurement). */
enum svgf_variant { SVGF_VARIANT_AUTO = 0, SVGF_VARIANT_DIRECT = 1, SVGF_VARIANT_LDS = 2,
                    SVGF_VARIANT_LDS_GENERAL = 3 };   /* the LDS kernels with the a-trous uniform-normal fast path switched off (same results): what
                                                       * geometry without planar regions costs; bench.py reports it next to the headline */

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

/* What the temporal stage does with a NaN in the radiance it is given or in the history it reprojects onto ("Non-finite input" above).
 *   SVGF_NAN_REFERENCE  what the reference does: the NaN stays (Filter.cuh:63-83), settles in the history, and — through iteration 0's
 *                       feedback — reaches two more pixels in every direction with every frame: a single NaN texel ends up covering every
 *                       connected surface (the reference has no protection; its path tracer avoids producing them, PathTrace.cuh:338).
 *   SVGF_NAN_ZERO       an extension: svgf_temporal / svgf_temporal_moments / svgf_denoise_frame / the strip driver read a NaN channel of
 *                       the radiance, of the previous colour and of the previous moments as 0, so that nothing behind the temporal stage
 *                       ever sees one.  With finite input the two policies give the same bits. */
enum svgf_nan_policy { SVGF_NAN_REFERENCE = 0, SVGF_NAN_ZERO = 1 };

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
/* application::ResizeRenderTextures (App.cu:742-778): new render size (svgf_resize: whole frame; svgf_resize_strip: a strip
 * of the new frame).  Synchronises the stream, frees every state plane; the next svgf_denoise_frame allocates them again
 * (exact size, zeroed: ResetRender, App.cu:777).  Tunables, stream, device, debug mode and timing settings stay. */
int  svgf_resize(svgf_ctx* ctx, int width, int height);
int  svgf_resize_strip(svgf_ctx* ctx, int width, int height, const svgf_strip* strip);
int  svgf_get_size(const svgf_ctx* ctx, int* width, int* height, svgf_strip* strip);     /* any pointer may be NULL */
/* Wait for the context's stream.  Returns SVGF_ERR_HALO if, since the last call, the temporal stage of a STRIP context
 * reprojected a pixel to a row inside the frame that the strip does not hold (motion larger than its state halo): such a
 * pixel was treated as a rejection, so the strip is no longer bit-identical to the whole frame.  svgf_halo_violations
 * returns the count (and zeroes it if clear != 0) without turning it into an error. */
int  svgf_sync(svgf_ctx* ctx);
int  svgf_halo_violations(svgf_ctx* ctx, unsigned long long* count, int clear);
/* Global rows [row_begin,row_end) of the previous-frame planes (colour, moments, history, previous G-buffer) that hold VALID
 * state; default (-1,-1) = every row the strip holds.  A strip whose planes are taller than the rows it keeps up to date (the
 * a-trous halos are wider than the state halo) declares the valid ones here: a reprojection beyond them counts as a halo
 * violation instead of silently reading stale rows.  The strip driver sets this itself. */
int  svgf_set_valid_rows(svgf_ctx* ctx, int row_begin, int row_end);

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

/* Stage 3, iterations 0 and 1 in ONE launch — the first two trips of the loop in application::WaveletFilter (App.cu:497-507:
 * steps 1 and 2, FilterBuffer[0] -> [1] -> [0]) without the plane in between: iteration 0's rows stay on the chip for iteration 1
 * and reach memory only as `feedback` (RenderOutput, Filter.cuh:619-622; may be NULL).  `out` receives what two svgf_atrous calls
 * would leave in their second `out`, bit for bit, on the rows set by svgf_set_rows; `feedback` is written on those rows and the
 * 4 rows beyond them inside the frame (iteration 1 reads iteration 0 there), so the planes must hold 6 rows around the launch
 * rows (SVGF_ERR_HALO otherwise).  `in`, `out` and `feedback` are three different planes.  Needs variant != SVGF_VARIANT_DIRECT
 * and PhiNormal != 0.  Measured on MI355X the pair launch is ~10 % SLOWER than the two launches it replaces (the iterations are
 * bound by their tap arithmetic, not by the 48 B/px the fusion saves: DESIGN.md 3.3c), so svgf_denoise_frame and the strip driver
 * use it only after svgf_set_iteration_fusion(ctx, 1) (default 0; same results either way; with steps >= 2; svgf_denoise_frame fuses on the
 * WHOLE frame only — with svgf_set_rows narrower than the frame the feedback rows beyond the range would be computed from rows this
 * frame's temporal launch did not write — and the strip driver where the halo plan keeps iterations 0 and 1 in one group). */
int svgf_atrous_pair(svgf_ctx* ctx, const void* in, void* out, void* feedback, const svgf_gbuffer* gbuf);
int svgf_set_iteration_fusion(svgf_ctx* ctx, int enable);

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
 * (`max` is fmaxf, as CUDA's max(float, float) is: a NaN albedo reads as the floor 1e-3; a NaN / inf colour goes through the division / product.)
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
/* Two frames in flight — a throughput mode the reference has no counterpart of (application::Render runs one frame at a time on
 * the default stream, App.cu:545-556).  A frame's temporal launch is HBM-bound and its wavelet iterations are bound by their tap
 * arithmetic; from iteration 0 on nothing a frame still does is read by the next frame's temporal launch (iteration 0 feeds the
 * history back, App.cu:504-505).  With svgf_set_frames_in_flight(ctx, 2), svgf_denoise_frame enqueues the temporal, moments and
 * iteration-0 launches on the context's stream and iterations 1.. on a stream of its own, where they run beside the NEXT frame's
 * temporal launch (measured: -1 to -5 % per 4K fp32 frame depending on the board, -7 % at 1080p; results bit-identical).  What changes for the caller:
 *   - *result of call f is returned at once but is ORDERED on the context's stream only by the next svgf_denoise_frame, svgf_flush
 *     or svgf_sync (enqueue the consumer of frame f after one of those); it stays valid until the call after the next one (frames
 *     alternate between two pairs of filter planes: +2 colour planes of memory);
 *   - the planes of `cur` are not read after the call's launches on the context's stream: iterations that read them (the direct kernel:
 *     variant DIRECT, PhiNormal == 0, a step beyond 64) keep the frame's tail on the context's stream — such a frame simply does not
 *     overlap with the next one;
 *   - the debug views (svgf_set_debug_mode) and strip-driver contexts do not combine with it (refused).
 * frames = 1 (default) restores stream order at once: the frame in flight is ordered on the context's stream by that call and its
 * result is then valid until the next svgf_denoise_frame, as ever.  svgf_flush orders the frame in flight on the context's stream
 * without waiting for it; svgf_reset_history / svgf_resize / svgf_destroy wait for or order it themselves. */
int svgf_set_frames_in_flight(svgf_ctx* ctx, int frames);
int svgf_flush(svgf_ctx* ctx);
/* Stream capture — a host that records its frame into a hipGraph (hipStreamBeginCapture on the context's stream) can record
 * svgf_denoise_frame and the stage calls with it: in steady state they only enqueue (kernel launches, two 4-byte memsets on an error
 * path, with two frames in flight the driver's own event record / wait pairs, which take the side stream into the capture and back).
 * What a graph replays is what the captured calls enqueued, so:
 *   - capture an EVEN number of svgf_denoise_frame calls: the context ping-pongs its state, guide and (two frames in flight) filter
 *     planes per frame, and the second call leaves it where the first one found it;
 *   - the planes passed to the captured calls (radiance, cur, prev, and whatever consumes *result) are the ones every replay reads and
 *     writes: the host refills them, normally by nodes of the same graph;
 *   - the first three frames after svgf_create / svgf_resize / svgf_reset_history cannot be captured (the first one allocates, all
 *     three run the cold-start moments kernel): under capture they are refused with SVGF_ERR_INVALID and record nothing — enqueue
 *     them directly; tunables, row ranges, debug mode and the switches are those in force at capture time, and so is the kernel that serves the
 *     young pixels (svgf_set_adaptive_moments: chosen per call from a sample of recent frames — a graph keeps the choice of the call it recorded);
 *   - with two frames in flight: svgf_flush before hipStreamBeginCapture (a frame enqueued before the capture cannot be joined inside
 *     it: refused) and again before hipStreamEndCapture (the side stream must be back on the captured one: HIP refuses to end a capture
 *     with unjoined work, and on ROCm 7.2 leaves its streams unusable afterwards);
 *   - per-stage timing skips captured frames; svgf_sync / svgf_halo_violations / svgf_timing_read wait for the device and are not
 *     capturable, as any synchronising call; the strip driver (svgf_strips_frame) is not capturable.
 * Replayed frames equal directly enqueued ones bit for bit (tests/test_gpu_graph.py).  Measured (tools/graph_replay.py): with one
 * frame in flight a replay costs the device what the calls cost (the launches are not host-bound: 7 us against 30 us of host time
 * per frame, no device time saved); with two frames in flight the cross-stream edges of a graph are cheaper than the event waits of
 * the calls: -5 % at 1080p and -9 % at 720p against one frame in flight enqueued call by call. */
/* The sequences application::Render runs in its debug views (SVGFDebugOutput, App.cu:545-649) on the same state:
 *   SVGF_DEBUG_FINAL     TemporalFilter, FilterMoments, WaveletFilter (App.cu:552-556)                  — the default
 *   SVGF_DEBUG_TEMPORAL  TemporalFilter only; *result = the temporally accumulated colour (App.cu:602-609)
 *   SVGF_DEBUG_ATROUS    TemporalFilter, then WaveletFilter WITHOUT FilterMoments (App.cu:611-620; also the Depth view,
 *                        :632-638): the filter's input is whatever FilterBuffer[0] holds — the previous frame's result
 *                         — and iteration 0 still feeds RenderBuffer back. */
enum svgf_debug_mode { SVGF_DEBUG_FINAL = 0, SVGF_DEBUG_TEMPORAL = 1, SVGF_DEBUG_ATROUS = 2 };
int svgf_set_debug_mode(svgf_ctx* ctx, int mode);
/* The frame and strip drivers keep, of every frame's current G-buffer, the 16 bytes per pixel the filter reads of it ({depth,
 * ddepth, normal, instance ID}: the "guide" plane).  After svgf_set_prev_guide(ctx, 1), when the next frame's `prev` is that very
 * G-buffer — the same three plane addresses, and not the new frame's `cur` — its reprojection test (LoadPreviousData,
 * Filter.cuh:225-258) reads the kept plane instead of the three planes of `prev` (16 instead of 32 B per pixel: -3 % of a 4K fp32
 * frame, -6 % with fp16 storage; bit-identical results).
 * PRECONDITION the host vouches for by enabling it: the planes of `prev` still hold what they held when they were passed as `cur`
 * — true of the reference, where Framebuffer[1 - PingPongInx] is not written between the two frames (App.cu:374,545-556); NOT
 * true of a host that re-renders into those addresses without running the denoiser on that frame (it would be tested against a
 * stale depth / normal / ID, silently).  Default 0: `prev` is read as it is.  Any `prev` at other addresses is read as it is. */
int svgf_set_prev_guide(svgf_ctx* ctx, int enable);
/* Which kernel serves a frame's young pixels (history < 4: FilterMoments' 7x7 estimate, Filter.cuh:444-516) is the frame driver's choice: a launch
 * over the young pixels alone (what they cost: 0.005 ms per 4K frame for none, 0.03 under a pan, 0.5 for 12 % of the frame, 1.6 for half of it) or the
 * LDS-streaming kernel over every pixel (0.21 ms whatever is young; always for the first three frames after a reset).  With enable = 1 (default)
 * the driver goes by a sample of the young pixels of a recent frame, which the temporal launch leaves in host-mapped memory (no synchronisation: it
 * is a few frames old): above 8 % of the frame — or with more waves holding young pixels than the young-pixel list takes appends from (a quarter of the frame's waves: 32 400 at 4K) — the
 * streaming kernel, back below 5 % (and three quarters of that).  Both evaluate the estimate on the same bits, so the choice
 * never shows in the results (finite input; around a NaN texel the two round the luminance term differently, both within the stated tolerance).
 * enable = 0: the young-pixel launch whenever the frame is not one of the first three.  The strip driver's contexts (svgf_strips_context) choose the
 * same way, every rank for itself: the results do not depend on it. */
int svgf_set_adaptive_moments(svgf_ctx* ctx, int enable);
int svgf_adaptive_moments_state(const svgf_ctx* ctx);                  /* 1: the last frame was served by the streaming kernel because of the sample */
/* the latest sample as the driver reads it (x 64: an estimate of a recent frame's young pixels and of its waves that hold some); SVGF_ERR_INVALID before the first frame */
int svgf_adaptive_moments_sample(const svgf_ctx* ctx, unsigned* young_pixels, unsigned* appending_waves);

/* Texture / pitched adapters — what the reference gets from its CUDA <-> OpenGL mappings (CreateMapping, CudaUtil.h:68-99;
 * render targets Framebuffer.cpp:7-49): the G-buffer planes arrive as array-backed textures or pitched surfaces and are
 * copied (device to device, on the context's stream) into the tight linear planes the filter reads; the filtered plane
 * goes back into the display texture's array (cudaMemcpyToArray, App.cu:561).  `dst` / `plane_data` hold the context's
 * rows [y0, y0+rows); an array holds the whole frame.  Texel formats as in svgf_gbuffer. */
enum svgf_gbuffer_plane { SVGF_GBUF_MOTION = 0, SVGF_GBUF_NORMAL = 1, SVGF_GBUF_UV = 2 };
int svgf_import_gbuffer_pitched(svgf_ctx* ctx, int plane, const void* src, size_t src_pitch_bytes, void* dst);
int svgf_import_gbuffer_array(svgf_ctx* ctx, int plane, const void* hip_array /* hipArray_const_t */, void* dst);
int svgf_export_to_array(svgf_ctx* ctx, const void* plane_data, void* hip_array /* hipArray_t */);

/* Debug taps / state access (the reference's SVGFDebugOutput modes read these, App.cu:567-649). */
enum svgf_plane { SVGF_PLANE_COLOUR = 0, SVGF_PLANE_MOMENTS = 1, SVGF_PLANE_FILTER = 2, SVGF_PLANE_HISTORY = 3 };
void* svgf_state_plane(svgf_ctx* ctx, int plane, int index);           /* index 0/1; NULL before first frame */
int   svgf_state_pingpong(const svgf_ctx* ctx);                        /* PingPongInx (App.cu:374)           */
size_t svgf_plane_bytes(const svgf_ctx* ctx, int plane);

/* Per-stage device timing with HIP events on the context's stream (the reference only prints whole
 * frame time, App.cu:727-731).  Slots: 0 temporal, 1 moments, 2+i à-trous iteration i (when iterations 0 and 1 run as one
 * launch, slot 2 holds the pair and slot 3 the ~1 us between two events). */
#define SVGF_MAX_STEPS 10                                              /* GUI range 0-10, GUI.cpp:988 */
int svgf_timing_enable(svgf_ctx* ctx, int on);                                  /* 0 = off, n = time every n-th frame (events cost ~1 us each) */
int svgf_timing_read(svgf_ctx* ctx, double* ms_sum, int* frames, int slots);   /* synchronises; resets sums */

/* ---- Multi-GPU: row strips with RCCL halo exchange (the reference is single-GPU) ------------------------
 * The frame is cut into `world` contiguous row strips, one per GPU, and application::Render's filter sequence
 * (App.cu:552-556) runs on every strip; rows a strip needs from its neighbours travel as RCCL send/recv groups over xGMI,
 * posted from a communication stream of the driver's own and tied to the filter stream by HIP events.  Results are
 * bit-identical to the single-GPU frame.  A driver holds the strips of the ranks of THIS process: one per process (one
 * process per GPU), several on several devices, or — for tests — several virtual ranks on one device (svgf_strip_transport
 * below).  See svgf_amd/csrc/svgf_strip.hip.
 * librccl is opened at run time (the one already in the process, else ROCm's; SVGF_RCCL_LIBRARY overrides). */
enum svgf_halo_plan { SVGF_PLAN_AUTO = 0, SVGF_PLAN_GHOST = 1, SVGF_PLAN_GROUPED = 2, SVGF_PLAN_PER_ITERATION = 3 };
typedef struct svgf_strips svgf_strips;
typedef struct svgf_strip_layout {
    int plan;                             /* the plan in force (AUTO resolved: the fewest exchanges whose halo fits the strips) */
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
/* How the ranks of a driver reach their neighbours.
 *   SVGF_TRANSPORT_RCCL           ncclSend / ncclRecv to the neighbour's rank of comms[k] — the product transport (one process per GPU, or one
 *                                 process driving several devices with one communicator per device).
 *   SVGF_TRANSPORT_RCCL_LOOPBACK  tests and the one-GPU simulation: ONE communicator of size 1 (comms[0]); every peer is its rank 0 and all virtual
 *                                 ranks share one communication stream.  Exercises RCCL's groups and kernels, not the peer addressing.
 *   SVGF_TRANSPORT_MAILBOX        tests: every rank of the partition lives in this process (nlocal == world, `comms` ignored) with a communication
 *                                 stream of its own, addresses its neighbours by their real rank numbers — the code path of a multi-GPU run — and the
 *                                 library matches each send to the receive its peer posted for it (posting order per {source, destination} pair,
 *                                 same group: RCCL's rule) and turns the pair into a device-to-device copy on the receiver's stream.  A send nobody
 *                                 receives, a receive nobody sends or a size mismatch — what deadlocks a real run — fails the frame with
 *                                 SVGF_ERR_COMM.  Not a product transport: it cannot cross a process boundary. */
enum svgf_strip_transport { SVGF_TRANSPORT_RCCL = 0, SVGF_TRANSPORT_RCCL_LOOPBACK = 1, SVGF_TRANSPORT_MAILBOX = 2 };
/* ranks / devices / compute_streams (hipStream_t, NULL entries = the null stream) / comms (ncclComm_t; loop-back: comms[0] only;
 * may be NULL when world == 1 or with the mailbox) describe the nlocal ranks of this process.  motion_reach = the largest |mv.y| (rows) the
 * temporal reprojection may need beyond what a strip computes itself; exceeding it is reported by svgf_strips_sync. */
int svgf_strips_create(svgf_strips** out, int width, int height, int world, const svgf_params* params, int plan, int motion_reach,
                       int nlocal, const int* ranks, const int* devices, void* const* compute_streams, void* const* comms, int transport);
/* The messages of ONE frame as rank `rank` posts them, in posting order (pure geometry, no device): what svgf_strips_frame hands to the
 * transport.  exchange 0 = the frame's state for the next frame's reprojection (posted once iteration 0 has fed the colour back,
 * waited for at the start of the next frame); exchange g >= 1 = the filter rows in front of iteration group g of the halo plan.
 * Every send has its mirror among the peer's receives of the same exchange — same plane, same global rows, same bytes — in the same
 * order per pair of ranks (tests/test_strips_cpu.py walks world = 2..8).  *count receives the number of messages; SVGF_ERR_INVALID
 * if it exceeds `capacity` (the first `capacity` are written). */
typedef struct svgf_strip_message {
    int exchange;
    int send;                 /* 1: this rank sends, 0: it receives */
    int peer;                 /* the neighbour's rank */
    int plane;                /* svgf_plane */
    int row_begin, row_end;   /* global rows */
    size_t bytes;
} svgf_strip_message;
int svgf_strips_messages(int width, int height, int rank, int world, int steps, int plan, int moments_radius, int motion_reach, int storage,
                         svgf_strip_message* out, int capacity, int* count);
/* SVGF_TRANSPORT_MAILBOX only, for the tests of the matching itself: the next send / receive rank `rank` posts is dropped, or its next
 * receive posted with half its size — the defects of a schedule that a multi-GPU run would answer with a hang.  The frame that meets the
 * defect fails with SVGF_ERR_COMM (the text names the ranks and the bytes) and the driver refuses further frames. */
enum svgf_mailbox_fault { SVGF_FAULT_NONE = 0, SVGF_FAULT_DROP_SEND = 1, SVGF_FAULT_DROP_RECV = 2, SVGF_FAULT_SHORT_RECV = 3 };
int svgf_strips_mailbox_fault(svgf_strips* s, int rank, int fault);
/* SVGF_TRANSPORT_MAILBOX only: groups matched, copies enqueued and bytes copied so far (any pointer may be NULL). */
int svgf_strips_transport_stats(const svgf_strips* s, unsigned long long* groups, unsigned long long* copies, unsigned long long* bytes);
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
/* Two frames in flight for the strips — svgf_set_frames_in_flight for the driver's contexts: with frames = 2, iterations 1.. of a frame
 * (their halo exchanges included) run on a stream of the driver's own beside the NEXT frame's temporal launch; results are bit-identical.
 * results[k] of call f is ORDERED on the rank's compute stream only by call f + 1 or svgf_strips_sync — enqueue its consumer after one of
 * those — and stays valid until call f + 2 (frames alternate between two pairs of filter planes); cur[k] is not read after the call has
 * returned (a frame whose iterations would read it — the direct kernel — keeps its tail on the compute stream).  Default 1. */
int svgf_strips_set_frames_in_flight(svgf_strips* s, int frames);
/* Edge rows first (default 1).  The iteration in front of a halo exchange produces the rows its neighbours wait for FIRST.  With enable = 1 that is
 * ONE launch over {the two edge ranges, the first third of the interior}: its first workgroups compute the edge ranges (written through to memory), the
 * last of them to finish writes a sequence number into device memory, and the communication stream — created at the highest priority — waits for
 * that word (hipStreamWaitValue64) and posts the exchange while the interior still runs; the rest of the interior is a second launch, so that the
 * exchange's kernel finds a compute unit at the boundary between the two (beside a launch that oversubscribes every CU it does not, whatever its
 * priority) and is done when the next iteration wants its rows.  The frame's state exchange is posted behind its last exchange of filter rows, and no
 * event is recorded on the filter stream that nothing waits for.  Measured on an 8K/8 strip: the per-iteration plan -14 %, grouped -5 % per frame
 * (DESIGN.md 5).  enable = 0 (and any iteration the direct kernel runs, and devices without stream memory operations,
 * hipDeviceAttributeCanUseStreamWaitValue): round 4's schedule — two edge launches, an event, the exchange, an interior launch.  Same bits either way. */
int svgf_strips_set_edge_first(svgf_strips* s, int enable);
/* HIP events around the a-trous launches of the first local rank on every n-th frame (0 = off); read: launches, their summed
 * ms, the pixels they covered in all iterations and in iteration 0 (for the roofline's algorithmic bytes). */
int svgf_strips_timing_enable(svgf_strips* s, int every);
int svgf_strips_timing_read(svgf_strips* s, int* launches, double* ms, double* px_all, double* px_iter0);

#ifdef __cplusplus
}
#endif
#endif /* SVGF_MI355X_H */
