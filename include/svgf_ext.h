/* svgf_ext.h — opt-ins and diagnostics of libsvgf_mi355x.so (same library, same ABI version as svgf.h).
 * Nothing here is needed to replace the reference's three call sites: strip contexts and row ranges for hosts that
 * partition the frame themselves, fused stage calls, throughput modes, measurement hooks, the strip driver's switches. */
#ifndef SVGF_MI355X_EXT_H
#define SVGF_MI355X_EXT_H

#include "svgf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Strip contexts: a context that holds only rows [y0, y0+rows) of a WxH frame (what the strip driver builds on).  "Inside the
 * frame" tests always use the global frame, so strip results are bit-identical to the whole-frame result while the halo rows hold valid data. */
int  svgf_create_strip(svgf_ctx** out, int width, int height, const svgf_strip* strip,
                       const svgf_params* params, int device, void* hip_stream);
int  svgf_resize_strip(svgf_ctx* ctx, int width, int height, const svgf_strip* strip);
/* Restrict the following stage calls to global rows [row_begin,row_end) (interior/boundary split of
 * a strip); (-1,-1) restores the owned rows. */
int  svgf_set_rows(svgf_ctx* ctx, int row_begin, int row_end);
/* svgf_sync returns SVGF_ERR_HALO if, since the last call, the temporal stage of a STRIP context reprojected a pixel to a row inside
 * the frame that the strip does not hold (such a pixel was treated as a rejection: the strip is no longer bit-identical to the whole
 * frame).  svgf_halo_violations returns the count (and zeroes it if clear != 0) without turning it into an error. */
int  svgf_halo_violations(svgf_ctx* ctx, unsigned long long* count, int clear);
/* Global rows [row_begin,row_end) of the previous-frame planes (colour, moments, history, previous G-buffer) that hold VALID
 * state; default (-1,-1) = every row the strip holds.  A strip whose planes are taller than the rows it keeps up to date (the
 * a-trous halos are wider than the state halo) declares the valid ones here: a reprojection beyond them counts as a halo
 * violation instead of silently reading stale rows.  The strip driver sets this itself. */
int  svgf_set_valid_rows(svgf_ctx* ctx, int row_begin, int row_end);

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
 * Replayed frames equal directly enqueued ones bit for bit (tests/test_gpu_graph.py).  Measured (tools/archive/graph_replay.py): with one
 * frame in flight a replay costs the device what the calls cost (the launches are not host-bound: 7 us against 30 us of host time
 * per frame, no device time saved); with two frames in flight the cross-stream edges of a graph are cheaper than the event waits of
 * the calls: -5 % at 1080p and -9 % at 720p against one frame in flight enqueued call by call. */
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

/* Per-stage device timing with HIP events on the context's stream (the reference only prints whole
 * frame time, App.cu:727-731).  Slots: 0 temporal, 1 moments, 2+i a-trous iteration i (when iterations 0 and 1 run as one
 * launch, slot 2 holds the pair and slot 3 the ~1 us between two events). */
int svgf_timing_enable(svgf_ctx* ctx, int on);                                  /* 0 = off, n = time every n-th frame (events cost ~1 us each) */
int svgf_timing_read(svgf_ctx* ctx, double* ms_sum, int* frames, int slots);   /* synchronises; resets sums */

/* Which tap path the waves of the LDS-streaming a-trous launches take.  After svgf_path_stats_enable(ctx, 1) every such launch of the
 * context adds, per step 1 << i (i < SVGF_PATH_STAT_STEPS), {wave-steps that filtered a surface pixel, those of them on the uniform-normal
 * path (8 instead of 13 vector instructions per tap, same bits)} to device counters: one pair of atomics per wave and band, results
 * unchanged.  svgf_path_stats_read synchronises, copies counts[2 * i], counts[2 * i + 1] (slots = the array's length) and zeroes them. */
#define SVGF_PATH_STAT_STEPS 7
int svgf_path_stats_enable(svgf_ctx* ctx, int on);
int svgf_path_stats_read(svgf_ctx* ctx, unsigned long long* counts, int slots);

/* ---- The strip driver's switches --------------------------------------------------------------------------------------- */
/* Two frames in flight for the strips — svgf_set_frames_in_flight for the driver's contexts: with frames = 2, iterations 1.. of a frame
 * (their halo exchanges included) run on a stream of the driver's own beside the NEXT frame's temporal launch; results are bit-identical.
 * results[k] of call f is ORDERED on the rank's compute stream only by call f + 1 or svgf_strips_sync — enqueue its consumer after one of
 * those — and stays valid until call f + 2 (frames alternate between two pairs of filter planes); cur[k] is not read after the call has
 * returned (a frame whose iterations would read it — the direct kernel — keeps its tail on the compute stream).  Default 1. */
int svgf_strips_set_frames_in_flight(svgf_strips* s, int frames);
/* Edge rows first (opt-in, default 0).  The iteration in front of a halo exchange produces the rows its neighbours wait for FIRST.  Default: two edge
 * launches, an event, the exchange, an interior launch — the exchange ordered by the event, inside HIP's memory model.  With enable = 1 that is
 * ONE launch over {the two edge ranges, the first third of the interior}: its first workgroups compute the edge ranges (written through to memory), the
 * last of them to finish writes a sequence number into signal memory, and the communication stream — created at the highest priority — waits for
 * that word (hipStreamWaitValue64) and posts the exchange while the interior still runs; the rest of the interior is a second launch.  Measured on
 * an 8K/8 strip of a one-GPU simulation: per-iteration plan -14 %, grouped -5 % per frame.  The visibility of the edge rows to RCCL's send kernel
 * rests on write-through stores and their acknowledgement, not on a release fence (svgf_atrous_lds.h says why and what it relies on): bit-identical
 * in every test on one device, NOT yet run on real peers — `bench.py --gpus N` verifies it against the one-GPU frame on the node it runs on before it
 * times it.  Any iteration the direct kernel runs, devices without stream memory operations (hipDeviceAttributeCanUseStreamWaitValue) or without
 * signal memory keep the default schedule.  Same bits either way. */
int svgf_strips_set_edge_first(svgf_strips* s, int enable);
/* HIP events around the a-trous launches of the first local rank on every n-th frame (0 = off); read: launches, their summed
 * ms, the pixels they covered in all iterations and in iteration 0 (for the roofline's algorithmic bytes). */
int svgf_strips_timing_enable(svgf_strips* s, int every);
int svgf_strips_timing_read(svgf_strips* s, int* launches, double* ms, double* px_all, double* px_iter0);

#ifdef __cplusplus
}
#endif
#endif /* SVGF_MI355X_EXT_H */
