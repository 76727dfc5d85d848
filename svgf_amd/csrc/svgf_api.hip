// svgf_api.hip — the C ABI of include/svgf.h: context, argument validation, the frame driver that
// replaces application::TemporalFilter/FilterMoments/WaveletFilter (src/App.cu:469-514) and the
// buffer lifecycle of application::ResizeRenderTextures (src/App.cu:742-778).
//
// There is deliberately NO CPU path here: every entry point either launches the gfx950 kernels
// or returns an error.

#include "svgf_ctx.h"

#include <algorithm>
#ifndef SVGF_GUIDE_MIN_STEPS
#ifndef SVGF_GUIDE_F16
#define SVGF_GUIDE_F16 1            // the guide plane with fp16 storage too
#endif
#define SVGF_GUIDE_MIN_STEPS 1       // the guide plane pays for itself with any wavelet iteration: +16 B/px written, -16 B/px in the next frame's reprojection test, -8 B/px per iteration
#endif
#include <cstdio>
#include <cstring>
#include <new>

namespace svgf_host {

int fail(svgf_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int hip_fail(svgf_ctx* c, hipError_t e, const char* what) {
    // the failure is reported through this library's own status and text: the runtime's "last error" is cleared, so that the host's next HIP call —
    // or the next hipGetLastError() of a framework that checks after every launch — does not trip over an error it did not cause
    (void)hipGetLastError();
    return fail(c, SVGF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

size_t colour_bytes(const svgf_ctx* c) { return (size_t)c->strip.rows * c->W * (c->p.storage == SVGF_F16 ? 8 : 16); }
size_t moments_bytes(const svgf_ctx* c) { return (size_t)c->strip.rows * c->W * (c->p.storage == SVGF_F16 ? 4 : 8); }
size_t hist_bytes(const svgf_ctx* c) { return (size_t)c->strip.rows * c->W; }

int check_params(svgf_ctx* c, const svgf_params* p) {
    if (!p) return fail(c, SVGF_ERR_INVALID, "params is null");
    if (p->steps < 0 || p->steps > SVGF_MAX_STEPS) return fail(c, SVGF_ERR_INVALID, "steps must be in [0,10] (GUI.cpp:988)");
    if (p->storage != SVGF_F32 && p->storage != SVGF_F16) return fail(c, SVGF_ERR_INVALID, "storage must be SVGF_F32 or SVGF_F16");
    if (p->moments_radius < 0 || p->moments_radius > 3) return fail(c, SVGF_ERR_INVALID, "moments_radius must be in [0,3]");
    if (p->variant < SVGF_VARIANT_AUTO || p->variant > SVGF_VARIANT_LDS_GENERAL) return fail(c, SVGF_ERR_INVALID, "unknown variant");
    if (p->nan_policy != SVGF_NAN_REFERENCE && p->nan_policy != SVGF_NAN_ZERO) return fail(c, SVGF_ERR_INVALID, "unknown nan_policy");
    return SVGF_OK;
}

// Geometry checks shared by svgf_create_strip and svgf_resize.  The kernels address a plane with 32-bit byte offsets
// (buffer resources: row offset in an SGPR, column offset in a VGPR), so the widest plane (16 B per pixel) of the strip
// must stay below 2 GiB.
int check_geometry(int width, int height, const svgf_strip* strip) {
    if (width <= 0 || height <= 0 || !strip) return SVGF_ERR_INVALID;
    if (strip->y0 < 0 || strip->rows <= 0 || strip->y0 + strip->rows > height) return SVGF_ERR_INVALID;
    if (strip->own_begin < strip->y0 || strip->own_end > strip->y0 + strip->rows || strip->own_begin > strip->own_end) return SVGF_ERR_INVALID;
    if ((unsigned long long)strip->rows * (unsigned long long)width * 16ull >= (1ull << 31)) return SVGF_ERR_INVALID;
    return SVGF_OK;
}

svgf::Geo geo_of(const svgf_ctx* c) { return svgf::Geo{c->W, c->H, c->strip.y0, c->strip.rows, c->rb, c->re}; }

// rows [rb,re) reading taps up to `reach` rows away must find them in [y0, y0+rows) or outside the frame
int check_halo(svgf_ctx* c, int reach, const char* stage) {
    const int lo = std::max(0, c->rb - reach), hi = std::min(c->H, c->re + reach);
    if (c->re > c->rb && (lo < c->strip.y0 || hi > c->strip.y0 + c->strip.rows))
        return fail(c, SVGF_ERR_HALO, std::string(stage) + ": rows need a halo of " + std::to_string(reach) + " rows that this strip does not hold");
    return SVGF_OK;
}

bool halo_held(const svgf_ctx* c, int reach) {
    const int lo = std::max(0, c->rb - reach), hi = std::min(c->H, c->re + reach);
    return c->re <= c->rb || (lo >= c->strip.y0 && hi <= c->strip.y0 + c->strip.rows);
}

int check_gbuf(svgf_ctx* c, const svgf_gbuffer* g, bool need_uv, const char* what) {
    if (!g || !g->motion || !g->normal || (need_uv && !g->uv)) return fail(c, SVGF_ERR_INVALID, std::string(what) + ": null G-buffer plane");
    return SVGF_OK;
}

hipEvent_t take_event(svgf_ctx* c) {
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Stream capture (svgf.h): 0 while the context's stream is not being captured, else a number that names the capture.
int capture_of(svgf_ctx* c, unsigned long long* id) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long n = 0;
    *id = 0;
    if (!c->stream) return SVGF_OK;                 // the legacy default stream cannot be captured
    SVGF_HIP(c, hipStreamGetCaptureInfo(c->stream, &st, &n));
    if (st == hipStreamCaptureStatusInvalidated) return fail(c, SVGF_ERR_HIP, "the capture of the context's stream has been invalidated (hipStreamEndCapture will report it)");
    if (st == hipStreamCaptureStatusActive) *id = n | (1ull << 63);
    return SVGF_OK;
}

int alloc_flags(svgf_ctx* c) {
    if (c->young_masks && c->young_list && c->young_count && c->nan_count && c->nan_list && c->sample_count && c->estimate_host) return SVGF_OK;
    // (svgf_temporal_moments under stream capture, first use of a context: the allocations below would invalidate the capture and their memsets
    // would be recorded into the graph, zeroing the counters on every replay — refused like svgf_denoise_frame's first frames)
    unsigned long long cap = 0;
    if (int rc = capture_of(c, &cap); rc != SVGF_OK) return rc;
    if (cap) return fail(c, SVGF_ERR_INVALID, "the context's stream is being captured and this call's first use of the context allocates its scratch: enqueue one call before the capture begins");
    // all or none: a failed allocation leaves nothing behind that a later call would mistake for a complete set
    auto drop = [&]() {
        if (c->young_masks) (void)hipFree(c->young_masks);
        if (c->young_list) (void)hipFree(c->young_list);
        if (c->young_count) (void)hipFree(c->young_count);
        if (c->nan_count) (void)hipFree(c->nan_count);
        if (c->nan_list) (void)hipFree(c->nan_list);
        if (c->sample_count) (void)hipFree(c->sample_count);
        if (c->estimate_host) (void)hipHostFree(c->estimate_host);
        c->young_masks = nullptr; c->young_list = nullptr; c->young_count = nullptr; c->nan_count = nullptr; c->nan_list = nullptr;
        c->sample_count = nullptr; c->estimate_host = nullptr;
    };
    drop();
    const size_t nmasks = (size_t)c->strip.rows * ((c->W + 63) / 64);
    hipError_t e = hipMalloc((void**)&c->young_masks, nmasks * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c->young_masks, 0, nmasks * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&c->young_list, svgf::young_list_entries(c->strip.rows, c->W) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&c->young_count, 2 * svgf::kYoungCounterStride * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c->young_count, 0, 2 * svgf::kYoungCounterStride * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&c->nan_count, 2 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(c->nan_count, 0, 2 * sizeof(unsigned), c->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&c->nan_list, (size_t)svgf::kNanListCap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&c->sample_count, 32 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c->sample_count, 0, 32 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->estimate_host, 64, hipHostMallocMapped);
    if (e == hipSuccess) { c->estimate_host[0] = 0ull; c->dense_moments = false; }
    if (e != hipSuccess) { drop(); return hip_fail(c, e, "alloc_flags"); }
    c->young_phase = 0;
    c->young_pending = false;
    return SVGF_OK;
}

bool is_strip(const svgf_ctx* c) { return c->strip.y0 != 0 || c->strip.rows != c->H; }

// the violation counter of a strip context (a whole-frame context cannot lose a reprojection: it holds every row)
int alloc_halo_counter(svgf_ctx* c) {
    if (c->halo_violations || !is_strip(c)) return SVGF_OK;
    unsigned long long cap = 0;
    if (int rc = capture_of(c, &cap); rc != SVGF_OK) return rc;
    if (cap) return fail(c, SVGF_ERR_INVALID, "the context's stream is being captured and a strip context's first temporal launch allocates: enqueue one before the capture begins");
    SVGF_HIP(c, hipMalloc((void**)&c->halo_violations, sizeof(unsigned)));
    SVGF_HIP(c, hipMemsetAsync(c->halo_violations, 0, sizeof(unsigned), c->stream));
    return SVGF_OK;
}

int alloc_state(svgf_ctx* c) {
    if (c->have_state) return SVGF_OK;
    for (int i = 0; i < 2; i++) {
        SVGF_HIP(c, hipMalloc(&c->colour[i], colour_bytes(c)));
        SVGF_HIP(c, hipMalloc(&c->moments[i], moments_bytes(c)));
        SVGF_HIP(c, hipMalloc(&c->filter[i], colour_bytes(c)));
        SVGF_HIP(c, hipMalloc((void**)&c->hist[i], hist_bytes(c)));
    }
    SVGF_HIP(c, hipMalloc(&c->guide, (size_t)c->strip.rows * c->W * 16));
    SVGF_HIP(c, hipMalloc(&c->guide_prev, (size_t)c->strip.rows * c->W * 16));
    c->guide_prev_valid = false;
    c->have_state = true;
    return reset_history(c);
}

// frames_in_flight == 2: the second pair of filter planes (zeroed like the first, so that both pairs read the same after a reset)
int alloc_alt(svgf_ctx* c) {
    if (c->frames_in_flight < 2 || (c->filter_alt[0] && c->filter_alt[1])) return SVGF_OK;
    for (int i = 0; i < 2; i++) {
        if (!c->filter_alt[i]) SVGF_HIP(c, hipMalloc(&c->filter_alt[i], colour_bytes(c)));
        SVGF_HIP(c, hipMemsetAsync(c->filter_alt[i], 0, colour_bytes(c), c->stream));
    }
    return SVGF_OK;
}

int join_side(svgf_ctx* c, hipStream_t onto) {
    if (!c->in_flight) return SVGF_OK;
    if (c->in_flight_capture) {
        // the tail was recorded in a stream capture.  If that capture is over it ended without svgf_flush, i.e. hipStreamEndCapture refused
        // it (unjoined work) and nothing of it will ever run: there is nothing to wait for, and its event must not be waited on.
        unsigned long long cap = 0;
        if (int rc = capture_of(c, &cap); rc != SVGF_OK) return rc;
        if (cap != c->in_flight_capture) { c->in_flight = false; c->in_flight_capture = 0; return SVGF_OK; }
    }
    SVGF_HIP(c, hipStreamWaitEvent(onto, c->ev_done, 0));
    c->in_flight = false;
    c->in_flight_capture = 0;
    return SVGF_OK;
}

void free_state(svgf_ctx* c) {
    for (int i = 0; i < 2; i++) {
        if (c->colour[i]) (void)hipFree(c->colour[i]);
        if (c->moments[i]) (void)hipFree(c->moments[i]);
        if (c->filter[i]) (void)hipFree(c->filter[i]);
        if (c->hist[i]) (void)hipFree(c->hist[i]);
        if (c->filter_alt[i]) (void)hipFree(c->filter_alt[i]);
        c->colour[i] = c->moments[i] = c->filter[i] = c->filter_alt[i] = nullptr;
        c->hist[i] = nullptr;
    }
    if (c->guide) (void)hipFree(c->guide);
    if (c->guide_prev) (void)hipFree(c->guide_prev);
    c->guide = c->guide_prev = nullptr;
    c->guide_prev_valid = false;
    if (c->young_masks) (void)hipFree(c->young_masks);
    if (c->young_list) (void)hipFree(c->young_list);
    if (c->young_count) (void)hipFree(c->young_count);
    if (c->nan_count) (void)hipFree(c->nan_count);
    if (c->nan_list) (void)hipFree(c->nan_list);
    if (c->sample_count) (void)hipFree(c->sample_count);
    if (c->estimate_host) (void)hipHostFree(c->estimate_host);
    c->young_masks = nullptr; c->young_list = nullptr; c->young_count = nullptr; c->nan_count = nullptr; c->nan_list = nullptr;
    c->sample_count = nullptr; c->estimate_host = nullptr;
    c->have_state = false;
}

int reset_history(svgf_ctx* c) {
    if (!c->have_state) return SVGF_OK;
    int rc = join_side(c, c->stream);             // the frame in flight still writes filter planes
    if (rc != SVGF_OK) return rc;
    for (int i = 0; i < 2; i++) {
        SVGF_HIP(c, hipMemsetAsync(c->colour[i], 0, colour_bytes(c), c->stream));
        SVGF_HIP(c, hipMemsetAsync(c->moments[i], 0, moments_bytes(c), c->stream));
        SVGF_HIP(c, hipMemsetAsync(c->filter[i], 0, colour_bytes(c), c->stream));
        if (c->filter_alt[i]) SVGF_HIP(c, hipMemsetAsync(c->filter_alt[i], 0, colour_bytes(c), c->stream));
        SVGF_HIP(c, hipMemsetAsync(c->hist[i], 0, hist_bytes(c), c->stream));
    }
    c->pingpong = 0;
    c->frames_since_reset = 0;
    c->result_index = 0;
    c->guide_prev_valid = false;
    return SVGF_OK;
}

// The violations counted so far (the stream is synchronised first); clear != 0 zeroes the counter.
int read_halo_violations(svgf_ctx* c, unsigned long long* count, int clear) {
    *count = 0;
    if (!c->halo_violations) return SVGF_OK;
    unsigned v = 0;
    SVGF_HIP(c, hipMemcpyAsync(&v, c->halo_violations, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    SVGF_HIP(c, hipStreamSynchronize(c->stream));
    if (clear && v) SVGF_HIP(c, hipMemsetAsync(c->halo_violations, 0, sizeof(unsigned), c->stream));
    *count = v;
    return SVGF_OK;
}

int temporal_impl(svgf_ctx* c, const void* prev_colour, const void* radiance, void* colour_out, const svgf_gbuffer* cur,
                  const svgf_gbuffer* prev, const uint8_t* hist_prev, uint8_t* hist_cur, void* moments_cur,
                  const void* moments_prev, void* passthrough_out, int sparse_colour, void* guide_out = nullptr, const void* guide_prev = nullptr) {
    if (!prev_colour || !radiance || !colour_out || !hist_prev || !hist_cur || !moments_cur || !moments_prev)
        return fail(c, SVGF_ERR_INVALID, "svgf_temporal: null plane");
    int rc = check_gbuf(c, cur, true, "svgf_temporal(cur)");
    if (rc == SVGF_OK) rc = check_gbuf(c, prev, true, "svgf_temporal(prev)");
    if (rc == SVGF_OK) rc = alloc_halo_counter(c);
    if (rc != SVGF_OK) return rc;
    if (prev_colour == colour_out || hist_prev == hist_cur || moments_prev == moments_cur)
        return fail(c, SVGF_ERR_INVALID, "svgf_temporal: previous and current state planes must differ (App. B #1)");
    if (passthrough_out && c->young_pending) {      // the lists of an earlier launch were never consumed (an error in between): start them again
        SVGF_HIP(c, hipMemsetAsync(c->young_count + c->young_phase * svgf::kYoungCounterStride, 0, svgf::kYoungCounterStride * sizeof(unsigned long long), c->stream));
        SVGF_HIP(c, hipMemsetAsync(c->nan_count + c->young_phase, 0, sizeof(unsigned), c->stream));
    }
    svgf::TemporalArgs a{prev_colour, radiance, colour_out,
                         (const float4*)cur->motion, (const uint2*)cur->normal, (const uint2*)cur->uv,
                         (const float4*)prev->motion, (const uint2*)prev->normal, (const uint2*)prev->uv,
                         hist_prev, hist_cur, moments_cur, moments_prev,
                         c->p.depth_threshold, c->p.normal_threshold, c->p.history_base, c->p.mesh_id_test, passthrough_out,
                         passthrough_out ? c->young_masks : nullptr, passthrough_out && !c->dense_now ? c->young_list : nullptr,
                         passthrough_out ? c->young_count + c->young_phase * svgf::kYoungCounterStride : nullptr,
                         passthrough_out ? c->young_count + (c->young_phase ^ 1) * svgf::kYoungCounterStride : nullptr, svgf::young_append_cap(c->strip.rows, c->W),
                         sparse_colour, c->p.phi_normal > 0.0f, c->halo_violations,
                         std::max(c->vy0, c->strip.y0) - c->strip.y0, std::min(c->vy1, c->strip.y0 + c->strip.rows) - c->strip.y0, (uint4*)guide_out,
                         (const uint4*)guide_prev,
                         c->strip.y0, c->strip.y0 + c->strip.rows,       // the guide texels of every row held (a strip runs the stage on fewer)
                         passthrough_out ? c->nan_list : nullptr, passthrough_out ? c->nan_count + c->young_phase : nullptr,
                         passthrough_out ? c->nan_count + (c->young_phase ^ 1) : nullptr,
                         passthrough_out ? c->sample_count + c->young_phase * 16 : nullptr, passthrough_out ? c->sample_count + (c->young_phase ^ 1) * 16 : nullptr,
                         passthrough_out ? c->estimate_host : nullptr, c->cold_now, c->p.nan_policy == SVGF_NAN_ZERO};
    if (c->re <= c->rb) return SVGF_OK;             // nothing to launch: the young masks and the counters stay as they are
    SVGF_HIP(c, svgf::launch_temporal(geo_of(c), c->p.storage, a, c->stream));
    if (passthrough_out) c->young_pending = true;
    return SVGF_OK;
}

int moments_impl(svgf_ctx* c, const void* colour, void* out, const void* moments, const svgf_gbuffer* g, const uint8_t* hist, int cold_only,
                 int dense, int sparse_colour) {
    if (!colour || !out || !moments || !hist) return fail(c, SVGF_ERR_INVALID, "svgf_moments: null plane");
    if (colour == out) return fail(c, SVGF_ERR_INVALID, "svgf_moments: in-place filtering is a race");
    int rc = check_gbuf(c, g, false, "svgf_moments");
    if (rc == SVGF_OK) rc = check_halo(c, c->p.moments_radius, "svgf_moments");
    if (rc != SVGF_OK) return rc;
    svgf::MomentsArgs a{colour, out, moments, (const float4*)g->motion, (const uint2*)g->normal, hist,
                        c->p.phi_colour, c->p.phi_normal, c->p.moments_radius, cold_only, dense, sparse_colour,
                        cold_only ? c->young_masks : nullptr, cold_only ? c->young_list : nullptr, cold_only ? c->young_count + c->young_phase * svgf::kYoungCounterStride : nullptr, svgf::young_append_cap(c->strip.rows, c->W),
                        cold_only ? c->nan_list : nullptr, cold_only ? c->nan_count + c->young_phase : nullptr,
                        c->p.variant == SVGF_VARIANT_LDS_GENERAL};
    if (c->re <= c->rb) {
        // No moments rows.  If the temporal launch of this frame did run (young_pending), its list is dropped: the counter pair still
        // has to turn over, because that launch zeroed the OTHER counter for the next frame.
        if (cold_only && c->young_pending) { c->young_phase ^= 1; c->young_pending = false; }
        return SVGF_OK;
    }
    SVGF_HIP(c, svgf::launch_moments(geo_of(c), c->p.storage, a, c->p.variant == SVGF_VARIANT_DIRECT, c->stream));
    if (cold_only) { c->young_phase ^= 1; c->young_pending = false; }   // the next frame's temporal launch appends to the counter this frame's one zeroed
    return SVGF_OK;
}

bool use_guide(const svgf_ctx* c) {
    return (c->p.storage == SVGF_F32 || SVGF_GUIDE_F16) && c->p.steps >= SVGF_GUIDE_MIN_STEPS && c->p.variant != SVGF_VARIANT_DIRECT && c->guide != nullptr;
}


// The reprojection test reads {depth, normal, instance ID} of the previous G-buffer; the guide plane the previous frame wrote
// holds exactly those bits.  It stands in for `prev` only when `prev` IS the G-buffer that frame was given as current (same
// three planes, by address) and is not the current one: a host that passes anything else gets the planes read as they are.
const void* prev_guide_for(const svgf_ctx* c, const svgf_gbuffer* cur, const svgf_gbuffer* prev) {
    if (!c->prev_guide_enabled || !c->guide_prev_valid || !c->guide_prev || !use_guide(c) || !cur || !prev) return nullptr;
    const svgf_gbuffer& k = c->guide_prev_of;
    if (prev->motion != k.motion || prev->normal != k.normal || prev->uv != k.uv) return nullptr;
    if (prev->motion == cur->motion || prev->normal == cur->normal || prev->uv == cur->uv) return nullptr;
    return c->guide_prev;
}

void commit_guide(svgf_ctx* c, const svgf_gbuffer* cur, bool written) {
    if (!written || !cur) { c->guide_prev_valid = false; return; }
    std::swap(c->guide, c->guide_prev);
    c->guide_prev_of = *cur;
    c->guide_prev_valid = true;
}

// svgf_path_stats_enable: the counter pair of step 1 << i (the LDS-streaming kernel's steps), or null
static unsigned long long* path_stats_of(const svgf_ctx* c, int step) {
    if (!c->path_stats || step < 1 || (step & (step - 1)) != 0) return nullptr;
    int i = 0;
    while ((1 << i) < step) i++;
    return i < SVGF_PATH_STAT_STEPS ? c->path_stats + 2 * i : nullptr;
}

int atrous_impl(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g, int step, int iteration, const void* guide) {
    if (!in || !out) return fail(c, SVGF_ERR_INVALID, "svgf_atrous: null plane");
    if (in == out || in == feedback) return fail(c, SVGF_ERR_INVALID, "svgf_atrous: in-place filtering is a race");
    if (step < 1 || step > (1 << SVGF_MAX_STEPS)) return fail(c, SVGF_ERR_INVALID, "svgf_atrous: step out of range");
    int rc = check_gbuf(c, g, false, "svgf_atrous");
    if (rc == SVGF_OK) rc = check_halo(c, 2 * step, "svgf_atrous");
    if (rc != SVGF_OK) return rc;
    svgf::AtrousArgs a{in, out, iteration == 0 ? feedback : nullptr, (const float4*)g->motion, (const uint2*)g->normal,
                       step, c->p.phi_colour, c->p.phi_normal, (const uint4*)guide, c->p.variant == SVGF_VARIANT_LDS_GENERAL, path_stats_of(c, step)};
    SVGF_HIP(c, svgf::launch_atrous(geo_of(c), c->p.storage, c->p.variant, a, c->stream));
    return SVGF_OK;
}

// One iteration over up to three row ranges in ONE launch, the first `nfirst` ranges produced first and signalled (svgf_kernels.h: AtrousRanges;
// the strip driver's edge rows, svgf_strip.hip).  The caller has checked atrous_ranges_ok().
bool atrous_ranges_ok(const svgf_ctx* c, int step) {
    svgf::AtrousArgs a{};
    a.step = step; a.phi_normal = c->p.phi_normal;
    return svgf::atrous_ranges_available(c->p.variant, a);
}
int atrous_ranges_impl(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g, int step, int iteration, const void* guide, const svgf::AtrousRanges& r) {
    if (!in || !out) return fail(c, SVGF_ERR_INVALID, "svgf_atrous: null plane");
    if (in == out || in == feedback) return fail(c, SVGF_ERR_INVALID, "svgf_atrous: in-place filtering is a race");
    int rc = check_gbuf(c, g, false, "svgf_atrous");
    if (rc != SVGF_OK) return rc;
    const int rb = c->rb, re = c->re;
    for (int k = 0; k < r.n && rc == SVGF_OK; k++) { c->rb = r.yb[k]; c->re = r.ye[k]; rc = check_halo(c, 2 * step, "svgf_atrous"); }
    c->rb = rb; c->re = re;
    if (rc != SVGF_OK) return rc;
    svgf::AtrousArgs a{in, out, iteration == 0 ? feedback : nullptr, (const float4*)g->motion, (const uint2*)g->normal,
                       step, c->p.phi_colour, c->p.phi_normal, (const uint4*)guide, c->p.variant == SVGF_VARIANT_LDS_GENERAL, path_stats_of(c, step)};
    SVGF_HIP(c, svgf::launch_atrous_ranges(geo_of(c), c->p.storage, a, r, c->stream));
    return SVGF_OK;
}

// Iterations 0 and 1 of application::WaveletFilter's loop (App.cu:497-507: steps 1 and 2) as ONE launch: iteration 0's rows never
// leave the chip except as the feedback plane (svgf_atrous_fused.h).  `out` receives iteration 1's result on rows [rb, re).
int atrous_pair_impl(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g, const void* guide) {
    if (!in || !out) return fail(c, SVGF_ERR_INVALID, "svgf_atrous_pair: null plane");
    if (in == out || in == feedback || out == feedback) return fail(c, SVGF_ERR_INVALID, "svgf_atrous_pair: in, out and feedback must be three planes");
    int rc = check_gbuf(c, g, false, "svgf_atrous_pair");
    if (rc == SVGF_OK) rc = check_halo(c, 2 + 4 + 0, "svgf_atrous_pair");       // iteration 1 reaches 4 rows of iteration 0, which reaches 2
    if (rc != SVGF_OK) return rc;
    svgf::AtrousArgs a{in, out, feedback, (const float4*)g->motion, (const uint2*)g->normal,
                       1, c->p.phi_colour, c->p.phi_normal, (const uint4*)guide, c->p.variant == SVGF_VARIANT_LDS_GENERAL};
    if (!svgf::atrous_fused_available(c->p.variant, a)) return fail(c, SVGF_ERR_INVALID, "svgf_atrous_pair: needs the LDS kernels (variant != direct) and PhiNormal != 0");
    SVGF_HIP(c, svgf::launch_atrous_fused(geo_of(c), c->p.storage, a, c->stream));
    return SVGF_OK;
}

bool can_fuse01(const svgf_ctx* c) {
    return c->fuse01 && c->p.steps >= 2 && c->p.variant != SVGF_VARIANT_DIRECT && c->p.phi_normal != 0.0f;
}

// Which kernel serves this frame's young pixels (frame and strip drivers).  The first three frames after a reset have history <= 3 everywhere: the
// LDS-streaming kernel, which visits every pixel (0.21 ms per 4K frame, whatever is young).  Afterwards the young-pixel launch, which costs what the
// young pixels cost (0.005 ms for none, 0.03 under a pan, 0.5 for 12 % of the frame, 1.6 for half of it) — unless a recent frame's SAMPLE of young
// pixels (temporal_kernel, one wave in 64; read here without synchronising: it is a few frames old) says that more than 8 % of the frame are young
// (fast camera motion, a cut without a reset; back below 5 %) or that more waves hold young pixels than the list takes appends from (thin geometry
// under motion; back below three quarters of that: the bench pan's 8 000 - 10 000 such waves must not keep a context there).  Both kernels evaluate
// the estimate on the same bits (moments_group8, ARITH = 1), so the choice — and the timing it depends on — changes nothing but the frame time; the
// ranks of a strip driver choose each for itself.  Rows: [c->rb, c->re) as they are when this is called (the temporal rows).
void choose_moments_kernel(svgf_ctx* c, bool* cold, bool* crowded) {
    // Only where launch_moments has a kernel that visits every pixel for a dense frame: the LDS-streaming one (the reference's radius, PhiNormal != 0)
    // or the 3x3 shuffle kernel.  Any other setting keeps the young-pixel launch — which needs the temporal launch's LIST, and a cold or crowded
    // frame's temporal launch appends to none (radius 0 / 2 or PhiNormal == 0 under the default variants lost the young pixels of the
    // partly young segments of the first three frames).
    const bool streams = c->p.variant != SVGF_VARIANT_DIRECT && ((c->p.moments_radius == 3 && c->p.phi_normal != 0.0f) || c->p.moments_radius == 1);
    *cold = c->frames_since_reset < 3 && streams;
    *crowded = false;
    if (!*cold && c->adaptive_moments && c->estimate_host && streams && c->p.moments_radius == 3 && c->re > c->rb) {
        const unsigned long long sample = *(volatile unsigned long long*)c->estimate_host;   // {waves that hold some young pixels: high word, young pixels: low word}, of one wave in 64
        const double est = 64.0 * (double)(unsigned)sample / ((double)c->W * (double)(c->re - c->rb));
        const unsigned long long appends = 64ull * (sample >> 32), cap = svgf::young_append_cap(c->strip.rows, c->W);   // what the list of such a frame takes, and its cap
        if (est > 0.08 || appends > cap) c->dense_moments = true;
        else if (est < 0.05 && appends < cap / 4 * 3) c->dense_moments = false;
        *crowded = c->dense_moments;
    }
    c->dense_now = *cold || *crowded;               // (the temporal launch of such a frame appends to no list)
    c->cold_now = *cold;                            // (... and of a cold one adds nothing to the sample)
}

int temporal_moments_impl(svgf_ctx* c, const void* prev_colour, const void* radiance, void* colour_out, void* filter_out,
                          const svgf_gbuffer* cur, const svgf_gbuffer* prev, const uint8_t* hist_prev, uint8_t* hist_cur,
                          void* moments_cur, const void* moments_prev, int mrb, int mre, int feedback_follows, void* guide_out, const void* guide_prev, int dense) {
    if (!filter_out || filter_out == colour_out) return fail(c, SVGF_ERR_INVALID, "svgf_temporal_moments: filter_out must be a plane of its own");
    const int rb = c->rb, re = c->re;
    if (mrb == -1 && mre == -1) { mrb = rb; mre = re; }
    if (mrb < rb || mre > re || mrb > mre) return fail(c, SVGF_ERR_INVALID, "svgf_temporal_moments: moments rows outside the temporal rows");
    int rc = alloc_flags(c);
    if (rc == SVGF_OK) rc = temporal_impl(c, prev_colour, radiance, colour_out, cur, prev, hist_prev, hist_cur, moments_cur, moments_prev, filter_out, feedback_follows != 0, guide_out, guide_prev);
    c->dense_now = c->cold_now = false;             // (a driver's choice for THIS frame, choose_moments_kernel: the stage calls keep their lists)
    if (rc != SVGF_OK) return rc;
    c->rb = mrb; c->re = mre;
    rc = moments_impl(c, colour_out, filter_out, moments_cur, cur, hist_cur, 1, dense, feedback_follows != 0);
    c->rb = rb; c->re = re;
    return rc;
}

}  // namespace svgf_host

using namespace svgf_host;

extern "C" {

void svgf_default_params(svgf_params* p) {
    if (!p) return;
    p->steps = 3;                 // App.h:109
    p->depth_threshold = 0.8f;    // App.h:110
    p->normal_threshold = 0.9f;   // App.h:111
    p->history_base = 24;         // App.h:112
    p->phi_colour = 10.0f;        // App.h:113
    p->phi_normal = 128.0f;       // App.h:114
    p->moments_radius = 3;        // Filter.cuh:465
    p->storage = SVGF_F16;        // Filter.cuh:15-16
    p->mesh_id_test = 1;          // the test Filter.cuh:245-247 intends; 0 = what the reference's binary does (see svgf.h)
    p->variant = SVGF_VARIANT_AUTO;
    p->nan_policy = SVGF_NAN_REFERENCE;  // what Filter.cuh does with a NaN texel (svgf.h)
}

const char* svgf_status_string(int s) {
    switch (s) {
        case SVGF_OK: return "ok";
        case SVGF_ERR_INVALID: return "invalid argument";
        case SVGF_ERR_HIP: return "HIP runtime error";
        case SVGF_ERR_NO_DEVICE: return "no usable gfx950 device";
        case SVGF_ERR_HALO: return "strip halo too small";
        case SVGF_ERR_ALLOC: return "allocation failed";
        case SVGF_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}

const char* svgf_last_error(const svgf_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int svgf_abi_version(void) { return SVGF_ABI_VERSION; }

int svgf_create_strip(svgf_ctx** out, int width, int height, const svgf_strip* strip, const svgf_params* params,
                      int device, void* hip_stream) {
    if (!out) return SVGF_ERR_INVALID;
    *out = nullptr;
    if (!params) return SVGF_ERR_INVALID;
    int rc = check_geometry(width, height, strip);
    if (rc != SVGF_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return SVGF_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return SVGF_ERR_NO_DEVICE;
    svgf_ctx* c = new (std::nothrow) svgf_ctx();
    if (!c) return SVGF_ERR_ALLOC;
    c->W = width; c->H = height; c->strip = *strip; c->rb = strip->own_begin; c->re = strip->own_end;
    c->vy0 = strip->y0; c->vy1 = strip->y0 + strip->rows;
    c->device = device; c->stream = (hipStream_t)hip_stream;
    rc = check_params(c, params);
    if (rc != SVGF_OK) { delete c; return rc; }
    c->p = *params;
    c->p.history_base = std::min(std::max(c->p.history_base, 1), 255);
    *out = c;
    return SVGF_OK;
}

int svgf_create(svgf_ctx** out, int width, int height, const svgf_params* params, int device, void* hip_stream) {
    svgf_strip s{0, height, 0, height};
    return svgf_create_strip(out, width, height, &s, params, device, hip_stream);
}

// the side stream and its events (frames_in_flight back to 1, destruction); waits for what is on it
static void drop_side(svgf_ctx* c) {
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); c->side = nullptr; }
    if (c->ev_first) { (void)hipEventDestroy(c->ev_first); c->ev_first = nullptr; }
    if (c->ev_done) { (void)hipEventDestroy(c->ev_done); c->ev_done = nullptr; }

    c->in_flight = false;
}

void svgf_destroy(svgf_ctx* c) {
    if (!c || c->strip_drv) return;            // a context handed out by svgf_strips_context belongs to its strip driver (svgf_strips_destroy)
    DeviceGuard dg(c->device);
    if (c->have_state || !c->pool.empty() || !c->pending.empty() || c->halo_violations) (void)hipStreamSynchronize(c->stream);
    drop_side(c);
    free_state(c);
    if (c->halo_violations) (void)hipFree(c->halo_violations);
    if (c->path_stats) (void)hipFree(c->path_stats);
    for (auto& f : c->pending) for (auto e : f.ev) (void)hipEventDestroy(e);
    for (auto e : c->pool) (void)hipEventDestroy(e);
    delete c;
}

// application::ResizeRenderTextures (App.cu:742-778): the render size changed, every filter buffer is reallocated and the
// accumulation restarts (ResetRender, App.cu:777).  State planes are freed here and come back — exact size, zeroed — with the
// next svgf_denoise_frame; tunables, stream, device and timing settings stay.
int svgf_resize_strip(svgf_ctx* c, int width, int height, const svgf_strip* strip) {
    if (!c) return SVGF_ERR_INVALID;
    if (check_geometry(width, height, strip) != SVGF_OK) return fail(c, SVGF_ERR_INVALID, "svgf_resize: bad frame size or strip");
    if (c->strip_drv) return fail(c, SVGF_ERR_INVALID, "svgf_resize: this context belongs to a strip driver: svgf_strips_destroy it and create one for the new size");
    DeviceGuard dg(c->device);
    SVGF_HIP(c, hipStreamSynchronize(c->stream));        // cudaDeviceSynchronize() in the reference (App.cu:758)
    if (c->side) SVGF_HIP(c, hipStreamSynchronize(c->side));
    c->in_flight = false;
    free_state(c);
    if (c->halo_violations) { (void)hipFree(c->halo_violations); c->halo_violations = nullptr; }
    c->W = width; c->H = height; c->strip = *strip; c->rb = strip->own_begin; c->re = strip->own_end;
    c->vy0 = strip->y0; c->vy1 = strip->y0 + strip->rows;
    c->pingpong = 0; c->frames_since_reset = 0; c->result_index = 0; c->filter_set = 0; c->last_pair_alt = false;
    return SVGF_OK;
}

int svgf_resize(svgf_ctx* c, int width, int height) {
    svgf_strip s{0, height, 0, height};
    return svgf_resize_strip(c, width, height, &s);
}

int svgf_set_params(svgf_ctx* c, const svgf_params* p) {
    if (!c) return SVGF_ERR_INVALID;
    int rc = check_params(c, p);
    if (rc != SVGF_OK) return rc;
    // the halo plan of a strip driver was laid out for the iteration count and moments radius it was created with
    if (c->strip_drv && (p->steps != c->p.steps || p->moments_radius != c->p.moments_radius))
        return fail(c, SVGF_ERR_INVALID, "svgf_set_params: steps / moments_radius of a strip driver's context are fixed (its halo plan depends on them)");
    if (p->storage != c->p.storage) return fail(c, SVGF_ERR_INVALID, "storage cannot change after creation");
    c->p = *p;
    c->p.history_base = std::min(std::max(c->p.history_base, 1), 255);
    return SVGF_OK;
}

int svgf_set_stream(svgf_ctx* c, void* s) {
    if (!c) return SVGF_ERR_INVALID;
    if (c->in_flight) {                                   // the frame in flight is ordered on the stream the caller works on from now on
        DeviceGuard dg(c->device);
        int rc = join_side(c, (hipStream_t)s);
        if (rc != SVGF_OK) return rc;
    }
    c->stream = (hipStream_t)s;
    return SVGF_OK;
}

int svgf_set_rows(svgf_ctx* c, int rb, int re) {
    if (!c) return SVGF_ERR_INVALID;
    if (c->strip_drv) return fail(c, SVGF_ERR_INVALID, "svgf_set_rows: the strip driver sets the rows of its contexts");
    if (rb == -1 && re == -1) { c->rb = c->strip.own_begin; c->re = c->strip.own_end; return SVGF_OK; }
    if (rb < c->strip.y0 || re > c->strip.y0 + c->strip.rows || rb > re) return fail(c, SVGF_ERR_INVALID, "row range outside the strip");
    c->rb = rb; c->re = re;
    return SVGF_OK;
}

int svgf_set_valid_rows(svgf_ctx* c, int rb, int re) {
    if (!c) return SVGF_ERR_INVALID;
    if (rb == -1 && re == -1) { rb = c->strip.y0; re = c->strip.y0 + c->strip.rows; }
    if (rb < c->strip.y0 || re > c->strip.y0 + c->strip.rows || rb > re) return fail(c, SVGF_ERR_INVALID, "valid row range outside the strip");
    c->vy0 = rb; c->vy1 = re;
    return SVGF_OK;
}

int svgf_set_debug_mode(svgf_ctx* c, int mode) {
    if (!c) return SVGF_ERR_INVALID;
    if (mode < SVGF_DEBUG_FINAL || mode > SVGF_DEBUG_ATROUS) return fail(c, SVGF_ERR_INVALID, "unknown debug mode");
    if (mode != SVGF_DEBUG_FINAL && c->frames_in_flight > 1)
        return fail(c, SVGF_ERR_INVALID, "svgf_set_debug_mode: the debug views filter what the previous frame left in FilterBuffer[0]; set frames in flight back to 1 first");
    c->debug_mode = mode;
    return SVGF_OK;
}

int svgf_set_adaptive_moments(svgf_ctx* c, int enable) {
    if (!c) return SVGF_ERR_INVALID;
    c->adaptive_moments = enable != 0;
    if (!enable) c->dense_moments = false;
    return SVGF_OK;
}

int svgf_adaptive_moments_state(const svgf_ctx* c) { return c && c->dense_moments ? 1 : 0; }

int svgf_adaptive_moments_sample(const svgf_ctx* c, unsigned* young_pixels, unsigned* appending_waves) {
    if (!c || !c->estimate_host) return SVGF_ERR_INVALID;
    const unsigned long long sample = *(volatile unsigned long long*)c->estimate_host;
    auto sat = [](unsigned long long v) { return v > 0xffffffffull ? 0xffffffffu : (unsigned)v; };
    if (young_pixels) *young_pixels = sat(64ull * (sample & 0xffffffffull));
    if (appending_waves) *appending_waves = sat(64ull * (sample >> 32));
    return SVGF_OK;
}

int svgf_set_prev_guide(svgf_ctx* c, int enable) {
    if (!c) return SVGF_ERR_INVALID;
    c->prev_guide_enabled = enable != 0;
    return SVGF_OK;
}

int svgf_temporal(svgf_ctx* c, const void* prev_colour, const void* radiance, void* colour_out, const svgf_gbuffer* cur,
                  const svgf_gbuffer* prev, const uint8_t* hist_prev, uint8_t* hist_cur, void* moments_cur,
                  const void* moments_prev) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    return temporal_impl(c, prev_colour, radiance, colour_out, cur, prev, hist_prev, hist_cur, moments_cur, moments_prev, nullptr, 0);
}

int svgf_moments(svgf_ctx* c, const void* colour, void* out, const void* moments, const svgf_gbuffer* g, const uint8_t* hist) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    // stage call: every pixel is visited — the LDS-streaming kernel (variant DIRECT, a radius other than 3 or PhiNormal == 0: the per-pixel one).
    // The frame driver's young-pixel launch evaluates its taps as whichever of the two this call runs does (moments_group8): bit-identical results.
    return moments_impl(c, colour, out, moments, g, hist, 0, c->p.variant != SVGF_VARIANT_DIRECT, 0);
}

int svgf_temporal_moments(svgf_ctx* c, const void* prev_colour, const void* radiance, void* colour_out, void* filter_out,
                          const svgf_gbuffer* cur, const svgf_gbuffer* prev, const uint8_t* hist_prev, uint8_t* hist_cur,
                          void* moments_cur, const void* moments_prev, int mrb, int mre, int feedback_follows) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    return temporal_moments_impl(c, prev_colour, radiance, colour_out, filter_out, cur, prev, hist_prev, hist_cur, moments_cur, moments_prev, mrb, mre, feedback_follows);
}

int svgf_atrous(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g, int step, int iteration) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    return atrous_impl(c, in, out, feedback, g, step, iteration);
}

int svgf_atrous_pair(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    return atrous_pair_impl(c, in, out, feedback, g);
}

int svgf_set_iteration_fusion(svgf_ctx* c, int enable) {
    if (!c) return SVGF_ERR_INVALID;
    c->fuse01 = enable != 0;
    return SVGF_OK;
}

int svgf_set_frames_in_flight(svgf_ctx* c, int frames) {
    if (!c) return SVGF_ERR_INVALID;
    if (frames != 1 && frames != 2) return fail(c, SVGF_ERR_INVALID, "svgf_set_frames_in_flight: 1 or 2");
    if (frames == c->frames_in_flight) return SVGF_OK;
    if (frames == 2 && c->strip_drv) return fail(c, SVGF_ERR_INVALID, "svgf_set_frames_in_flight: the strip driver schedules its contexts itself");
    if (frames == 2 && c->debug_mode != SVGF_DEBUG_FINAL) return fail(c, SVGF_ERR_INVALID, "svgf_set_frames_in_flight: not with a debug view selected");
    DeviceGuard dg(c->device);
    if (frames == 2) {
        if (!c->side) SVGF_HIP(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        if (!c->ev_first) SVGF_HIP(c, hipEventCreateWithFlags(&c->ev_first, hipEventDisableTiming));
        if (!c->ev_done) SVGF_HIP(c, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    } else {
        int rc = join_side(c, c->stream);
        if (rc != SVGF_OK) return rc;
        // With one frame in flight every frame uses c->filter[], and svgf_state_plane(SVGF_PLANE_FILTER, ..) and the SVGF_DEBUG_ATROUS view
        // (which filters what the previous frame left in FilterBuffer, App.cu:611-620) index c->filter[] too: if the last frame used the
        // second pair, the pairs change names (the result pointer handed out stays valid: the planes themselves do not move).
        // (last_pair_alt is what the last FRAME wrote — not what filter_set implies: 2 -> 1 -> 2 -> 1 without a frame between the last two
        // switches must not swap again)
        if (c->last_pair_alt && c->filter_alt[0] && c->filter_alt[1]) {
            std::swap(c->filter[0], c->filter_alt[0]);
            std::swap(c->filter[1], c->filter_alt[1]);
            c->last_pair_alt = false;
        }
        c->filter_set = 0;
    }
    c->frames_in_flight = frames;
    return SVGF_OK;
}

int svgf_flush(svgf_ctx* c) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    return join_side(c, c->stream);
}

int svgf_taa(svgf_ctx* c, const void* filtered, const void* history, void* out) {
    if (!c) return SVGF_ERR_INVALID;
    if (!filtered || !history || !out) return fail(c, SVGF_ERR_INVALID, "svgf_taa: null plane");
    if (out == filtered || out == history) return fail(c, SVGF_ERR_INVALID, "svgf_taa: in-place filtering is a race");
    int rc = check_halo(c, 3, "svgf_taa");                                // samples sit 1-3 texels up-left of the pixel (fp32 rounding of uv*(N-1))
    if (rc != SVGF_OK) return rc;
    DeviceGuard dg(c->device);
    SVGF_HIP(c, svgf::launch_taa(geo_of(c), c->p.storage, filtered, history, out, c->p.variant == SVGF_VARIANT_DIRECT, c->stream));
    return SVGF_OK;
}

static int albedo_impl(svgf_ctx* c, int mode, const void* in, const void* albedo, void* out, const char* what) {
    if (!c) return SVGF_ERR_INVALID;
    if (!in || !albedo || !out) return fail(c, SVGF_ERR_INVALID, std::string(what) + ": null plane");
    if (albedo == out) return fail(c, SVGF_ERR_INVALID, std::string(what) + ": out must not alias the albedo plane");
    DeviceGuard dg(c->device);
    SVGF_HIP(c, svgf::launch_albedo(geo_of(c), c->p.storage, mode, in, albedo, out, c->stream));
    return SVGF_OK;
}
int svgf_demodulate(svgf_ctx* c, const void* radiance, const void* albedo, void* out) { return albedo_impl(c, 0, radiance, albedo, out, "svgf_demodulate"); }
int svgf_modulate(svgf_ctx* c, const void* filtered, const void* albedo, void* out) { return albedo_impl(c, 1, filtered, albedo, out, "svgf_modulate"); }

int svgf_pack_gbuffer(svgf_ctx* c, const void* position, const void* normal, const void* bary, const svgf_camera* cam,
                      void* motion_out, void* normal_out, void* uv_out) {
    if (!c) return SVGF_ERR_INVALID;
    if (!position || !normal || !bary || !cam || !motion_out || !normal_out || !uv_out) return fail(c, SVGF_ERR_INVALID, "svgf_pack_gbuffer: null argument");
    int rc = check_halo(c, 1, "svgf_pack_gbuffer");                       // the 2x2 quad partner row
    if (rc != SVGF_OK) return rc;
    svgf::PackArgs a;
    a.position = (const float4*)position; a.normal = (const float4*)normal; a.bary = (const float4*)bary;
    std::memcpy(a.vp, cam->view_proj, sizeof(a.vp)); std::memcpy(a.pvp, cam->prev_view_proj, sizeof(a.pvp)); std::memcpy(a.cam, cam->position, sizeof(a.cam));
    a.motion = (float4*)motion_out; a.normal_out = (uint2*)normal_out; a.uv_out = (uint2*)uv_out;
    DeviceGuard dg(c->device);
    SVGF_HIP(c, svgf::launch_pack_gbuffer(geo_of(c), a, c->stream));
    return SVGF_OK;
}

// ---- texture / pitched adapters: what the reference gets from its CUDA <-> OpenGL mappings (CudaUtil.h:68-99) ---------------
static size_t texel_bytes(int plane) { return plane == SVGF_GBUF_MOTION ? 16 : 8; }

int svgf_import_gbuffer_pitched(svgf_ctx* c, int plane, const void* src, size_t src_pitch_bytes, void* dst) {
    if (!c) return SVGF_ERR_INVALID;
    if (plane < SVGF_GBUF_MOTION || plane > SVGF_GBUF_UV || !src || !dst) return fail(c, SVGF_ERR_INVALID, "svgf_import_gbuffer_pitched: bad argument");
    const size_t row = (size_t)c->W * texel_bytes(plane);
    if (src_pitch_bytes < row) return fail(c, SVGF_ERR_INVALID, "svgf_import_gbuffer_pitched: pitch smaller than a row");
    DeviceGuard dg(c->device);
    SVGF_HIP(c, hipMemcpy2DAsync(dst, row, src, src_pitch_bytes, row, (size_t)c->strip.rows, hipMemcpyDeviceToDevice, c->stream));
    return SVGF_OK;
}

int svgf_import_gbuffer_array(svgf_ctx* c, int plane, const void* hip_array, void* dst) {
    if (!c) return SVGF_ERR_INVALID;
    if (plane < SVGF_GBUF_MOTION || plane > SVGF_GBUF_UV || !hip_array || !dst) return fail(c, SVGF_ERR_INVALID, "svgf_import_gbuffer_array: bad argument");
    const size_t row = (size_t)c->W * texel_bytes(plane);
    DeviceGuard dg(c->device);
    // the array holds the WHOLE frame (a render target); the strip's rows start at array row y0
    SVGF_HIP(c, hipMemcpy2DFromArrayAsync(dst, row, (hipArray_const_t)hip_array, 0, (size_t)c->strip.y0, row, (size_t)c->strip.rows, hipMemcpyDeviceToDevice, c->stream));
    return SVGF_OK;
}

int svgf_export_to_array(svgf_ctx* c, const void* plane_data, void* hip_array) {
    if (!c) return SVGF_ERR_INVALID;
    if (!plane_data || !hip_array) return fail(c, SVGF_ERR_INVALID, "svgf_export_to_array: null argument");
    const size_t row = (size_t)c->W * (c->p.storage == SVGF_F16 ? 8 : 16);
    DeviceGuard dg(c->device);
    // cudaMemcpyToArray(RenderTextureMapping->CudaTextureArray, 0, 0, FilterBuffer[1]->Data, ...) — App.cu:561
    SVGF_HIP(c, hipMemcpy2DToArrayAsync((hipArray_t)hip_array, 0, (size_t)c->strip.y0, plane_data, row, row, (size_t)c->strip.rows, hipMemcpyDeviceToDevice, c->stream));
    return SVGF_OK;
}

int svgf_reset_history(svgf_ctx* c) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    return reset_history(c);
}

int svgf_denoise_frame(svgf_ctx* c, const void* radiance, const svgf_gbuffer* cur, const svgf_gbuffer* prev, const void** result) {
    if (!c) return SVGF_ERR_INVALID;
    if (!radiance) return fail(c, SVGF_ERR_INVALID, "svgf_denoise_frame: null radiance");
    int rc = check_gbuf(c, cur, true, "svgf_denoise_frame(cur)");
    if (rc != SVGF_OK) return rc;
    DeviceGuard dg(c->device);
    // First frame: state is zero, so reprojecting onto the current G-buffer gives h = 1, alpha = 1 —
    // the same result as the reference's rejection against its cleared previous framebuffer.
    if (!prev) prev = cur;
    // Under stream capture a call records launches, it cannot allocate; and what a graph replays is what THIS call enqueues — the
    // cold-start moments kernel of the first three frames after a reset would be replayed for ever.
    unsigned long long cap = 0;
    rc = capture_of(c, &cap);
    if (rc != SVGF_OK) return rc;
    if (cap) {
        const bool allocates = !c->have_state || !(c->young_masks && c->young_list && c->young_count && c->nan_count && c->nan_list && c->sample_count && c->estimate_host) ||
                               (c->frames_in_flight > 1 && !(c->filter_alt[0] && c->filter_alt[1])) || (is_strip(c) && !c->halo_violations);
        if (allocates || c->frames_since_reset < 3)
            return fail(c, SVGF_ERR_INVALID, "svgf_denoise_frame: the context's stream is being captured and this frame cannot be: the first three frames after "
                                             "svgf_create / svgf_resize / svgf_reset_history allocate and take the cold-start path - enqueue them before the capture begins");
        if (c->in_flight && c->in_flight_capture == 0)      // (a tail left behind by an abandoned capture is dropped by join_side)
            return fail(c, SVGF_ERR_INVALID, "svgf_denoise_frame: the context's stream is being captured while a frame enqueued before the capture is still in flight: "
                                             "svgf_flush before hipStreamBeginCapture");
    }
    rc = alloc_state(c);
    if (rc == SVGF_OK) rc = alloc_flags(c);
    if (rc == SVGF_OK) rc = alloc_alt(c);
    if (rc != SVGF_OK) return rc;
    const int P = c->pingpong;
    // frames in flight: this frame's pair of filter planes (the previous frame's result sits in the other one until the call after this)
    void** const F = c->frames_in_flight > 1 && c->filter_set ? c->filter_alt : c->filter;
    c->last_pair_alt = F == c->filter_alt;
    if (c->frames_in_flight > 1) c->filter_set ^= 1;

    svgf_ctx::FrameEvents fe;
    const bool timed = !cap && c->timing > 0 && (c->timing_phase++ % c->timing) == 0;     // (events of a captured frame are graph nodes: nothing to read back)
    auto stamp = [&]() {                            // on c->stream AS IT IS when called (the side stream for the tail of a frame in flight)
        if (!timed) return;
        hipEvent_t e = take_event(c);
        if (e && hipEventRecord(e, c->stream) == hipSuccess) fe.ev.push_back(e);
        else if (e) c->pool.push_back(e);
    };
    auto bail = [&](int code) {                     // an error path hands the events already taken back to the pool
        for (auto e : fe.ev) c->pool.push_back(e);
        fe.ev.clear();
        if (code != SVGF_OK) c->guide_prev_valid = false;      // a frame that failed half-way leaves no guide plane to trust
        return code;
    };

    stamp();
    if (c->debug_mode != SVGF_DEBUG_FINAL) {
        // SVGFDebugOutput::TemporalFilter (App.cu:602-609): the temporal stage alone, its result is the frame.
        // SVGFDebugOutput::ATrousWaveletFilter / Depth (App.cu:611-620, 632-638): temporal, then the wavelet filter WITHOUT
        // FilterMoments — FilterBuffer[0] still holds the previous frame's result (App. B #11), which is what gets filtered.
        rc = temporal_impl(c, c->colour[1 - P], radiance, c->colour[P], cur, prev, c->hist[1 - P], c->hist[P], c->moments[P], c->moments[1 - P], nullptr, 0);
        if (rc != SVGF_OK) return bail(rc);
        stamp(); stamp();
        const void* res = c->colour[P];
        if (c->debug_mode == SVGF_DEBUG_ATROUS) {
            int pp = c->result_index;
            for (int i = 0; i < c->p.steps; i++) {
                rc = atrous_impl(c, c->filter[pp], c->filter[1 - pp], c->colour[P], cur, 1 << i, i);
                if (rc != SVGF_OK) return bail(rc);
                stamp();
                pp ^= 1;
            }
            c->result_index = pp;
            res = c->filter[pp];
        }
        if (timed) {
            fe.nstage = (int)fe.ev.size() - 1;
            if (fe.nstage >= 2) c->pending.push_back(std::move(fe)); else bail(0);
        }
        if (result) *result = res;
        commit_guide(c, cur, false);
        c->pingpong ^= 1;
        if (c->frames_since_reset < (1 << 30)) c->frames_since_reset++;
        return SVGF_OK;
    }
    // With at least one wavelet iteration the temporal result in colour[P] is dead where iteration 0's feedback will
    // overwrite it (:619-622): it is only stored for young pixels (the moments estimate reads them) and depth-0 texels.
    bool cold = false, crowded = false;
    choose_moments_kernel(c, &cold, &crowded);
    // (a crowded frame keeps every pixel's temporal colour: the streaming kernel reads its taps from one plane)
    const int sparse = c->p.steps >= 1 && !crowded;
    // The temporal launch also writes the filter buffer where history >= 4 (there FilterMoments is a copy), the
    // moments launch then only works on young pixels: same planes, 32 B/px less traffic in steady state.
    // ... and repacks what the wavelet iterations read of the G-buffer ({depth, ddepth, normal}: 16 B instead of 24 B of lines per
    // pixel and iteration) into the guide plane: +16 B/px here, -8 B/px in each iteration
    // (measured, tools/abn.sh on one device: -1.5 % per 4K fp32 frame; with fp16 storage the iterations alone gained less than the
    // temporal launch lost — since the NEXT frame's reprojection test reads the plane too (below) it pays there as well: -5.7 %)
    // ... and the NEXT frame's reprojection test reads this frame's guide plane instead of the three planes of its previous
    // G-buffer (prev_guide_for): -16 B/px of the temporal launch's 146
    void* guide = use_guide(c) ? c->guide : nullptr;
    rc = temporal_impl(c, c->colour[1 - P], radiance, c->colour[P], cur, prev, c->hist[1 - P], c->hist[P],
                       c->moments[P], c->moments[1 - P], F[0], sparse, guide, prev_guide_for(c, cur, prev));  // App.cu:552
    c->dense_now = c->cold_now = false;             // (the stage calls on this context keep their lists)
    if (rc != SVGF_OK) return bail(rc);
    stamp();
    // the first three frames after a reset have history <= 3 everywhere: the LDS-streaming moments kernel
    rc = moments_impl(c, c->colour[P], F[0], c->moments[P], cur, c->hist[P], 1, cold || crowded, sparse);   // App.cu:554 (current moments: App. B #4)
    if (rc != SVGF_OK) return bail(rc);
    stamp();
    int pp = 0, first = 0;
    hipStream_t const caller_stream = c->stream;
    // The pair launch stores iteration 0's feedback on [rb - 4, re + 4) inside the frame: with svgf_set_rows narrower than the frame those
    // rows lie outside what this frame's temporal launch wrote, and feedback computed from last frame's filter plane would replace
    // colour rows the two-launch sequence leaves alone.  So: only on the whole frame.
    const bool pair = can_fuse01(c) && halo_held(c, 6) && c->rb == 0 && c->re == c->H;
    bool aside = false;
    auto go_aside = [&]() -> int {
        // the rest of this frame goes onto the side stream; the frame that was there is ordered on the caller's stream first (its
        // result may be consumed, its planes reused)
        hipError_t e = hipEventRecord(c->ev_first, caller_stream);
        if (e != hipSuccess) return hip_fail(c, e, "hipEventRecord");
        int r = join_side(c, caller_stream);
        if (r != SVGF_OK) return r;
        e = hipStreamWaitEvent(c->side, c->ev_first, 0);
        if (e != hipSuccess) return hip_fail(c, e, "hipStreamWaitEvent");
        c->stream = c->side;
        aside = true;
        stamp();                                                                // the tail's own start: it may have waited for the frame before it
        return SVGF_OK;
    };
    if (pair) {
        // iterations 0 and 1 as one launch: F[0] -> F[1] (iteration 0's own plane is never written), feedback as ever
        rc = atrous_pair_impl(c, F[0], F[1], c->colour[P], cur, guide);
        if (rc == SVGF_OK) { stamp(); stamp(); pp = 1; first = 2; }             // timing slot 2 holds the pair, slot 3 (next to) nothing
    } else if (c->p.steps >= 1) {
        rc = atrous_impl(c, F[0], F[1], c->colour[P], cur, 1, 0, guide);        // App.cu:497-507, iteration 0: feeds the history back
        if (rc == SVGF_OK) { stamp(); pp = 1; first = 1; }
    }
    // Everything the NEXT frame's temporal launch reads is written now: with two frames in flight the remaining iterations leave the
    // caller's stream.  (Iteration 0 on the side stream as well - the next temporal launch waiting for an event behind it - measured
    // 1-2 % slower: beside a queue of nothing but wavelet launches the temporal launch gets too few workgroup slots, profiles/r03_small_experiments.txt.)
    // The tail may only leave the caller's stream if it reads nothing of the caller's: every remaining iteration an LDS launch on the
    // guide plane.  An iteration the direct kernel runs (variant DIRECT, PhiNormal == 0, a step beyond 64) reads cur->motion / cur->normal,
    // which nothing orders against the caller's stream once the call has returned.
    bool tail_reads_cur = guide == nullptr || c->p.variant == SVGF_VARIANT_DIRECT || c->p.phi_normal == 0.0f;
    for (int i = first; i < c->p.steps; i++) tail_reads_cur = tail_reads_cur || (1 << i) > 64;
    if (rc == SVGF_OK && c->frames_in_flight > 1) {
        if (c->side && first < c->p.steps && !tail_reads_cur) { fe.split = 2 + first; rc = go_aside(); }
        else rc = join_side(c, caller_stream);
    }
    for (int i = first; i < c->p.steps && rc == SVGF_OK; i++) {
        rc = atrous_impl(c, F[pp], F[1 - pp], c->colour[P], cur, 1 << i, i, guide);
        if (rc == SVGF_OK) { stamp(); pp ^= 1; }
    }
    if (aside) {
        // (also after a failed launch: whatever did get onto the side stream must be waited for before its planes are touched again)
        hipError_t e = hipEventRecord(c->ev_done, c->side);
        if (e == hipSuccess) { c->in_flight = true; c->in_flight_capture = cap; } else if (rc == SVGF_OK) rc = hip_fail(c, e, "hipEventRecord");
        c->stream = caller_stream;
    }
    if (rc != SVGF_OK) return bail(rc);
    if (timed) {
        fe.nstage = 2 + c->p.steps;
        if ((int)fe.ev.size() == fe.nstage + 1 + (aside ? 1 : 0)) c->pending.push_back(std::move(fe));
        else bail(0);
    }
    if (result) *result = F[pp];
    c->result_index = pp;
    commit_guide(c, cur, guide != nullptr);
    c->pingpong ^= 1;                                                           // App.cu:374
    if (c->frames_since_reset < (1 << 30)) c->frames_since_reset++;
    return SVGF_OK;
}

void* svgf_state_plane(svgf_ctx* c, int plane, int index) {
    if (!c || !c->have_state || index < 0 || index > 1) return nullptr;
    switch (plane) {
        case SVGF_PLANE_COLOUR: return c->colour[index];
        case SVGF_PLANE_MOMENTS: return c->moments[index];
        case SVGF_PLANE_FILTER: return c->filter[index];
        case SVGF_PLANE_HISTORY: return c->hist[index];
        default: return nullptr;
    }
}

int svgf_state_pingpong(const svgf_ctx* c) { return c ? c->pingpong : 0; }

size_t svgf_plane_bytes(const svgf_ctx* c, int plane) {
    if (!c) return 0;
    switch (plane) {
        case SVGF_PLANE_COLOUR: case SVGF_PLANE_FILTER: return colour_bytes(c);
        case SVGF_PLANE_MOMENTS: return moments_bytes(c);
        case SVGF_PLANE_HISTORY: return hist_bytes(c);
        default: return 0;
    }
}

int svgf_get_size(const svgf_ctx* c, int* width, int* height, svgf_strip* strip) {
    if (!c) return SVGF_ERR_INVALID;
    if (width) *width = c->W;
    if (height) *height = c->H;
    if (strip) *strip = c->strip;
    return SVGF_OK;
}

int svgf_halo_violations(svgf_ctx* c, unsigned long long* count, int clear) {
    if (!c || !count) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    return read_halo_violations(c, count, clear);
}

// Wait for everything enqueued on the context's stream and report what the kernels could only count: a strip whose
// temporal stage reprojected into rows it does not hold (motion beyond the state halo) is no longer the whole frame's result.
int svgf_sync(svgf_ctx* c) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    unsigned long long n = 0;
    int rc = join_side(c, c->stream);
    if (rc == SVGF_OK) rc = read_halo_violations(c, &n, 1);
    if (rc != SVGF_OK) return rc;
    if (!c->halo_violations) SVGF_HIP(c, hipStreamSynchronize(c->stream));
    if (n) return fail(c, SVGF_ERR_HALO, "temporal stage: " + std::to_string(n) + " reprojection(s) landed inside the frame but outside the rows this strip holds "
                                                                                "(motion larger than the state halo): the strip differs from the whole frame there");
    return SVGF_OK;
}

// Diagnostics: which tap path the waves of the LDS-streaming a-trous launches of this context take (svgf_kernels.h: AtrousArgs::path_stats).
int svgf_path_stats_enable(svgf_ctx* c, int on) {
    if (!c) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    SVGF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->side) SVGF_HIP(c, hipStreamSynchronize(c->side));
    if (on && !c->path_stats) {
        SVGF_HIP(c, hipMalloc((void**)&c->path_stats, 2 * SVGF_PATH_STAT_STEPS * sizeof(unsigned long long)));
        SVGF_HIP(c, hipMemset(c->path_stats, 0, 2 * SVGF_PATH_STAT_STEPS * sizeof(unsigned long long)));
    } else if (!on && c->path_stats) {
        (void)hipFree(c->path_stats);
        c->path_stats = nullptr;
    }
    return SVGF_OK;
}

int svgf_path_stats_read(svgf_ctx* c, unsigned long long* counts, int slots) {
    if (!c || !counts || slots <= 0) return SVGF_ERR_INVALID;
    if (!c->path_stats) return fail(c, SVGF_ERR_INVALID, "svgf_path_stats_read: not enabled (svgf_path_stats_enable)");
    DeviceGuard dg(c->device);
    SVGF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->side) SVGF_HIP(c, hipStreamSynchronize(c->side));
    unsigned long long h[2 * SVGF_PATH_STAT_STEPS];
    SVGF_HIP(c, hipMemcpy(h, c->path_stats, sizeof(h), hipMemcpyDeviceToHost));
    SVGF_HIP(c, hipMemset(c->path_stats, 0, sizeof(h)));
    for (int i = 0; i < slots; i++) counts[i] = i < 2 * SVGF_PATH_STAT_STEPS ? h[i] : 0ull;
    return SVGF_OK;
}

int svgf_timing_enable(svgf_ctx* c, int on) {
    if (!c) return SVGF_ERR_INVALID;
    c->timing = on > 0 ? on : 0;
    c->timing_phase = 0;
    return SVGF_OK;
}

int svgf_timing_read(svgf_ctx* c, double* ms_sum, int* frames, int slots) {
    if (!c || !ms_sum || slots <= 0) return SVGF_ERR_INVALID;
    DeviceGuard dg(c->device);
    for (auto& f : c->pending) {
        SVGF_HIP(c, hipEventSynchronize(f.ev.back()));
        for (int i = 0; i < f.nstage; i++) {
            float ms = 0.f;
            const int b = i >= f.split ? i + 1 : i;
            SVGF_HIP(c, hipEventElapsedTime(&ms, f.ev[b], f.ev[b + 1]));
            c->ms_sum[i] += ms;
        }
        for (auto e : f.ev) c->pool.push_back(e);
        c->timed_frames++;
    }
    c->pending.clear();
    for (int i = 0; i < slots; i++) ms_sum[i] = i < 2 + SVGF_MAX_STEPS ? c->ms_sum[i] : 0.0;
    if (frames) *frames = c->timed_frames;
    std::memset(c->ms_sum, 0, sizeof(c->ms_sum));
    c->timed_frames = 0;
    return SVGF_OK;
}

}  // extern "C"
