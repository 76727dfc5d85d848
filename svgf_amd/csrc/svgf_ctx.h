// Internal: the context behind the C ABI and the helpers shared by svgf_api.hip (stages, frame driver, lifecycle) and
// svgf_strip.hip (the multi-GPU strip driver).
#pragma once
#include "../../include/svgf.h"
#include "../../include/svgf_ext.h"
#include "../../include/svgf_test.h"
#include "svgf_kernels.h"

#include <string>
#include <vector>

struct svgf_strip_driver;

#ifndef SVGF_PREV_GUIDE_DEFAULT
#define SVGF_PREV_GUIDE_DEFAULT 0      // opt-in (svgf_set_prev_guide): it relies on the host not rewriting the previous G-buffer's planes
#endif

#ifndef SVGF_FUSE01_DEFAULT
#define SVGF_FUSE01_DEFAULT 0         // off: the pair launch is bit-identical and ~10 % slower than two launches
#endif

struct svgf_ctx {
    int W = 0, H = 0;
    svgf_strip strip{};
    int rb = 0, re = 0;                 // active compute rows (global)
    svgf_params p{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // context-owned state (frame driver): RenderBuffer[2], MomentsBuffer[2], FilterBuffer[2] (App.h:138-140)
    // and the ping-ponged history plane (App.h:141 + SURVEY App. B #1)
    void* colour[2] = {nullptr, nullptr};
    void* moments[2] = {nullptr, nullptr};
    void* filter[2] = {nullptr, nullptr};
    uint8_t* hist[2] = {nullptr, nullptr};
    void* guide = nullptr;                 // {depth, ddepth, normal, instance ID} of the current G-buffer repacked by the temporal launch for the wavelet iterations
    void* guide_prev = nullptr;            // ... and the plane the previous frame wrote (the two swap at the end of a frame): the next reprojection test reads it
    svgf_gbuffer guide_prev_of{};          // the G-buffer guide_prev was made from (the planes' addresses), valid while guide_prev_valid
    bool guide_prev_valid = false;
    bool prev_guide_enabled = SVGF_PREV_GUIDE_DEFAULT != 0;   // svgf_set_prev_guide
    bool fuse01 = SVGF_FUSE01_DEFAULT != 0;   // svgf_set_iteration_fusion: iterations 0 and 1 in one launch (frame / strip drivers)
    // svgf_set_frames_in_flight(2): the frame driver runs iterations 1.. of a frame on `side` while the NEXT frame's temporal, moments
    // and first-iteration launches run on `stream` (the HBM-bound launch beside the arithmetic-bound ones).  Frames then alternate
    // between two pairs of filter planes (the guide planes alternate anyway), and a frame's result is ordered on `stream` by the next
    // svgf_denoise_frame / svgf_flush / svgf_sync.
    int frames_in_flight = 1;
    hipStream_t side = nullptr;
    hipEvent_t ev_first = nullptr, ev_done = nullptr;   // iteration 0 of the frame being enqueued is on `stream`; the last iteration of the frame in flight is on `side`
    void* filter_alt[2] = {nullptr, nullptr};
    int filter_set = 0;                    // which pair the NEXT frame uses (toggles per frame while frames_in_flight == 2)
    bool last_pair_alt = false;            // the last frame wrote filter_alt[] (under its present name): what svgf_set_frames_in_flight(1) renames
    bool in_flight = false;                // a frame's tail is on `side` and `stream` has not been made to wait for it yet
    unsigned long long in_flight_capture = 0;   // ... and the stream capture that tail was recorded in (0: none; svgf.h, Stream capture)
    unsigned long long* young_masks = nullptr;   // scratch, temporal -> moments: per (row, 64-column segment) the lanes whose pixel (history < 4) needs the spatial estimate
    uint32_t* young_list = nullptr;        // ... and the indices of the pixels of the partly young segments (svgf::young_list_entries)
    unsigned long long* young_count = nullptr;   // ... two {appends, pixels} counters used in turn (the temporal launch of a frame zeroes the next frame's)
    unsigned* nan_count = nullptr;         // two device counters of nan_list used in turn (the temporal launch of a frame zeroes the next frame's)
    unsigned long long* sample_count = nullptr;   // two 64-bit device counters (128 B apart) used in turn: the sampled number of young pixels of a frame (TemporalArgs::sample_count)
    unsigned long long* estimate_host = nullptr;    // host-mapped: the latest sample a temporal launch has published (read without synchronising: some frames old)
    bool adaptive_moments = true;          // svgf_set_adaptive_moments
    bool dense_moments = false;            // the frame driver's current choice (hysteresis)
    bool dense_now = false;                // the frame being enqueued is served by the streaming kernel (a cold or a crowded frame)
    bool cold_now = false;                 // ... it is one of the first three after a reset
    uint32_t* nan_list = nullptr;          // scratch, temporal -> moments: the pixels whose accumulated colour / moments are NaN or inf (kNanListCap entries)
    int young_phase = 0;
    bool young_pending = false;            // a temporal launch wrote the masks / appended to nan_count[young_phase] and no moments launch has consumed them yet
    int vy0 = 0, vy1 = 0;                  // global rows of the previous-frame planes that hold valid state (svgf_set_valid_rows; default: all held)
    unsigned* halo_violations = nullptr;   // strips: device counter of reprojections that left the rows this strip holds (temporal_kernel)
    int pingpong = 0;                      // PingPongInx, App.cu:374
    int frames_since_reset = 0;
    int result_index = 0;                  // which filter plane holds the last result (the reference copies it back into FilterBuffer[0], App.cu:510-513)
    int debug_mode = SVGF_DEBUG_FINAL;     // SVGFDebugOutput, App.cu:545-649
    bool have_state = false;
    svgf_strip_driver* strip_drv = nullptr;
    unsigned long long* path_stats = nullptr;   // svgf_path_stats_enable: device counters, a pair per step 1 << i (svgf_kernels.h: AtrousArgs::path_stats), or null
    // per-stage timing
    int timing = 0;               // 0 = off, n = stage events on every n-th frame
    int timing_phase = 0;
    // stage i ran between ev[i] and ev[i + 1]; from stage `split` on (the launches on the side stream) between ev[i + 1] and ev[i + 2]
    struct FrameEvents { std::vector<hipEvent_t> ev; int nstage = 0; int split = 1 << 30; };
    std::vector<FrameEvents> pending;
    std::vector<hipEvent_t> pool;
    double ms_sum[2 + SVGF_MAX_STEPS] = {0};
    int timed_frames = 0;
};

namespace svgf_host {

// Every entry point runs with the context's device current and restores the caller's on the way out: a host that drives
// several devices from one thread (or whose current device is not the context's) finds its own device unchanged.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) {
            switched = hipSetDevice(device) == hipSuccess;
            if (!switched) (void)hipGetLastError();        // (a device that does not exist: the caller's entry point fails on its own; nothing is left pending for the host)
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int fail(svgf_ctx* c, int code, const std::string& msg);
int hip_fail(svgf_ctx* c, hipError_t e, const char* what);
#define SVGF_HIP(c, call)                                                    \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return svgf_host::hip_fail((c), e_, #call);    \
    } while (0)

size_t colour_bytes(const svgf_ctx* c);
size_t moments_bytes(const svgf_ctx* c);
size_t hist_bytes(const svgf_ctx* c);
bool is_strip(const svgf_ctx* c);
int reset_history(svgf_ctx* c);
int alloc_state(svgf_ctx* c);
int alloc_flags(svgf_ctx* c);
int read_halo_violations(svgf_ctx* c, unsigned long long* count, int clear);
int join_side(svgf_ctx* c, hipStream_t onto);   // `onto` waits for the frame in flight on the side stream (frames_in_flight == 2); no-op otherwise

// the stages on caller- or driver-owned planes, rows [c->rb, c->re); the device is already current
int temporal_moments_impl(svgf_ctx* c, const void* prev_colour, const void* radiance, void* colour_out, void* filter_out,
                          const svgf_gbuffer* cur, const svgf_gbuffer* prev, const uint8_t* hist_prev, uint8_t* hist_cur,
                          void* moments_cur, const void* moments_prev, int mrb, int mre, int feedback_follows, void* guide_out = nullptr,
                          const void* guide_prev = nullptr, int dense = 0);
void choose_moments_kernel(svgf_ctx* c, bool* cold, bool* crowded);   // frame / strip drivers, before the temporal launch of a frame (svgf_api.hip)
int atrous_impl(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g, int step, int iteration, const void* guide = nullptr);
// one iteration over several row ranges in one launch, the first ranges signalled (the strip driver's edge rows; svgf_kernels.h: AtrousRanges)
bool atrous_ranges_ok(const svgf_ctx* c, int step);
int atrous_ranges_impl(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g, int step, int iteration, const void* guide, const svgf::AtrousRanges& r);
// iterations 0 and 1 in one launch on rows [c->rb, c->re) (iteration 1's; iteration 0 and the feedback store cover 4 more rows either side)
int atrous_pair_impl(svgf_ctx* c, const void* in, void* out, void* feedback, const svgf_gbuffer* g, const void* guide = nullptr);
bool can_fuse01(const svgf_ctx* c);     // the drivers run iterations 0 and 1 as one launch
const void* prev_guide_for(const svgf_ctx* c, const svgf_gbuffer* cur, const svgf_gbuffer* prev);   // the guide plane that stands in for `prev`, or null
void commit_guide(svgf_ctx* c, const svgf_gbuffer* cur, bool written);   // end of a frame: the guide just written (the temporal launch covers all held rows) becomes the previous one
bool use_guide(const svgf_ctx* c);      // the frame / strip drivers repack {depth, ddepth, normal, instance ID} for the iterations (any storage, >= 1 iteration, LDS kernels)

}  // namespace svgf_host
