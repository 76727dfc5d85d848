// Internal launcher interface between the C-ABI layer (svgf_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svgf {

// Local planes hold global rows [y0, y0+rows) of a W x H frame; a launch computes rows [yb, ye).
struct Geo { int W, H, y0, rows, yb, ye; };

// The list of pixels whose temporal result is not finite holds at most kNanListCap entries (more than that and the moments launch goes over
// every pixel instead); its two counters are used in turn, frame by frame.
constexpr unsigned kNanListCap = 1u << 16;
// The young-pixel list takes a bounded number of appends per frame (one per wave that holds SOME young pixels).  Same-address atomics retire at
// ~11 ns each on this part (tools/ubench/atomic_one_address.hip), one after the other: the bench pan's 8 000 - 10 000 such waves are 0.1 ms spread
// over a 0.19 ms 4K launch — unseen — but its 4 500 at 1080p are 0.05 ms in a 0.043 ms launch, and a frame in which every wave holds a young pixel
// made the 4K launch a 1.39 ms one.  So: the list is kYoungShards lists with a counter each on a line of its own (a workgroup's waves go to shard
// (blockIdx.x + blockIdx.y) mod kYoungShards: a column and a row of such waves both spread over all of them), and each takes at most an eighth of
// young_append_cap = a quarter of the waves of the rows the context holds (4K: 32 400).  A frame with more — thin geometry under motion — stops
// appending (a few thousand waves late: those already past the test when a shard's cap is reached), and the moments launch works from the per-
// segment lane masks instead.  A counter is 64 bits: {appends, pixels}.
constexpr int kYoungShards = 8;
inline unsigned young_append_cap(int rows, int W) {
    const long long waves = (long long)rows * ((W + 63) / 64);
    const long long cap = waves / 4 > 1024 ? waves / 4 : 1024;
    return (unsigned)(cap / kYoungShards * kYoungShards);
}
inline size_t young_list_entries(int rows, int W) { return (size_t)young_append_cap(rows, W) * 63; }   // shard s: entries [s, s + 1) x cap / kYoungShards x 63
// "A cap is reached" is a word of its own: a wave reads THAT before it appends — a load of a counter's own line between the atomics makes each of
// them cost 50 instead of 11 ns, a load of a line nobody writes is free.  A context holds two sets {kYoungShards counters, flag}, each word on a
// 128-byte line of its own: counter s at s * kYoungLine, the flag at kYoungFlagOffset, the sets kYoungCounterStride 64-bit words apart.
constexpr int kYoungLine = 16, kYoungFlagOffset = kYoungShards * kYoungLine, kYoungCounterStride = (kYoungShards + 1) * kYoungLine;

struct TemporalArgs {
    const void* prev_colour; const void* radiance; void* colour_out;
    const float4* motion_c; const uint2* normal_c; const uint2* uv_c;
    const float4* motion_p; const uint2* normal_p; const uint2* uv_p;
    const uint8_t* hist_prev; uint8_t* hist_cur; void* mom_cur; const void* mom_prev;
    float depth_thr, normal_thr; int history_base; int mesh_id_test;
    void* passthrough_out;   // frame driver only: where history >= 4 the moments stage is a copy (Filter.cuh:521) — write it here directly
    unsigned long long* young_masks;   // with passthrough_out: which pixels still need the moments estimate — one 64-bit lane mask per (local row,
                             // 64-column segment), stored by the wave that computes the segment ...
    uint32_t* young_list;    // ... and, for the waves that hold SOME young pixels (disocclusions are sparse: frame borders under a pan, silhouettes),
    unsigned long long* young_count;        // their local indices (row * W + x) appended to a list: one 64-bit atomic per wave on {appends, pixels}, at most
    unsigned long long* young_count_next;   // young_cap of them per frame (above).  The OTHER counter of the context's pair is zeroed by this launch.
    unsigned young_cap;
    int sparse_colour;       // with passthrough_out and >= 1 a-trous iteration: colour_out is only stored where the iteration-0 feedback
                             // will not overwrite it or the moments estimate reads it (young pixels, depth-0 texels)
    int sky_zero;            // with passthrough_out: PhiNormal > 0, so a young pixel with an all-zero normal filters to exactly 0 (written here)
    unsigned* halo_violations;   // strips only: counts reprojections that land inside the frame but outside the valid rows of the strip
    int valid_lo, valid_hi;      // local rows [valid_lo, valid_hi) of the previous-frame planes hold valid state (a strip allocates more rows than
                                 // it keeps up to date: the a-trous halos are wider than the state halo)
    uint4* guide_out;            // frame / strip drivers: {depth, ddepth, (nx,ny) half bits, (nz, instance ID) half bits} of the CURRENT G-buffer,
                                 // 16 B per pixel — all the wavelet iterations read of it (24 B per pixel in two planes otherwise), or null
    const uint4* guide_prev;     // the guide_out plane of the frame whose current G-buffer is this frame's previous one: read instead of
                                 // motion_p / normal_p / uv_p (16 instead of 32 B per pixel), or null
    int guide_lo, guide_hi;      // global rows [guide_lo, guide_hi) whose guide texel this launch writes (>= the compute rows of Geo: a strip
                                 // needs the texels of every row it holds)
    uint32_t* nan_list;          // with young_masks: the local indices of the pixels whose temporal colour / moments are NaN or inf, appended with one
    unsigned* nan_count;         // atomic per wave that holds any (none in a frame without a NaN); the moments launch redoes the zero-normal shortcut
    unsigned* nan_count_next;    // pixels around them.  The OTHER counter of the context's pair is zeroed by this launch for the next frame
    unsigned long long* sample_count;   // with young_masks: every 64th wave (hashed) adds the number of its young pixels here, fire and forget — what the frame
    unsigned long long* sample_prev;      // driver's choice between the young-pixel launch and the streaming kernel goes by (svgf_set_adaptive_moments).  This
    unsigned long long* estimate_host;    // launch publishes the PREVIOUS frame's sum (sample_prev, then zeroed) in host-mapped memory; all three may be null
    int sample_off;              // a frame after a reset (every pixel young by construction): publishes, but adds nothing — its sample would say "crowded"
                                 // two frames later, when nothing is
    int heal_nan;                // svgf_params::nan_policy == SVGF_NAN_ZERO: a NaN channel of the radiance / previous colour / previous moments reads as 0
};
struct MomentsArgs {
    const void* colour; void* out; const void* mom; const float4* motion; const uint2* normal; const uint8_t* hist;
    float phi_colour, phi_normal; int radius;
    int cold_only;           // 1: pixels with history >= 4 were already written by the temporal stage (passthrough_out)
    int dense;               // 1: (nearly) every pixel has history < 4 (first frames of a sequence): use the LDS-streaming kernel
    int sparse_colour;            // TemporalArgs::sparse_colour of the same frame: an old, non-sky neighbour's colour is in `out`
    const unsigned long long* young_masks;   // with cold_only: TemporalArgs::young_masks / young_list / young_count of the same frame — only those pixels are visited
    const uint32_t* young_list;
    const unsigned long long* young_count;
    unsigned young_cap;
    const uint32_t* nan_list;     // TemporalArgs::nan_list / nan_count of the same frame
    const unsigned* nan_count;
    int no_fastpath;              // SVGF_VARIANT_LDS_GENERAL: the LDS-streaming kernel without its uniform-normal form (bit-identical, slower)
};
struct AtrousArgs {
    const void* in; void* out; void* feedback; const float4* motion; const uint2* normal;
    int step; float phi_colour, phi_normal;
    const uint4* guide;          // TemporalArgs::guide_out of the same frame (LDS kernel only; motion / normal are then not read), or null
    int no_fastpath;             // SVGF_VARIANT_LDS_GENERAL: every wave takes the general tap path (bit-identical, slower)
    unsigned long long* path_stats;   // svgf_path_stats_enable (diagnostics; LDS-streaming kernel): {wave-steps that filtered a surface pixel, those of them on the
                                 // uniform-normal tap path} of this launch are ADDED here, one pair of atomics per wave and band; null: nothing is counted out
};
// Strip driver: ONE launch over up to three row ranges of an iteration, the first `nfirst` of them — the rows a neighbour rank waits for —
// produced by the launch's first workgroups; the last of those to finish publishes `value` in `signal` (device memory), which the
// communication stream waits for with hipStreamWaitValue64: the halo exchange starts while the interior tiles of the same launch still
// run.  (Round 4 launched the two edge ranges and the interior separately: three launches' ramp and tail per iteration.)  Every range is
// cut into bands by itself, so a pixel's result is what a launch over its range alone would give: bit-identical.
struct AtrousRanges {
    int n, nfirst;               // ranges; how many of them come first and signal
    int yb[3], ye[3];            // global rows
    unsigned long long* signal;  // 8 bytes, written once per launch by the last of the first ranges' workgroups (release, system scope) ...
    unsigned* arrivals;          // ... counted here (left at 0)
    unsigned long long value;
};

hipError_t launch_temporal(const Geo& g, int storage, const TemporalArgs& a, hipStream_t s);
hipError_t launch_moments(const Geo& g, int storage, const MomentsArgs& a, bool direct, hipStream_t s);
hipError_t launch_atrous(const Geo& g, int storage, int variant, const AtrousArgs& a, hipStream_t s);
bool atrous_ranges_available(int variant, const AtrousArgs& a);     // the LDS-streaming kernel serves this step (else: one launch_atrous per range)
hipError_t launch_atrous_ranges(const Geo& g, int storage, const AtrousArgs& a, const AtrousRanges& r, hipStream_t s);   // g.yb / g.ye are not used
// iterations 0 and 1 (steps 1 and 2) in ONE launch: `in` -> `out` is iteration 1's result, `feedback` iteration 0's (written on the launch rows
// and 4 rows beyond inside the frame); the launch rows [yb, ye) are iteration 1's and the planes hold 6 rows around them
bool atrous_fused_available(int variant, const AtrousArgs& a);
hipError_t launch_atrous_fused(const Geo& g, int storage, const AtrousArgs& a, hipStream_t s);
struct PackArgs {
    const float4* position; const float4* normal; const float4* bary;
    float vp[16], pvp[16], cam[3];
    float4* motion; uint2* normal_out; uint2* uv_out;
};
hipError_t launch_pack_gbuffer(const Geo& g, const PackArgs& a, hipStream_t s);
hipError_t launch_albedo(const Geo& g, int storage, int mode, const void* in, const void* albedo, void* out, hipStream_t s);
hipError_t launch_taa(const Geo& g, int storage, const void* filtered, const void* history, void* out, bool direct, hipStream_t s);

}  // namespace svgf
