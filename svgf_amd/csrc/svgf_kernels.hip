// svgf_kernels.hip — SVGF hot-path kernels for gfx950 (MI355X, CDNA4, wave64).
//
// Behavioural contract: src/Filter.cuh of jacquespillet/SVGF (TemporalFilter :359-404,
// FilterMoments :430-525, FilterKernel :527-624), restated in SURVEY.md Appendix A.
// Nothing here is translated from the reference: planes are linear device memory instead of
// texture objects, launches are wave64-row shaped (64 consecutive pixels of a row per wave, 16 B
// per lane per plane), and the edge-stopping weight is evaluated as ONE exp2 of a fused exponent.
//
// Built with -ffp-contract=off: FMAs appear only where written (fmaf), so the temporal stage and
// every accept/reject test round exactly like the scalar oracle (bit-exact parity), while the
// tap loops use explicit FMAs and the hardware exp2/log2/rcp (parity within a stated tolerance).

#include "svgf_kernels.h"

#include <algorithm>
#include <numeric>
#include <atomic>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

// The streaming a-trous kernels live in headers of their own (shared tap code: svgf_atrous_taps.h).
#include "svgf_atrous_lds.h"
#include "svgf_atrous_fused.h"
#include "svgf_moments_lds.h"

namespace svgf {
namespace {


// ------------------------------------------------------------------ temporal ------------------
// Filter.cuh:359-404 + LoadPreviousData :225-258.  All previous-frame gathers are issued before the
// accept/reject tests are evaluated (one round of latency instead of the reference's chain of seven
// dependent fetches); rejected pixels simply discard them.
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void temporal_kernel(Geo g, TemporalArgs a) {
    keep_nan_in_clamps();                                             // imageLoad / imageStore keep a NaN (svgf_device.h)
    if (a.young_masks && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && threadIdx.y == 0) {
#pragma unroll
        for (int sh = 0; sh <= kYoungShards; sh++) a.young_count_next[sh * kYoungLine] = 0ull;                 // (the counters and the flag)
        a.nan_count_next[0] = 0u;
        if (a.sample_count) {                       // last frame's sample is final: to the host (no answer awaited), and its counter starts again
            __hip_atomic_store(a.estimate_host, a.sample_prev[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            a.sample_prev[0] = 0ull;
        }
    }
    // The grid covers the compute rows [yb, ye) and, where a guide plane is written, the rows [guide_lo, guide_hi) around them (a strip
    // holds more rows than it runs the temporal stage on: the later iterations' halos and the next frame's reprojection read their
    // guide texels too): a row outside the compute rows gets its guide texel and nothing else (a wave is one row: no divergence).
    const int ylo = a.guide_out ? min(g.yb, a.guide_lo) : g.yb, yhi = a.guide_out ? max(g.ye, a.guide_hi) : g.ye;
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = ylo + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= yhi) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    if (y < g.yb || y >= g.ye) {
        a.guide_out[idx] = guide_texel(a.motion_c[idx], a.normal_c[idx], a.uv_c[idx]);
        return;
    }

    const float4 mc = a.motion_c[idx];
    const int qx = add_wrap(x, cvt_rzi_sat(mc.x)), qy = add_wrap(y, cvt_rzi_sat(mc.y));   // :232, CUDA's float -> int: toward zero, saturating, NaN -> 0
    bool ok = qx >= 0 && qx < g.W && qy >= 0 && qy < g.H;             // :235
    const int ql = add_wrap(qy, -g.y0);                               // (only used when `ok`: qy is inside the frame then)
    // Strip guard: never read outside the local planes.  A reprojection that lands inside the FRAME but outside the rows
    // this strip holds would silently turn into a rejection (history reset) and the strip would no longer equal the whole
    // frame: it is counted, and the host reports SVGF_ERR_HALO at its next synchronising call (svgf_sync).
    const bool in_strip = ql >= a.valid_lo && ql < a.valid_hi;      // [valid_lo, valid_hi) lies inside [0, rows)
    if (a.halo_violations) {
        const unsigned long long lost = __ballot(ok && !in_strip);
        if (lost != 0ull && threadIdx.x == (unsigned)__builtin_ctzll(lost)) atomicAdd(a.halo_violations, (unsigned)__builtin_popcountll(lost));
    }
    ok = ok && in_strip;
    const size_t q = ok ? (size_t)ql * g.W + qx : idx;

    float4 c = clamp01_ref(Store<ST>::ld4(a.radiance, idx));          // :370 imageLoad (a NaN stays NaN)
    auto healed = [](float v) { return v != v ? 0.0f : v; };          // SVGF_NAN_ZERO (an extension, svgf.h): a NaN channel reads as 0
    if (a.heal_nan) c = make_float4(healed(c.x), healed(c.y), healed(c.z), healed(c.w));
    const uint2 nc_raw = a.normal_c[idx];
    const uint2 uc_raw = a.uv_c[idx];
    // What the test needs of the PREVIOUS G-buffer is {depth, normal, instance ID} at q.  The drivers kept exactly that when
    // that G-buffer was the current one (guide_out of the previous frame, 16 B) and pass it instead of the three planes (32 B).
    float4 mp;
    uint2 np_raw, up_raw;
    if (a.guide_prev) {
        const uint4 gp = a.guide_prev[q];
        mp = make_float4(0.f, 0.f, __uint_as_float(gp.x), __uint_as_float(gp.y));
        np_raw = make_uint2(gp.z, gp.w);             // normal_of() reads the low half of .y only
        up_raw = make_uint2(0u, gp.w);               // instance ID: high half of .y, where the uv plane has it
    } else {
        mp = a.motion_p[q];
        np_raw = a.normal_p[q];
        up_raw = a.uv_p[q];
    }
    float4 pc = clamp01_ref(Store<ST>::ld4(a.prev_colour, q));        // :254 imageLoad
    const int hp = a.hist_prev[q];                                    // :255
    float2 pm = Store<ST>::ld2(a.mom_prev, q);                        // :256
    if (a.heal_nan) { pc = make_float4(healed(pc.x), healed(pc.y), healed(pc.z), healed(pc.w)); pm = make_float2(healed(pm.x), healed(pm.y)); }

    float zc, dzc, zp, dzp;
    depth_of(mc, zc, dzc);
    depth_of(mp, zp, dzp);
    ok = ok && !(fabsf(zp - zc) > a.depth_thr);                       // :242
    if (a.mesh_id_test) {                                             // :245-247 (intended test)
        const int idc = cvt_rzi_sat(unpack_h2(uc_raw.y).y), idp = cvt_rzi_sat(unpack_h2(up_raw.y).y);
        ok = ok && idc == idp;
    }
    ok = ok && !(dot3_exact(normal_of(nc_raw), normal_of(np_raw)) < a.normal_thr);   // :252

    int h = 1;
    float alpha = 1.0f;                                               // :385-386
    float3 cp = make_float3(0.f, 0.f, 0.f);
    float2 mprev = make_float2(0.f, 0.f);
    if (ok) {
        h = min(a.history_base, hp + 1);                              // :380
        alpha = 1.0f / (float)h;                                      // :381 (correctly rounded; == float(1.0/h) for h <= 255)
        cp = make_float3(pc.x, pc.y, pc.z);
        mprev = pm;
    }
    const float L = lum_exact(c.x, c.y, c.z);                         // :391
    float2 m = make_float2(mix_exact(mprev.x, L, alpha), mix_exact(mprev.y, L * L, alpha));   // :392-393
    const float var = fmaxf(0.0f, m.y - m.x * m.x);                   // :396
    const float4 o = make_float4(mix_exact(cp.x, c.x, alpha), mix_exact(cp.y, c.y, alpha), mix_exact(cp.z, c.z, alpha), var);

    a.hist_cur[idx] = (uint8_t)h;                                     // :400
    if (a.guide_out) a.guide_out[idx] = guide_texel(mc, nc_raw, uc_raw);
    // :401 imageStore.  sparse_colour (frame driver): iteration 0 of the wavelet filter overwrites this texel with its
    // feedback (:619-622) unless it has no depth; until then only the moments estimate of young pixels reads it
    // (the same predicate as the feedback store of atrous_*_kernel: GetDepth() == sentinel, i.e. depth 0 or literally 1e30f)
    const float4 oc = clamp01_ref(o);                                 // imageStore, :63-69
    if (!a.sparse_colour || h < 4 || zc == kSkyZ) Store<ST>::st4(a.colour_out, idx, oc);
    Store<ST>::st2(a.mom_cur, idx, m);                                // :402
    // Frame-driver fusion: for history >= 4 FilterMoments only copies this pixel into the filter buffer
    // (:521; store(load(x)) == x in both storage types), so it is written from here and the moments launch
    // touches nothing but the history byte of such pixels (-32 B/px of traffic in steady state).
    // A young pixel whose normal is exactly (0,0,0) — the G-buffer's cleared sky texels, which never pass the normal
    // test and stay young for ever — filters to exactly (0,0,0,0) when PhiNormal > 0 (see moments_pixel): written here too.
    const bool young = h < 4;
    const bool zero_young = young && a.sky_zero && ((nc_raw.x & 0x7fff7fffu) | (nc_raw.y & 0x7fffu)) == 0u;
    if (a.passthrough_out) {
        if (!young) Store<ST>::st4(a.passthrough_out, idx, oc);
        else if (zero_young) Store<ST>::st4(a.passthrough_out, idx, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    // ... and the moments launch is told where the remaining young pixels are.  Every wave (one 64-column segment of a row) stores its lane
    // mask; a wave that holds SOME young pixels also appends their indices to a list, one atomic per wave (disocclusions are sparse: frame
    // borders under a pan, silhouettes) — the list is dense in young pixels, so the moments launch spreads them evenly over its waves
    // however they are spread over the frame.  A wave whose 64 pixels are ALL young appends nothing (its mask says it all: after a reset every
    // wave is one), and no wave appends once a shard of the list is full (svgf_kernels.h: the moments launch then works from the masks alone).
    if (a.young_masks) {
        const bool listed = young && !zero_young;
        const unsigned long long ym = __ballot(listed);
        if (threadIdx.x == 0)                                             // (lane 0 is always inside the frame; lanes beyond W have left: their bits are 0)
            a.young_masks[(size_t)(y - g.y0) * ((g.W + kBX - 1) / kBX) + blockIdx.x] = ym;
        // one wave in 64 (hashed over segment and row: columns and rows of young pixels are sampled like anything else) reports how many it holds
        // (low word: young pixels; high word: waves that hold some but not 64 — the ones that append to the list.  One 64-bit atomic: packed into
        // 20 + 12 bits the wave field overflowed above 262 000 waves — an 8K frame has 518 000 — ADVICE r04)
        if (a.sample_count && !a.sample_off && ym != 0ull && threadIdx.x == 0 && (((unsigned)blockIdx.x * 29u + (unsigned)y * 13u) & 63u) == 0u)
            (void)__hip_atomic_fetch_add(a.sample_count, (unsigned long long)__builtin_popcountll(ym) + (ym != ~0ull ? 1ull << 32 : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.young_list && ym != 0ull && ym != ~0ull) {                 // (no list for a frame the streaming kernel will serve)
            const int lane = threadIdx.x, first = __builtin_ctzll(ym);
            unsigned base = ~0u;
            const unsigned shard = (blockIdx.x + blockIdx.y) % kYoungShards, shard_cap = a.young_cap / kYoungShards;
            if (lane == first && __hip_atomic_load(a.young_count + kYoungFlagOffset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
                const unsigned long long old = atomicAdd(a.young_count + shard * kYoungLine, (1ull << 32) | (unsigned long long)__builtin_popcountll(ym));
                if ((unsigned)(old >> 32) < shard_cap) base = shard * shard_cap * 63u + (unsigned)old;       // (appends 0 .. cap-1 own their entries: < cap x 63 pixels)
                else __hip_atomic_store(a.young_count + kYoungFlagOffset, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            base = __shfl(base, first);
            if (listed && base != ~0u) a.young_list[base + (unsigned)__builtin_popcountll(ym & ((1ull << lane) - 1ull))] = (uint32_t)idx;
        }
        // A pixel whose accumulated colour or moments are not finite (a NaN in the radiance, or in the history it reprojects onto:
        // the reference's clamps keep it, :63-83,398) is listed as well.  The shortcut above is only right while a zero-normal pixel's
        // 7x7 window is finite — its weights are exactly 0, and 0 x NaN is NaN in :498-499 — so the moments launch goes over the
        // windows of the listed pixels again (moments_young_kernel).  Nothing but the test while every pixel is finite.
        const unsigned long long bad = lanes_where(__builtin_isunordered(oc.x, oc.y)) | lanes_where(__builtin_isunordered(oc.z, m.x)) |
                                       lanes_where(!(fabsf(m.x + m.y) < __builtin_inff()));     // (oc.w: fmax(0, NaN) = 0, :396)
        if (bad != 0ull) {
            const int lane = threadIdx.x, first = __builtin_ctzll(bad);
            unsigned base = 0;
            // (once the list has overflowed the counter says so and stays: no wave appends any more — a frame full of NaN would otherwise
            // queue 130 000 atomics on one word)
            if (lane == first) base = __hip_atomic_load(a.nan_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > kNanListCap
                                          ? kNanListCap : atomicAdd(a.nan_count, (unsigned)__builtin_popcountll(bad));
            base = __shfl(base, first);
            const unsigned at = base + (unsigned)__builtin_popcountll(bad & ((1ull << lane) - 1ull));
            if (((bad >> lane) & 1ull) && at < kNanListCap) a.nan_list[at] = (uint32_t)idx;
        }
    }
}

// ------------------------------------------------------------------ moments -------------------
// Filter.cuh:430-525.  Steady state (h >= 4) is a plane copy; the (2R+1)^2 bilateral estimate runs
// only for young pixels.
template <int ST>
__device__ __forceinline__ void moments_pixel(const Geo& g, const MomentsArgs& a, int x, int y) {
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    const float h = (float)a.hist[idx];                               // :442
    if (a.cold_only && !(h < 4.0f)) return;                           // already written by temporal_kernel (passthrough_out)
    const float4 cc = Store<ST>::ld4(a.colour, idx);                  // :450 raw load
    if (!(h < 4.0f)) { Store<ST>::st4(a.out, idx, cc); return; }      // :521

    const float lc = lum_exact(cc.x, cc.y, cc.z);
    float zc, dzc;
    depth_of(a.motion[idx], zc, dzc);
    const uint2 nraw = a.normal[idx];
    const float3 nc = normal_of(nraw);
    // A centre whose normal is exactly (0,0,0) — the G-buffer's cleared sky texels — has n.n' = 0 for every tap, so with
    // phi_normal > 0 every weight is exp(..)*pow(0,phi_n) = 0 (edge_weight: exp2(-inf)): the sums are 0 x the taps, sumW clamps to
    // 1e-6 and the result is (0,0,0,0) (:505-516) — unless a tap is NaN or inf, whose product with 0 is NaN
    // (:498-499).  This kernel therefore takes no shortcut for such a centre; the frame driver's temporal launch does, and lists the
    // non-finite pixels so that the windows around them are redone (moments_young_kernel).
    const float il = hw_rcp(a.phi_colour);                            // :460
    const float phi_d = fmaxf(dzc, 1e-8f) * 3.0f;                     // :461
    float sw = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sm1 = 0.f, sm2 = 0.f;
    const int R = a.radius;
    // One window row at a time, every load of the row issued before anything is consumed: two rounds of memory latency per
    // row (the tap's colour comes from `out` or `colour` depending on its own motion / history texel) instead of two per tap —
    // young pixels are sparse in steady state, so a wave of them is latency-bound, not bandwidth-bound.  Same taps, same order.
    constexpr int RM = 3;                                             // radius <= 3 (svgf_params)
    for (int yy = -R; yy <= R; yy++) {
        const int py = y + yy;
        if (py < 0 || py >= g.H) continue;                            // :473
        const size_t rowp = (size_t)(py - g.y0) * g.W;
        bool ok[2 * RM + 1];
        size_t p[2 * RM + 1];
        float4 mq[2 * RM + 1], cp[2 * RM + 1];
        float2 mp[2 * RM + 1];
        uint2 nq[2 * RM + 1];
        int hq[2 * RM + 1];
#pragma unroll
        for (int k = 0; k <= 2 * RM; k++) {
            const int xx = k - RM, px = x + xx;
            ok[k] = xx >= -R && xx <= R && px >= 0 && px < g.W;
            p[k] = ok[k] ? rowp + px : idx;                           // a tap that does not exist reads the centre and is dropped
            mq[k] = a.motion[p[k]];
            mp[k] = Store<ST>::ld2(a.mom, p[k]);                      // :480
            nq[k] = a.normal[p[k]];                                   // :483
            hq[k] = (int)a.hist[p[k]];        // (unconditional: a load inside a branch is waited for at the end of the branch — one memory round per tap row)
        }
#pragma unroll
        for (int k = 0; k <= 2 * RM; k++) {
            // sparse_colour: the temporal launch stored an old, non-sky texel only into `out` (same value)
            const bool in_out = a.sparse_colour && mq[k].z != 0.0f && mq[k].z != kSkyZ && hq[k] >= 4;
            cp[k] = Store<ST>::ld4(in_out ? (const void*)a.out : a.colour, p[k]);   // :479 raw
        }
#pragma unroll
        for (int k = 0; k <= 2 * RM; k++) {
            if (!ok[k]) continue;
            const int xx = k - RM;
            float zp, dzp;
            depth_of(mq[k], zp, dzp);                                 // :482
            const float3 np = normal_of(nq[k]);
            const float len = sqrtf((float)(xx * xx + yy * yy));      // :488
            const float iz = (xx == 0 && yy == 0) ? 0.0f : hw_rcp(phi_d * len);   // phiDepth == 0 -> wZ = 0, :420
            const float w = edge_weight(fabsf(lc - lum_exact(cp[k].x, cp[k].y, cp[k].z)), il, fabsf(zc - zp), iz, dot3_fma(nc, np), a.phi_normal);
            sw += w;                                                  // :497-499
            sr = fmaf(cp[k].x, w, sr); sg = fmaf(cp[k].y, w, sg); sb = fmaf(cp[k].z, w, sb);
            sm1 = fmaf(mp[k].x, w, sm1); sm2 = fmaf(mp[k].y, w, sm2);
        }
    }
    sw = fmaxf(sw, 1e-6f);                                            // :505
    const float inv = 1.0f / sw;
    sm1 *= inv; sm2 *= inv;
    const float var = (sm2 - sm1 * sm1) * (4.0f / h);                 // :511-514
    Store<ST>::st4(a.out, idx, make_float4(sr * inv, sg * inv, sb * inv, var));   // :516 unclamped
}

// The same estimate for the pixels of the young list (disocclusions under a moving camera: a few tens of thousands of pixels along
// the frame borders and the silhouettes).  One thread per listed pixel made that launch the slowest per pixel of the frame: a wave of
// 64 scattered pixels touches ~60 different cache lines with EVERY one of its ~350 loads, and the launch is bound by the L1s' lookup
// rate (rocprofv3: 9.9 M line accesses for 164 K load instructions, 48 us; fewer rounds of latency or more CUs did not move it).
// Here EIGHT LANES serve one pixel: lane j < 7 owns window column j-3 and loads its seven taps (one per row) — the eight lanes of a
// group read adjacent texels of one row, one or two cache lines per load — and evaluates their weights; then every lane walks the 49
// taps in the reference's order (rows outer, columns inner, :467-500), fetching each tap's weight, colour and moments from the lane
// that holds them (__shfl = ds_bpermute: no memory), so the sums are accumulated in exactly the order — and to exactly the bits —
// of moments_pixel.  The colour of a tap is fetched from BOTH planes it can live in (the choice depends on the tap's own depth /
// history texel), which keeps every load of the window independent: one round of memory latency for the taps.
// ARITH = 0: the tap weight as moments_pixel evaluates it (variant DIRECT, a radius other than 3, PhiNormal == 0: the stage calls then run
// moments_pixel).  ARITH = 1: as moments_lds_kernel evaluates it (svgf_moments_lds.h, moments_taps49: the fused exponent on the same
// bits) — the kernel the default variants run for a frame full of young pixels, so that which of the two serves a frame is a matter of
// speed alone: the frame driver may switch between them from frame to frame without a bit of the results changing — next to NaN / inf
// texels too: both evaluate a pixel whose fused-exponent sums hold a NaN again with the reference's `max(term, 0.0)` (:424) for all its taps,
// and no other pixel.
template <int ST, int ARITH>
__device__ __forceinline__ void moments_group8(const Geo& g, const MomentsArgs& a, bool valid, uint32_t pix) {
    constexpr int RM = 3, NW = 2 * RM + 1;
    const int lane = threadIdx.x & 63, j = lane & 7, base = lane & ~7;
    const int x = (int)(pix % (uint32_t)g.W), y = g.y0 + (int)(pix / (uint32_t)g.W);
    valid = valid && y >= g.yb && y < g.ye;                           // the moments rows may be a sub-range of the temporal rows
    const size_t idx = valid ? (size_t)(y - g.y0) * g.W + x : 0;
    const int R = a.radius, xx = j - RM;
    const float h = (float)a.hist[idx];                               // :442
    const float4 cc = Store<ST>::ld4(a.colour, idx);                  // :450 raw load
    const float4 mc = a.motion[idx];
    const uint2 nraw = a.normal[idx];
    // the window column of this lane
    bool ok[NW];
    float tz[NW];
    float2 tm[NW];
    uint2 tn[NW];
    int th[NW];
    float4 ca[NW], cb[NW];
#pragma unroll
    for (int r = 0; r < NW; r++) {
        const int yy = r - RM, py = y + yy, px = x + xx;
        ok[r] = valid && j < NW && yy >= -R && yy <= R && py >= 0 && py < g.H && xx >= -R && xx <= R && px >= 0 && px < g.W;   // :473
        // A tap that does not exist reads the nearest texel the strip holds and is dropped.  (Not the centre: a select between the tap's and the
        // centre's address makes the compiler load the taps under a branch whose other arm is the centre's DATA — every tap load then waits
        // for the centre's: two dependent memory rounds per pass instead of one.)
        const int pyc = min(max(py, g.y0), g.y0 + g.rows - 1), pxc = min(max(px, 0), g.W - 1);
        const size_t p = (size_t)(pyc - g.y0) * g.W + pxc;
        tz[r] = ((const float*)a.motion)[p * 4 + 2];
        tm[r] = Store<ST>::ld2(a.mom, p);                                              // :480
        tn[r] = a.normal[p];                                                           // :483
        th[r] = (int)a.hist[p];                                                        // (unconditional: see moments_pixel)
        ca[r] = Store<ST>::ld4(a.colour, p);                                           // :479 raw ...
        cb[r] = Store<ST>::ld4(a.sparse_colour ? (const void*)a.out : a.colour, p);    // ... or, for an old non-sky texel, where the temporal launch put it
    }
    const float lc = lum_exact(cc.x, cc.y, cc.z);
    float zc, dzc;
    depth_of(mc, zc, dzc);
    const float3 nc = normal_of(nraw);
    const float il = hw_rcp(a.phi_colour);                            // :460
    const float phi_d = fmaxf(dzc, 1e-8f) * 3.0f;                     // :461
    // ARITH = 1: moments_lds_kernel's centre (svgf_moments_lds.h)
    const float ncz1 = unpack_h2(nraw.y).x, il1 = il * kLog2e;
    const float izb1 = hw_rcp(fmaxf(zc == kSkyZ ? 0.0f : mc.w, 1e-8f) * 3.0f) * kLog2e;
    // this lane's seven taps: weight and the values the sums take from them
    float tw[NW], twx[NW], t0[NW], t1[NW], t2[NW];
    unsigned okbits = 0u;
#pragma unroll
    for (int r = 0; r < NW; r++) {
        const int yy = r - RM;
        const bool in_out = a.sparse_colour && tz[r] != 0.0f && tz[r] != kSkyZ && th[r] >= 4;
        const float4 va = ca[r], vb = cb[r];
        t0[r] = in_out ? vb.x : va.x; t1[r] = in_out ? vb.y : va.y; t2[r] = in_out ? vb.z : va.z;
        const float zp = tz[r] == 0.0f ? kSkyZ : tz[r];               // depth_of, :482
        const float3 np = normal_of(tn[r]);
        const float len = sqrtf((float)(xx * xx + yy * yy));          // :488 (IEEE sqrt of a small integer: the value a constant would have)
        const float iz = (xx == 0 && yy == 0) ? 0.0f : hw_rcp(phi_d * len);   // phiDepth == 0 -> wZ = 0, :420
        if constexpr (ARITH == 0) {
            tw[r] = edge_weight(fabsf(lc - lum_exact(t0[r], t1[r], t2[r])), il, fabsf(zc - zp), iz, dot3_fma(nc, np), a.phi_normal);
        } else {
            const int l2 = xx * xx + yy * yy;                         // moments_taps49's depth scale by tap distance (len_class7)
            const float km = l2 == 1 ? 1.0f : l2 == 2 ? 0.70710678118654752f : l2 == 4 ? 0.5f : l2 == 5 ? 0.44721359549995794f : l2 == 8 ? 0.35355339059327376f
                           : l2 == 9 ? 0.33333333333333333f : l2 == 10 ? 0.31622776601683794f : l2 == 13 ? 0.27735009811261456f : 0.23570226039551584f;
            // both of moments_taps49's forms: the fused exponent (kTapsGeneral), and the reference's `max(term, 0.0)` = fmax, which drops a NaN
            // term (:424, kTapsNaN) — the sums take the first; a pixel whose sums come out NaN takes the second for ALL its taps, below
            const float d = clamp01(fmaf(unpack_h2(tn[r].y).x, ncz1, dot2_h2(tn[r].x, nraw.x)));
            const float en = hw_log2(d) * a.phi_normal;
            const float adl = fabsf(lum_exact(t0[r], t1[r], t2[r]) - lc);
            const float dz = fabsf(zp - zc), izk = l2 == 1 ? izb1 : izb1 * km;
            float e = fmaf(-adl, il1, en), ex = en - fmaxf(adl * il1, 0.0f);
            if (l2 != 0) { e = fmaf(-dz, izk, e); ex -= fmaxf(dz * izk, 0.0f); }
            tw[r] = hw_exp2(e);
            twx[r] = hw_exp2(ex);
        }
        okbits |= ok[r] ? 1u << r : 0u;
    }
    // the 49 taps in the reference's order; every lane of the group accumulates the same sums
    unsigned okc[NW];
#pragma unroll
    for (int c = 0; c < NW; c++) okc[c] = (unsigned)__shfl((int)okbits, base + c);
    float sw, sr, sg, sb, sm1, sm2;
    auto sums = [&](const float (&wt)[NW]) __attribute__((always_inline)) {
        sw = 0.f; sr = 0.f; sg = 0.f; sb = 0.f; sm1 = 0.f; sm2 = 0.f;
#pragma unroll
        for (int r = 0; r < NW; r++) {
#pragma unroll
            for (int c = 0; c < NW; c++) {
                // (all of a row's shuffles issued before its seven taps are accumulated shortens a pass, but costs 40 registers and a third of the
                // resident waves: slower for this launch, whose passes are spread one per wave - profiles/r04_small_experiments.txt block 9)
                const float w = __shfl(wt[r], base + c), c0 = __shfl(t0[r], base + c), c1 = __shfl(t1[r], base + c), c2 = __shfl(t2[r], base + c);
                const float m1 = __shfl(tm[r].x, base + c), m2 = __shfl(tm[r].y, base + c);
                if ((okc[c] >> r) & 1u) {
                    sw += w;                                          // :497-499
                    sr = fmaf(c0, w, sr); sg = fmaf(c1, w, sg); sb = fmaf(c2, w, sb);
                    sm1 = fmaf(m1, w, sm1); sm2 = fmaf(m2, w, sm2);
                }
            }
        }
    };
    sums(tw);
    if constexpr (ARITH == 1) {
        // moments_lds_kernel's rule (svgf_moments_lds.h): a pixel whose sums hold a NaN — one with a NaN, inf - inf or 0 x inf in its window, and
        // no other — is evaluated again the reference's way; every other pixel keeps the fused exponent's bits, whichever kernel serves it
        if (wave_any(__builtin_isunordered(sw, sm2) | __builtin_isunordered(sr, sg) | __builtin_isunordered(sb, sm1))) {
            const float w1 = sw, r1 = sr, g1 = sg, b1 = sb, a1 = sm1, a2 = sm2;
            const bool redo = __builtin_isunordered(w1, a2) | __builtin_isunordered(r1, g1) | __builtin_isunordered(b1, a1);
            sums(twx);
            if (!redo) { sw = w1; sr = r1; sg = g1; sb = b1; sm1 = a1; sm2 = a2; }
        }
    }
    if (!valid || j != 0) return;                                     // one lane of the group writes
    if (a.cold_only && !(h < 4.0f)) return;                           // already written by temporal_kernel (passthrough_out)
    if (!(h < 4.0f)) { Store<ST>::st4(a.out, idx, cc); return; }      // :521
    // (a zero-normal centre needs no special case: all its weights came out as exactly 0 above, see moments_pixel)
    sw = fmaxf(sw, 1e-6f);                                            // :505
    const float inv = 1.0f / sw;
    sm1 *= inv; sm2 *= inv;
    const float var = (sm2 - sm1 * sm1) * (4.0f / h);                 // :511-514
    Store<ST>::st4(a.out, idx, make_float4(sr * inv, sg * inv, sb * inv, var));   // :516 unclamped
}

template <int ST>
__global__ __launch_bounds__(kBX* kBY) void moments_kernel(Geo g, MomentsArgs a) {
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    moments_pixel<ST>(g, a, x, y);
}

// The 3x3 variant of the estimate (moments_radius = 1) with the neighbourhood shared through wave64 shuffles: a wave is
// 64 consecutive pixels of a row; every lane loads its own column of rows y-1, y, y+1 once (coalesced) and takes the
// columns x-1 / x+1 from its neighbour lanes (__shfl_up / __shfl_down); only lanes 0 and 63 fetch the column beyond the
// wave.  3 row loads per plane instead of 9 gathers.  Same expressions in the same order as moments_pixel: bit-identical.
struct MomTap { float cx, cy, cz, m1, m2, z; uint32_t n01, n2; };
__device__ __forceinline__ MomTap shfl_tap(const MomTap& t, int dir) {
    MomTap r;
#define SVGF_SH(f) r.f = dir < 0 ? __shfl_up(t.f, 1) : __shfl_down(t.f, 1)
    SVGF_SH(cx); SVGF_SH(cy); SVGF_SH(cz); SVGF_SH(m1); SVGF_SH(m2); SVGF_SH(z); SVGF_SH(n01); SVGF_SH(n2);
#undef SVGF_SH
    return r;
}
template <int ST>
__device__ __forceinline__ MomTap load_tap(const Geo& g, const MomentsArgs& a, int px, int py) {
    const size_t p = (size_t)(py - g.y0) * g.W + px;
    const float4 c = Store<ST>::ld4(a.colour, p);                     // :479 raw
    const float2 m = Store<ST>::ld2(a.mom, p);                        // :480
    float z, dz;
    depth_of(a.motion[p], z, dz);                                     // :482
    const uint2 n = a.normal[p];                                      // :483
    return MomTap{c.x, c.y, c.z, m.x, m.y, z, n.x, n.y};
}
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void moments3x3_shfl_kernel(Geo g, MomentsArgs a) {
    const int lane = threadIdx.x;
    const int x = blockIdx.x * kBX + lane;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (y >= g.ye) return;                                            // the whole wave
    const int xl = min(x, g.W - 1);                                   // lanes beyond the frame load a valid texel nobody uses
    MomTap own[3], lft[3], rgt[3];
    bool rowok[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int py = y + r - 1;
        rowok[r] = py >= 0 && py < g.H;                               // :473 (wave-uniform)
        if (!rowok[r]) continue;
        own[r] = load_tap<ST>(g, a, xl, py);
        lft[r] = shfl_tap(own[r], -1);
        rgt[r] = shfl_tap(own[r], +1);
        if (lane == 0 || lane == kBX - 1) {                           // the columns just outside the wave: one masked load
            const int xe = lane == 0 ? x - 1 : x + 1;
            if (xe >= 0 && xe < g.W) {
                const MomTap e = load_tap<ST>(g, a, xe, py);
                if (lane == 0) lft[r] = e; else rgt[r] = e;
            }
        }
    }
    if (x >= g.W) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    const float h = (float)a.hist[idx];                               // :442
    if (a.cold_only && !(h < 4.0f)) return;
    if (!(h < 4.0f)) { Store<ST>::st4(a.out, idx, Store<ST>::ld4(a.colour, idx)); return; }   // :521
    const MomTap& cc = own[1];
    const float lc = lum_exact(cc.cx, cc.cy, cc.cz);
    float zc, dzc;
    depth_of(a.motion[idx], zc, dzc);
    const float3 nc = normal_of(make_uint2(cc.n01, cc.n2));
    const float il = hw_rcp(a.phi_colour);                            // :460
    const float phi_d = fmaxf(dzc, 1e-8f) * 3.0f;                     // :461
    float sw = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sm1 = 0.f, sm2 = 0.f;
#pragma unroll
    for (int yy = -1; yy <= 1; yy++) {
        if (!rowok[yy + 1]) continue;
#pragma unroll
        for (int xx = -1; xx <= 1; xx++) {
            const int px = x + xx;
            if (px < 0 || px >= g.W) continue;
            const MomTap& t = xx < 0 ? lft[yy + 1] : (xx == 0 ? own[yy + 1] : rgt[yy + 1]);
            const float3 np = normal_of(make_uint2(t.n01, t.n2));
            const float len = sqrtf((float)(xx * xx + yy * yy));      // :488
            const float iz = (xx == 0 && yy == 0) ? 0.0f : hw_rcp(phi_d * len);
            const float w = edge_weight(fabsf(lc - lum_exact(t.cx, t.cy, t.cz)), il, fabsf(zc - t.z), iz, dot3_fma(nc, np), a.phi_normal);
            sw += w;                                                  // :497-499
            sr = fmaf(t.cx, w, sr); sg = fmaf(t.cy, w, sg); sb = fmaf(t.cz, w, sb);
            sm1 = fmaf(t.m1, w, sm1); sm2 = fmaf(t.m2, w, sm2);
        }
    }
    sw = fmaxf(sw, 1e-6f);                                            // :505
    const float inv = 1.0f / sw;
    sm1 *= inv; sm2 *= inv;
    const float var = (sm2 - sm1 * sm1) * (4.0f / h);                 // :511-514
    Store<ST>::st4(a.out, idx, make_float4(sr * inv, sg * inv, sb * inv, var));   // :516 unclamped
}

// Steady state inside the frame driver: the temporal launch already copied every pixel with history >= 4 and told this launch where
// the young ones are: one 64-bit lane mask per (row, 64-column segment), and the LIST of the pixels of the partly young segments.
// All are served by moments_group8, eight pixels per wave-pass — a launch of a few thousand passes (their gathering loads: the L1s' lookup
// rate is its bound) that wants them spread over all resident waves, one each: 168 registers, three waves per SIMD (tests/test_kernel_budgets.py).
//  * List workgroups: eight consecutive entries per pass, group g to wave g mod (waves).
//  * Scan workgroups: 256 masks each (segment s belongs to scan slot s mod (scan_blocks / 2): a row of all-young segments — the rows that
//    have just entered the frame under a vertical pan — spreads over as many slots); the segments whose mask is FULL are shared through LDS
//    as four ballots, and the eight waves of the slot's TWO workgroups take one eighth of each.  Scan workgroups come first in the grid:
//    theirs are the longer chains.
//  * A frame whose list is over its cap (svgf_kernels.h; thin geometry under motion, or half the frame disoccluded — for the frame or two
//    until the frame driver's sample sends such frames to the streaming kernel): the list is ignored, the slot's workgroups turn their 256
//    masks into items {segment, eight of its young pixels} in LDS and their waves take the items in turn.
// (Bench pan, ~28 000 listed pixels + ~300 all-young segments per 4K frame: 0.051 ms in round 2, 0.034 in round 3 (eight lanes per pixel,
// two workgroups per slot), 0.030 now (the window loads of a pass in ONE memory round, not eight); nothing young: 0.006 ms by the trace.)
constexpr int kScanSplit = 2;

// position of the r-th (0-based) set bit of m; r < popcount(m)
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int r) {
    unsigned w = (unsigned)m;
    int pos = 0, c = __popc(w);
    if (r >= c) { r -= c; w = (unsigned)(m >> 32); pos = 32; }
    c = __popc(w & 0xffffu); if (r >= c) { r -= c; w >>= 16; pos += 16; }
    c = __popc(w & 0xffu);   if (r >= c) { r -= c; w >>= 8;  pos += 8; }
    c = __popc(w & 0xfu);    if (r >= c) { r -= c; w >>= 4;  pos += 4; }
    c = __popc(w & 0x3u);    if (r >= c) { r -= c; w >>= 2;  pos += 2; }
    return pos + (r >= (int)(w & 1u) ? 1 : 0);
}

template <int ST, int ARITH>
__global__ __launch_bounds__(256, 3) void moments_young_kernel(Geo g, MomentsArgs a, int scan_blocks) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t = threadIdx.x;
    __shared__ unsigned long long full[4];
    __shared__ unsigned long long smask[256];       // over the cap only: the slot's masks ...
    __shared__ uint16_t sitem[256 * 8];             // ... and its work items {thread that read the segment's mask, which eight of its young pixels}
    __shared__ int swave[4];
    const int nseg = (g.W + kBX - 1) / kBX;
    const int first = (g.yb - g.y0) * nseg, last = (g.ye - g.y0) * nseg;             // mask range of the launch rows
    if ((int)blockIdx.x >= scan_blocks) {
        if (a.young_count[kYoungFlagOffset] == 0ull) {          // no shard over its cap: the lists are complete
            const unsigned list_blocks = gridDim.x - (unsigned)scan_blocks, b = blockIdx.x - (unsigned)scan_blocks;
            // groups of eight entries, shard after shard: [first[s], first[s + 1])
            unsigned n[kYoungShards], first_grp[kYoungShards + 1];
            first_grp[0] = 0u;
#pragma unroll
            for (int sh = 0; sh < kYoungShards; sh++) { n[sh] = (unsigned)a.young_count[sh * kYoungLine]; first_grp[sh + 1] = first_grp[sh] + (n[sh] + 7u) / 8u; }
            const unsigned region = a.young_cap / kYoungShards * 63u;
            for (unsigned grp = b + list_blocks * (unsigned)w; grp < first_grp[kYoungShards]; grp += list_blocks * 4u) {     // (uniform over the wave)
                unsigned sh = 0u;
#pragma unroll
                for (int q = 1; q < kYoungShards; q++) sh += grp >= first_grp[q] ? 1u : 0u;
                unsigned f0 = 0u, ns = 0u;
#pragma unroll
                for (int q = 0; q < kYoungShards; q++) { if (sh == (unsigned)q) { f0 = first_grp[q]; ns = n[q]; } }
                const unsigned i = (grp - f0) * 8u + ((unsigned)lane >> 3);
                const bool valid = i < ns;
                moments_group8<ST, ARITH>(g, a, valid, valid ? a.young_list[sh * region + i] : 0u);
            }
        }
    } else {
        constexpr int F = kScanSplit;               // workgroups that share the masks of one scan slot
        const int part = (int)blockIdx.x % F, bid = (int)blockIdx.x / F, nslots = scan_blocks / F;
        for (int base = first; base + bid < last; base += nslots * 256) {    // (uniform over the workgroup)
            const int sidx = base + t * nslots + bid;
            unsigned long long m = sidx < last ? a.young_masks[sidx] : 0ull;          // (requested together with the counter: one memory round)
            const bool overflow = a.young_count[kYoungFlagOffset] != 0ull;           // (uniform over the launch)
            if (!overflow) {
                // the all-young segments of the slot; each of the slot's 4 F waves takes 8 / F octants of every one
                const unsigned long long fm = __ballot(m == ~0ull);
                if (lane == 0) full[w] = fm;
                __syncthreads();
#pragma unroll 1
                for (int ww = 0; ww < 4; ww++) {
                    unsigned long long mm = full[ww];
                    while (mm) {
                        const int bit = __builtin_ctzll(mm);
                        mm &= mm - 1;
                        const int seg = base + (ww * 64 + bit) * nslots + bid, yl = seg / nseg;
#pragma unroll 1
                        for (int o = part * (8 / F) + w; o < (part + 1) * (8 / F); o += 4) {
                            const int x = (seg % nseg) * kBX + o * 8 + (lane >> 3);
                            moments_group8<ST, ARITH>(g, a, x < g.W, (uint32_t)(yl * g.W + (x < g.W ? x : 0)));
                        }
                    }
                }
            } else {
                // the list is void: every young pixel of the slot's masks, as items {segment, eight of its young pixels} in LDS (built by each of
                // the slot's workgroups for itself), dealt out over the slot's 4 F waves.  A frame or two until the frame driver's sample says
                // "crowded" and the streaming kernel takes over (svgf_set_adaptive_moments): eight pixels per pass only where a segment holds them.
                const int np = (__builtin_popcountll(m) + 7) >> 3;            // passes this segment needs: 0 .. 8
                int inc = np;                                                 // inclusive scan over the wave
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
                if (lane == 63) swave[w] = inc;
                smask[t] = m;
                __syncthreads();
                int off = inc - np, total = 0;
#pragma unroll
                for (int ww = 0; ww < 4; ww++) { const int n = swave[ww]; if (ww < w) off += n; total += n; }
                for (int q = 0; q < np; q++) sitem[off + q] = (uint16_t)((t << 3) | q);
                __syncthreads();
#pragma unroll 1
                for (int it = part * 4 + w; it < total; it += 4 * F) {        // (uniform over the wave)
                    const int e = sitem[it], tt = e >> 3;
                    const unsigned long long mm = smask[tt];
                    const int r = (e & 7) * 8 + (lane >> 3);
                    const bool valid = r < __builtin_popcountll(mm);
                    const int seg = base + tt * nslots + bid;
                    moments_group8<ST, ARITH>(g, a, valid, (uint32_t)((seg / nseg) * g.W + (seg % nseg) * kBX + nth_set_bit(mm, valid ? r : 0)));
                }
            }
            __syncthreads();
        }
    }
    // The temporal launch wrote exact zeros for young pixels with an all-zero normal (its zero-normal shortcut) — right unless a
    // texel of the pixel's 7x7 window is NaN or inf (0 x NaN, :498-499).  It listed the pixels whose result is not finite: every
    // shortcut pixel in the window of a listed pixel gets the full estimate here (moments_group8 has no shortcut).  Nothing listed —
    // every frame without a NaN — and this is one word read per wave.
    const unsigned nn = *a.nan_count;
    if (nn == 0u) return;
    const unsigned wave = blockIdx.x * 4u + (unsigned)w, nwaves = gridDim.x * 4u;
    auto shortcut_px = [&](bool valid, uint32_t q) {             // young, with an all-zero normal
        const uint2 nq = a.normal[q];
        return valid && a.hist[q] < 4 && ((nq.x & 0x7fff7fffu) | (nq.y & 0x7fffu)) == 0u;
    };
    if (nn <= kNanListCap) {
        for (unsigned i = wave; i < nn; i += nwaves) {
            const uint32_t p = a.nan_list[i];
            const int px = (int)(p % (uint32_t)g.W), pyl = (int)(p / (uint32_t)g.W);
#pragma unroll 1
            for (int pass = 0; pass < 7; pass++) {
                const int k = pass * 8 + (lane >> 3), qx = px + k % 7 - 3, qyl = pyl + k / 7 - 3;
                const bool valid = k < 49 && qx >= 0 && qx < g.W && qyl >= 0 && qyl < g.rows;
                const uint32_t q = valid ? (uint32_t)(qyl * g.W + qx) : 0u;
                const bool need = shortcut_px(valid, q);         // (a NaN pixel on a surface far from the sky costs two small loads per pass)
                if (!wave_any(need)) continue;
                moments_group8<ST, ARITH>(g, a, need, q);
            }
        }
    } else {                                    // the list overflowed (a frame full of NaN): every shortcut pixel of the launch rows
        const unsigned firstp = (unsigned)(g.yb - g.y0) * (unsigned)g.W, lastp = (unsigned)(g.ye - g.y0) * (unsigned)g.W;
        for (unsigned q0 = firstp + wave * 8u; q0 < lastp; q0 += nwaves * 8u) {
            const unsigned q = q0 + ((unsigned)lane >> 3), qs = q < lastp ? q : firstp;
            const bool need = shortcut_px(q < lastp, qs);
            if (!wave_any(need)) continue;
            moments_group8<ST, ARITH>(g, a, need, qs);
        }
    }
}

// ------------------------------------------------------------------ a-trous (direct) ----------
// Filter.cuh:527-624, one thread per pixel, taps gathered straight from global memory (L1/L2).
// Kept as the simple variant (SVGF_VARIANT_DIRECT) the LDS-tiled kernel is A/B-tested against.
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void atrous_direct_kernel(Geo g, AtrousArgs a) {
    keep_nan_in_clamps();                                             // imageLoad keeps a NaN (svgf_device.h)
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    const float4 c = clamp01_ref(Store<ST>::ld4(a.in, idx));          // :543 (a NaN stays NaN)
    float zc, dzc;
    depth_of(a.motion[idx], zc, dzc);                                 // :552
    if (zc == kSkyZ) { Store<ST>::st4(a.out, idx, c); return; }       // :554-558
    const float3 nc = normal_of(a.normal[idx]);
    const float lc = lum_exact(c.x, c.y, c.z);
    const float il = inv_phi_l_log2e(c.w, hw_rcp(a.phi_colour) * kLog2e) * 0.6931471805599453f;   // :562 (the LDS kernels' expression, back in natural units)
    const float phi_d = fmaxf(dzc, 1e-6f) * (float)a.step;            // :563
    float sw = 1.0f, sr = c.x, sg = c.y, sb = c.z, sv = c.w;          // :567-568
    const float K[3] = {1.0f, (float)(2.0 / 3.0), (float)(1.0 / 6.0)};           // :540
#pragma unroll
    for (int yy = -2; yy <= 2; yy++) {
        const int py = y + yy * a.step;
        if (py < 0 || py >= g.H) continue;                            // :579
#pragma unroll
        for (int xx = -2; xx <= 2; xx++) {
            if (xx == 0 && yy == 0) continue;                         // :584
            const int px = x + xx * a.step;
            if (px < 0 || px >= g.W) continue;
            const size_t p = (size_t)(py - g.y0) * g.W + px;
            const float4 q = clamp01_ref(Store<ST>::ld4(a.in, p));    // :586
            float zp, dzp;
            depth_of(a.motion[p], zp, dzp);
            const float3 np = normal_of(a.normal[p]);
            const float len = sqrtf((float)(xx * xx + yy * yy));      // compile-time after unrolling
            const float w = edge_weight(fabsf(lc - lum_exact(q.x, q.y, q.z)), il, fabsf(zc - zp), hw_rcp(phi_d * len),
                                        dot3_fma(nc, np), a.phi_normal);
            const float gk = w * (K[xx < 0 ? -xx : xx] * K[yy < 0 ? -yy : yy]);   // :582,604
            sw += gk;                                                 // :607-608
            sr = fmaf(gk, q.x, sr); sg = fmaf(gk, q.y, sg); sb = fmaf(gk, q.z, sb);
            sv = fmaf(gk * gk, q.w, sv);
        }
    }
    const float inv = 1.0f / sw;
    const float4 o = make_float4(sr * inv, sg * inv, sb * inv, sv * (inv * inv));   // :615
    Store<ST>::st4(a.out, idx, o);                                    // :618 unclamped
    if (a.feedback) Store<ST>::st4(a.feedback, idx, o);               // :619-622
}



// ------------------------------------------------------------------ TAA + sRGB -----------------
// filter::TAAFilterKernel (Filter.cuh:288-357): the stage application::Render runs right after the wavelet
// filter (App.cu:558).  Neighbourhood clamp in gamma-2 PAL-YUV of the previous output against the 3x3
// neighbourhood of the filtered frame, then linear -> sRGB.  Quirks kept: textureSample returns
// the nearest texel (:101-102,130-131) at floor(uv*(W-1)), i.e. one pixel up-left of the fragment; the stored alpha
// is always 1 and the updated mixRate is never used.  The previous output is read from a separate plane: the
// reference reads it from the buffer it is writing (App.cu:520), at a different pixel — a race.
__device__ __forceinline__ int tex_coord(float uv, int n) {
    const int x0 = (int)floorf(uv * (float)(n - 1));
    return min(max(x0, 0), n - 1);
}
__device__ __forceinline__ float3 enc_yuv(float3 c) {                   // :267-275; pow(x,2) = x*x correctly rounded
    const float r = c.x * c.x, g = c.y * c.y, b = c.z * c.z;
    return make_float3((r * 0.299f + g * 0.587f) + b * 0.114f, (r * -0.14713f + g * -0.28886f) + b * 0.436f,
                       (r * 0.615f + g * -0.51499f) + b * -0.10001f);
}
__device__ __forceinline__ float to_srgb(float c) {                     // :145-148
    return (c <= 0.0031308f) ? 12.92f * c : 1.055f * hw_exp2(hw_log2(c) * (1.0f / 2.4f)) - 0.055f;
}

// :330-338: the neighbourhood clamp of the blended colour `ya` in YUV: y[0] the centre, y[1..4] the plus-shaped and y[5..8] the diagonal
// neighbours.  EXACT: glm's min / max, `(y < x) ? y : x` and `(x < y) ? y : x`, in the reference's association — what a NaN does there
// depends on its position (min(a, NaN) = a, min(NaN, b) = NaN); taken by a wave that holds a NaN (the reference's clamps keep a NaN texel,
// :78-83, and :351 then turns the pixel black — but its neighbours' min / max have seen it).  Otherwise fminf / fmaxf: the same values.
template <bool EXACT>
__device__ __forceinline__ float3 taa_clamp(float3 ya, const float3 (&y)[9]) {
    auto mn2 = [](float a, float b) { return EXACT ? ((b < a) ? b : a) : fminf(a, b); };
    auto mx2 = [](float a, float b) { return EXACT ? ((a < b) ? b : a) : fmaxf(a, b); };
    float r[3];
    const float yav[3] = {ya.x, ya.y, ya.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        auto ch = [&](int i) { return k == 0 ? y[i].x : (k == 1 ? y[i].y : y[i].z); };
        float mn, mx, mnd, mxd;
        if constexpr (EXACT) {
            mn = mn2(mn2(mn2(ch(0), ch(1)), mn2(ch(2), ch(3))), ch(4));
            mx = mx2(mx2(mx2(ch(0), ch(1)), mx2(ch(2), ch(3))), ch(4));
            mnd = mn2(mn2(mn2(ch(5), ch(6)), mn2(ch(7), ch(8))), mn);
            mxd = mx2(mx2(mx2(ch(5), ch(6)), mx2(ch(7), ch(8))), mx);
        } else {
            mn = mx = ch(0);
#pragma unroll
            for (int i = 1; i <= 4; i++) { mn = fminf(mn, ch(i)); mx = fmaxf(mx, ch(i)); }
            mnd = mn; mxd = mx;
#pragma unroll
            for (int i = 5; i <= 8; i++) { mnd = fminf(mnd, ch(i)); mxd = fmaxf(mxd, ch(i)); }
        }
        mn = mix_exact(mn, mnd, 0.5f);                                        // :332-335
        mx = mix_exact(mx, mxd, 0.5f);
        r[k] = mn2(mx2(yav[k], mn), mx);                                      // :338 clamp = min(max(x, lo), hi)
    }
    return make_float3(r[0], r[1], r[2]);
}
__device__ __forceinline__ bool any_nan3(float3 v) { return __builtin_isunordered(v.x, v.y) | (v.z != v.z); }

template <int ST>
__global__ __launch_bounds__(kBX* kBY) void taa_kernel(Geo g, const void* filtered, const void* history, void* out) {
    keep_nan_in_clamps();                                                 // imageLoad keeps a NaN (svgf_device.h)
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const float iw = 1.0f / (float)g.W, ih = 1.0f / (float)g.H;
    const float u = (float)x * iw, v = (float)y * ih;                    // :296
    const int sx[3] = {tex_coord(u - iw, g.W), tex_coord(u, g.W), tex_coord(u + iw, g.W)};
    const int sy[3] = {tex_coord(v - ih, g.H), tex_coord(v, g.H), tex_coord(v + ih, g.H)};
    auto at = [&](const void* img, int ix, int iy) { return clamp01_ref(Store<ST>::ld4(img, (size_t)(sy[iy] - g.y0) * g.W + sx[ix])); };
    const float4 last = at(history, 1, 1);                                // :299
    const float mix = fminf(last.w, 0.5f);                                // :302 (CUDA's min(float, double) is fmin: a NaN alpha gives 0.5)
    const float4 c0 = at(filtered, 1, 1);                                 // :305
    float3 aa = make_float3(sqrtf(mix_exact(last.x * last.x, c0.x * c0.x, mix)), sqrtf(mix_exact(last.y * last.y, c0.y * c0.y, mix)),
                            sqrtf(mix_exact(last.z * last.z, c0.z * c0.z, mix)));   // :307-308
    float3 ya = enc_yuv(aa);                                              // :319
    // :310-317,320-328: the centre, the plus-shaped and the diagonal neighbours
    float3 nbv[9];
    const int nx[9] = {1, 2, 0, 1, 1, 2, 0, 2, 0}, ny[9] = {1, 1, 1, 2, 0, 2, 2, 0, 0};
    bool nan_in = any_nan3(ya);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float4 c = k == 0 ? c0 : at(filtered, nx[k], ny[k]);
        nbv[k] = enc_yuv(make_float3(c.x, c.y, c.z));
        nan_in = nan_in | any_nan3(nbv[k]);
    }
    ya = wave_any(nan_in) ? taa_clamp<true>(ya, nbv) : taa_clamp<false>(ya, nbv);
    // :277-285; pow(x, 0.5) = sqrt(x), NaN for negative x
    float r = sqrtf((ya.x * 1.0f + ya.y * 0.0f) + ya.z * 1.13983f);
    float gg = sqrtf((ya.x * 1.0f + ya.y * -0.39465f) + ya.z * -0.58060f);
    float b = sqrtf((ya.x * 1.0f + ya.y * 2.03211f) + ya.z * 0.0f);
    if (r != r || gg != gg || b != b) { r = 0.f; gg = 0.f; b = 0.f; }     // :351
    const float4 o = make_float4(to_srgb(r), to_srgb(gg), to_srgb(b), 1.0f);   // :353
    Store<ST>::st4(out, (size_t)(y - g.y0) * g.W + x, clamp01_ref(o));   // :355 imageStore
}

// The same stage with the neighbourhood's YUV values computed ONCE per texel: a workgroup covers 64 x 8 pixels, encodes
// the 68 x 12 filtered texels its samples can touch into LDS (the nearest-texel coordinates floor(uv*(N-1)) land one to
// three texels up-left of the pixel, depending on fp32 rounding), and every pixel takes its nine neighbours from there
// instead of nine gathers + nine YUV encodings.  Same functions on the same inputs: bit-identical to taa_kernel.
constexpr int kTaaW = 68, kTaaH = 12, kTaaRows = 8;
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void taa_lds_kernel(Geo g, const void* filtered, const void* history, void* out) {
    keep_nan_in_clamps();                                                 // imageLoad keeps a NaN (svgf_device.h)
    __shared__ float4 yuv[kTaaH][kTaaW];                                  // 16-B records: one ds_read_b128 per neighbour
    __shared__ uint32_t wave_nan[kBY];                                    // per wave: a texel it staged holds a NaN in YUV (-> the workgroup clamps with glm's min / max)
    const int x0 = blockIdx.x * kBX, yb = g.yb + blockIdx.y * kTaaRows;
    bool staged_nan = false;
    auto stage = [&](int lx, int ly) {
        const int gx = x0 - 3 + lx, gy = yb - 3 + ly;
        float3 e = make_float3(0.f, 0.f, 0.f);
        if (gx >= 0 && gx < g.W && gy >= 0 && gy < g.H && gy >= g.y0 && gy < g.y0 + g.rows) {
            const float4 c = clamp01_ref(Store<ST>::ld4(filtered, (size_t)(gy - g.y0) * g.W + gx));
            e = enc_yuv(make_float3(c.x, c.y, c.z));
        }
        staged_nan = staged_nan | any_nan3(e);
        yuv[ly][lx] = make_float4(e.x, e.y, e.z, 0.f);
    };
#pragma unroll
    for (int k = 0; k < kTaaH / kBY; k++) stage(threadIdx.x, threadIdx.y + kBY * k);     // columns 0..63 of the 12 rows
    {
        const int id = threadIdx.y * kBX + threadIdx.x;                                   // columns 64..67
        if (id < (kTaaW - kBX) * kTaaH) stage(kBX + (id & 3), id >> 2);
    }
    { const bool w = wave_any(staged_nan); if (threadIdx.x == 0) wave_nan[threadIdx.y] = w ? 1u : 0u; }
    __syncthreads();
    const bool tile_exact = (wave_nan[0] | wave_nan[1] | wave_nan[2] | wave_nan[3]) != 0u;
    static_assert(kBY == 4, "one flag word per wave");
    const int x = x0 + threadIdx.x;
    if (x >= g.W) return;
    const float iw = 1.0f / (float)g.W, ih = 1.0f / (float)g.H;
    const float u = (float)x * iw;                                        // :296
    const int sx[3] = {tex_coord(u - iw, g.W), tex_coord(u, g.W), tex_coord(u + iw, g.W)};
#pragma unroll
    for (int r = 0; r < kTaaRows / kBY; r++) {
        const int y = yb + threadIdx.y + kBY * r;
        if (y >= g.ye) continue;
        const float v = (float)y * ih;
        const int sy[3] = {tex_coord(v - ih, g.H), tex_coord(v, g.H), tex_coord(v + ih, g.H)};
        auto nb = [&](int ix, int iy) { const float4 e = yuv[sy[iy] - (yb - 3)][sx[ix] - (x0 - 3)]; return make_float3(e.x, e.y, e.z); };
        const size_t ci = (size_t)(sy[1] - g.y0) * g.W + sx[1];
        const float4 last = clamp01_ref(Store<ST>::ld4(history, ci));     // :299
        const float mix = fminf(last.w, 0.5f);                            // :302 (CUDA's min(float, double) is fmin)
        const float4 c0 = clamp01_ref(Store<ST>::ld4(filtered, ci));      // :305
        float3 aa = make_float3(sqrtf(mix_exact(last.x * last.x, c0.x * c0.x, mix)), sqrtf(mix_exact(last.y * last.y, c0.y * c0.y, mix)),
                                sqrtf(mix_exact(last.z * last.z, c0.z * c0.z, mix)));   // :307-308
        float3 ya = enc_yuv(aa);                                          // :319
        const int nx[9] = {1, 2, 0, 1, 1, 2, 0, 2, 0}, ny[9] = {1, 1, 1, 2, 0, 2, 2, 0, 0};
        float3 nbv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) nbv[k] = nb(nx[k], ny[k]);
        // (taa_kernel takes the exact form per wave; a tile that holds a NaN takes it for all its waves: for NaN-free values the two forms agree)
        ya = (tile_exact || wave_any(any_nan3(ya))) ? taa_clamp<true>(ya, nbv) : taa_clamp<false>(ya, nbv);
        float rr = sqrtf((ya.x * 1.0f + ya.y * 0.0f) + ya.z * 1.13983f);
        float gg = sqrtf((ya.x * 1.0f + ya.y * -0.39465f) + ya.z * -0.58060f);
        float bb = sqrtf((ya.x * 1.0f + ya.y * 2.03211f) + ya.z * 0.0f);
        if (rr != rr || gg != gg || bb != bb) { rr = 0.f; gg = 0.f; bb = 0.f; }     // :351
        const float4 o = make_float4(to_srgb(rr), to_srgb(gg), to_srgb(bb), 1.0f);  // :353
        Store<ST>::st4(out, (size_t)(y - g.y0) * g.W + x, clamp01_ref(o));    // :355 imageStore
    }
}

// ------------------------------------------------------------------ G-buffer adapter -----------
// What resources/shaders/GBuffer.frag:62-88 (+ GBuffer.vert:21-34) writes, from linear attribute planes.  All
// arithmetic is unfused fp32 in a fixed order so that the CPU restatement reproduces it bit for bit.
__device__ __forceinline__ float4 mat_mul_point(const float* m, float3 p) {       // column-major m * (p,1)
    return make_float4(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12], ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
                       ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14], ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15]);
}
__device__ __forceinline__ float depth_at(const PackArgs& a, size_t idx, bool& covered) {
    const float4 n = a.normal[idx];
    covered = !(n.x == 0.0f && n.y == 0.0f && n.z == 0.0f);
    const float4 p = a.position[idx];
    const float dx = a.cam[0] - p.x, dy = a.cam[1] - p.y, dz = a.cam[2] - p.z;
    return sqrtf((dx * dx + dy * dy) + dz * dz);                                   // distance(), GBuffer.frag:70
}
__global__ __launch_bounds__(kBX* kBY) void pack_gbuffer_kernel(Geo g, PackArgs a) {
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    bool covered;
    const float depth = depth_at(a, idx, covered);
    if (!covered) {                                                               // cleared texel (App.cu:383-384)
        a.motion[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.normal_out[idx] = make_uint2(0u, 0u);
        a.uv_out[idx] = make_uint2(0u, 0u);
        return;
    }
    const float4 p = a.position[idx];
    const float4 cur = mat_mul_point(a.vp, make_float3(p.x, p.y, p.z)), prev = mat_mul_point(a.pvp, make_float3(p.x, p.y, p.z));
    const float mvx = (prev.x / prev.w - cur.x / cur.w) * (0.5f * (float)g.W);    // GBuffer.frag:65-67
    const float mvy = (prev.y / prev.w - cur.y / cur.w) * (0.5f * (float)g.H);
    // dFdx / dFdy: differences inside the 2x2 quad (GBuffer.frag:71)
    const int xp = x ^ 1, yp = y ^ 1;
    float ddx = 0.0f, ddy = 0.0f;
    bool c2;
    if (xp < g.W) { const float d2 = depth_at(a, (size_t)(y - g.y0) * g.W + xp, c2); if (c2) ddx = fabsf(d2 - depth); }
    if (yp < g.H && yp - g.y0 >= 0 && yp - g.y0 < g.rows) { const float d2 = depth_at(a, (size_t)(yp - g.y0) * g.W + x, c2); if (c2) ddy = fabsf(d2 - depth); }
    a.motion[idx] = make_float4(mvx, mvy, depth, fmaxf(ddx, ddy));
    const float4 n = a.normal[idx];
    const float len = sqrtf((n.x * n.x + n.y * n.y) + n.z * n.z);                 // normalize(), GBuffer.frag:62
    a.normal_out[idx] = make_uint2(pack_h2(n.x / len, n.y / len), pack_h2(n.z / len, n.w));   // Vec4ToUVec4 (packHalf2x16), :48-60,87
    const float4 b = a.bary[idx];
    a.uv_out[idx] = make_uint2(pack_h2(b.x, b.y), pack_h2(b.z, b.w));
}


inline dim3 grid_for(const Geo& g) { return dim3((g.W + kBX - 1) / kBX, (g.ye - g.yb + kBY - 1) / kBY); }

}  // namespace

hipError_t launch_temporal(const Geo& g, int storage, const TemporalArgs& a, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    const int ylo = a.guide_out ? std::min(g.yb, a.guide_lo) : g.yb, yhi = a.guide_out ? std::max(g.ye, a.guide_hi) : g.ye;
    const dim3 block(kBX, kBY), grid((g.W + kBX - 1) / kBX, (yhi - ylo + kBY - 1) / kBY);
    if (storage == 0) temporal_kernel<0><<<grid, block, 0, s>>>(g, a);
    else temporal_kernel<1><<<grid, block, 0, s>>>(g, a);
    return hipGetLastError();
}

hipError_t launch_moments(const Geo& g, int storage, const MomentsArgs& a, bool direct, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    // dense: the caller knows (nearly) every pixel is young — the LDS-streaming kernel; it needs the reference's
    // radius and a non-degenerate PhiNormal (the fused exponent would see 0 * -inf)
    if (a.dense && a.radius == kMR && a.phi_normal != 0.0f)
        return storage == 0 ? launch_moments_lds<0>(g, a, s) : launch_moments_lds<1>(g, a, s);
    // the 3x3 variant on a plane full of young pixels (stage call, or the first frames of a sequence): wave64 shuffles
    if (a.radius == 1 && !direct && (a.dense || !a.cold_only)) {
        const dim3 block(kBX, kBY), grid = grid_for(g);
        if (storage == 0) moments3x3_shfl_kernel<0><<<grid, block, 0, s>>>(g, a);
        else moments3x3_shfl_kernel<1><<<grid, block, 0, s>>>(g, a);
        return hipGetLastError();
    }
    if (a.cold_only && a.young_masks) {
        const int nsegs = (g.ye - g.yb) * ((g.W + kBX - 1) / kBX);
        int scan = (nsegs + 255) / 256;                            // >= one mask per lane and load ...
        if (scan > 4 * num_cus()) scan = 4 * num_cus();            // ... on at most one resident round
        const int walk = std::min(4 * num_cus(), std::max(1, nsegs / 16));   // the list holds at most 63 pixels per segment
        scan *= kScanSplit;
        // the arithmetic of the kernel the stage calls / the dense frames of this configuration run (moments_group8)
        const bool lds_arith = !direct && a.radius == kMR && a.phi_normal != 0.0f;
        if (storage == 0) { if (lds_arith) moments_young_kernel<0, 1><<<scan + walk, 256, 0, s>>>(g, a, scan); else moments_young_kernel<0, 0><<<scan + walk, 256, 0, s>>>(g, a, scan); }
        else { if (lds_arith) moments_young_kernel<1, 1><<<scan + walk, 256, 0, s>>>(g, a, scan); else moments_young_kernel<1, 0><<<scan + walk, 256, 0, s>>>(g, a, scan); }
        return hipGetLastError();
    }
    const dim3 block(kBX, kBY), grid = grid_for(g);
    if (storage == 0) moments_kernel<0><<<grid, block, 0, s>>>(g, a);
    else moments_kernel<1><<<grid, block, 0, s>>>(g, a);
    return hipGetLastError();
}

hipError_t launch_atrous(const Geo& g, int storage, int variant, const AtrousArgs& a, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    const bool lds_ok = a.step == 1 || a.step == 2 || a.step == 4 || a.step == 8 || a.step == 16 || a.step == 32 || a.step == 64;
    // phi_normal == 0 (pow(x,0) = 1 even at x = 0) is left to the direct kernel: the fused exponent would see 0 * -inf
    if (variant != 1 /* SVGF_VARIANT_DIRECT */ && lds_ok && a.phi_normal != 0.0f)
        return storage == 0 ? launch_atrous_lds_step<0>(g, a, s) : launch_atrous_lds_step<1>(g, a, s);
    const dim3 block(kBX, kBY), grid = grid_for(g);
    if (storage == 0) atrous_direct_kernel<0><<<grid, block, 0, s>>>(g, a);
    else atrous_direct_kernel<1><<<grid, block, 0, s>>>(g, a);
    return hipGetLastError();
}

// The strip driver's one-launch iteration (AtrousRanges, svgf_kernels.h): LDS-streaming kernel only.
bool atrous_ranges_available(int variant, const AtrousArgs& a) {
    const bool lds_ok = a.step == 1 || a.step == 2 || a.step == 4 || a.step == 8 || a.step == 16 || a.step == 32 || a.step == 64;
    return variant != 1 /* SVGF_VARIANT_DIRECT */ && lds_ok && a.phi_normal != 0.0f;
}
hipError_t launch_atrous_ranges(const Geo& g, int storage, const AtrousArgs& a, const AtrousRanges& r, hipStream_t s) {
    return storage == 0 ? launch_atrous_lds_step<0>(g, a, s, &r) : launch_atrous_lds_step<1>(g, a, s, &r);
}

// Iterations 0 and 1 (steps 1 and 2) in one launch (svgf_atrous_fused.h); Geo's launch rows are iteration 1's.
bool atrous_fused_available(int variant, const AtrousArgs& a) {
    return variant != 1 /* SVGF_VARIANT_DIRECT */ && a.phi_normal != 0.0f;
}
hipError_t launch_atrous_fused(const Geo& g, int storage, const AtrousArgs& a, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    return storage == 0 ? launch_atrous_fused12<0>(g, a, s) : launch_atrous_fused12<1>(g, a, s);
}

// Albedo demodulation (MODE 0) / re-modulation (MODE 1), SURVEY.md 8f-4: pointwise, IEEE division (bit-exact vs the oracle).
template <int ST, int MODE>
__global__ __launch_bounds__(kBX* kBY) void albedo_kernel(Geo g, const void* in, const void* albedo, void* out) {
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    const float4 c = Store<ST>::ld4(in, idx), al = Store<ST>::ld4(albedo, idx);
    const float dr = fmaxf(al.x, 1e-3f), dg = fmaxf(al.y, 1e-3f), db = fmaxf(al.z, 1e-3f);
    float4 o;
    if constexpr (MODE == 0) o = make_float4(c.x / dr, c.y / dg, c.z / db, c.w);
    else {
        // (the products are pinned in fp32 registers: a product that is only rounded to half hipcc folds into v_fma_mixlo_f16 with a +0 addend,
        // which turns -0 x albedo into +0 — found by tests/fuzz_parity.py)
        float px = c.x * dr, py = c.y * dg, pz = c.z * db;
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
        o = make_float4(px, py, pz, c.w);
    }
    Store<ST>::st4(out, idx, o);
}

hipError_t launch_albedo(const Geo& g, int storage, int mode, const void* in, const void* albedo, void* out, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    const dim3 block(kBX, kBY), grid = grid_for(g);
    if (storage == 0) { if (mode == 0) albedo_kernel<0, 0><<<grid, block, 0, s>>>(g, in, albedo, out); else albedo_kernel<0, 1><<<grid, block, 0, s>>>(g, in, albedo, out); }
    else { if (mode == 0) albedo_kernel<1, 0><<<grid, block, 0, s>>>(g, in, albedo, out); else albedo_kernel<1, 1><<<grid, block, 0, s>>>(g, in, albedo, out); }
    return hipGetLastError();
}

hipError_t launch_pack_gbuffer(const Geo& g, const PackArgs& a, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    pack_gbuffer_kernel<<<grid_for(g), dim3(kBX, kBY), 0, s>>>(g, a);
    return hipGetLastError();
}


hipError_t launch_taa(const Geo& g, int storage, const void* filtered, const void* history, void* out, bool direct, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    const dim3 block(kBX, kBY);
    if (direct) {
        const dim3 grid = grid_for(g);
        if (storage == 0) taa_kernel<0><<<grid, block, 0, s>>>(g, filtered, history, out);
        else taa_kernel<1><<<grid, block, 0, s>>>(g, filtered, history, out);
    } else {
        const dim3 grid((g.W + kBX - 1) / kBX, (g.ye - g.yb + kTaaRows - 1) / kTaaRows);
        if (storage == 0) taa_lds_kernel<0><<<grid, block, 0, s>>>(g, filtered, history, out);
        else taa_lds_kernel<1><<<grid, block, 0, s>>>(g, filtered, history, out);
    }
    return hipGetLastError();
}

}  // namespace svgf
