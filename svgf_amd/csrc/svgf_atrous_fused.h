// svgf_atrous_fused.h — wavelet iterations 0 and 1 (steps 1 and 2) of application::WaveletFilter (App.cu:497-507, kernel
// Filter.cuh:527-624) in ONE streaming launch: svgf_atrous_pair.
//
// STATUS: bit-identical to the two launches (tests/test_gpu_fused.py) and ~10 % SLOWER than them on MI355X in every configuration
// measured (4K / 1080p, fp32 / fp16: profiles/r03_fused_pair_ablations.txt), so the drivers use it only on request
// (svgf_set_iteration_fusion).  Why: DESIGN.md 3.3c — the tap phase, not HBM, bounds the iterations, and the fusion trades 48 B/px
// of traffic for 10-19 % more taps.
//
// The reference launches every iteration separately, and so did this library: iteration 0 read 32 B/px and wrote 32 (its
// result twice: the ping-pong plane and the feedback plane RenderOutput, :618-622), iteration 1 read the ping-pong plane back
// (+ the guide texel) and wrote 16: 112 B/px through HBM.  Here iteration 0's rows stay in LDS for iteration 1: the pair reads
// colour + guide once (32 B/px), writes the feedback plane (16) and iteration 1's result (16) — 64 B/px and one launch
// (one ramp, one tail) less.  The ping-pong plane of iteration 0 is never materialised.
//
// A workgroup of 8 waves streams down a band of rows of a 120-column block:
//   waves 0-3  ("iteration 0"): stage two input rows per step into ring A (6 rows x 132 columns, the layout of
//              atrous_lds_kernel), filter rows i, i+1 of 128 columns [x0-4, x0+124) with step 1, store the feedback
//              texels of their own 120 columns, and write the result — rounded to the storage type and clamped exactly as
//              iteration 1's imageLoad would read it back (:78-83,586) — into ring B (12 rows x 128 columns) together with
//              the centre's depth / normal / ddepth;
//   waves 4-7  ("iteration 1"): filter rows j, j+1 of the 120 columns [x0, x0+120) with step 2 from ring B rows written in
//              EARLIER steps (iteration 1 trails iteration 0 by five steps = ten rows), and store them.
// Both halves run their 24 taps at the same time on different data; two barriers per step order the ring refills
// (ring A's two oldest rows are replaced after every wave has read them; ring B's slot of the two rows being written was last
// read in the previous step).  Iteration 0 is computed on 128 of 120 columns and on band + 8 rows, the input is read on
// 132 columns and band + 12 rows: bands are long (launcher).
//
// Results are bitwise those of the two launches: the per-pixel expressions are the same functions (svgf_atrous_taps.h),
// out-of-frame rows and columns enter ring B as what iteration 0 makes of all-zero texels (a sky centre: copied, depth = sentinel,
// so weight exactly 0 as a tap).
#pragma once
#include "svgf_atrous_taps.h"

namespace svgf {
namespace {

#ifndef SVGF_FUSED_DIAG
#define SVGF_FUSED_DIAG 0            // measurement twins only (results are wrong): 1 no iteration-0 taps, 2 no iteration-1 taps, 4 no ring refill after
#endif                               // the prologue, 8 no stores (profiles/r03_fused_pair_ablations.txt, block 3)

constexpr int kFT0 = 128;                       // iteration-0 columns of a workgroup: two waves per row
constexpr int kFReach1 = 4;                     // iteration 1 (step 2) reaches 4 rows / columns
constexpr int kFT1 = kFT0 - 2 * kFReach1;       // 120 iteration-1 columns
constexpr int kFWA = kFT0 + 4;                  // ring A columns: step-1 halo of 2 each side
constexpr int kFRA = 6;                         // ring A rows (2 produced per step + 4)
constexpr int kFRB = 12;                        // ring B rows: 10 read by iteration 1 + the 2 iteration 0 is writing
constexpr int kFLag = 5;                        // steps by which iteration 1 trails iteration 0
constexpr int kFPrefetch = 2;                   // input rows are requested this many steps before the step whose end commits them (1, 2: equal; 3: slower)
constexpr size_t kFusedLds = (size_t)16 * (kFRA * kFWA + kFRB * kFT0) + (size_t)8 * 2 * (kFRA * kFWA + kFRB * kFT0) + (size_t)4 * kFRB * kFT0 +
                             (size_t)4 * (2 * (kFRA + kFRB) + 1);

// One workgroup streaming down its band (both iterations).  EXACT = false is the product path; -> (per wave) "one of my outputs came out
// NaN", which makes the kernel run the band again with EXACT = true — see atrous_band in svgf_atrous_lds.h.
template <int ST, bool EXACT>
__device__ __forceinline__ bool fused_band(const Geo& g, const AtrousArgs& a, char* smem, int x0, int band, int j0, int j1, int nrows) {
    constexpr int CB = ST == 0 ? 16 : 8;
    constexpr int TD = 3;                          // tap pipeline depth (99 registers; LDS allows two workgroups per CU = four waves per SIMD)
    f32x4* const aA = (f32x4*)smem;                                  // ring A: iteration 0's input
    f32x4* const bA = aA + kFRA * kFWA;                              // ring B: iteration 0's output = iteration 1's input
    f32x2* const aL = (f32x2*)(bA + kFRB * kFT0);
    f32x2* const aN = aL + kFRA * kFWA;
    f32x2* const bL = aN + kFRA * kFWA;
    f32x2* const bN = bL + kFRB * kFT0;
    float* const bD = (float*)(bN + kFRB * kFT0);                    // ddepth of ring B's pixels (iteration 1's centres)
    uint32_t* const flagA = (uint32_t*)(bD + kFRB * kFT0);           // [kFRA][2]: a texel of this ring row / half differs from the reference normal
    uint32_t* const flagB = flagA + 2 * kFRA;                        // [kFRB][2]

    const int t = threadIdx.x, lane = t & 63;
    const bool second = __builtin_amdgcn_readfirstlane(t >> 8) != 0;             // waves 4-7: iteration 1
    const int rg = __builtin_amdgcn_readfirstlane((t >> 7) & 1);                 // row of the step's pair
    const int wig = __builtin_amdgcn_readfirstlane((t >> 6) & 1);                // 64-column half of the row

    const int n1 = (j1 - j0 + 1) >> 1;             // iteration-1 steps; iteration 0 runs n1 + 4 steps (rows j0-4 .. j1+3), the loop n1 + 5
    const int K0 = n1 + 4;

    const GuideSel gs(a.guide != nullptr);
    const unsigned m_off = gs.m_off, n_off = gs.n_off, n_shift = gs.n_shift;
    const unsigned npx = (unsigned)g.rows * (unsigned)g.W;
    // the workgroup's reference normal: the texel at (first iteration-1 row, x0) — inside the frame.  Every wave reads it itself.
    uint32_t ref01, refz;
    {
        const PlaneRsrc rs = plane_rsrc(a, npx, CB, n_shift, true);
        const u32x2 n = __builtin_amdgcn_raw_buffer_load_b64(rs.normal, ((unsigned)x0 << n_shift) + n_off, ((g.yb + j0 - g.y0) * g.W) << n_shift, 0);
        ref01 = __builtin_amdgcn_readfirstlane(n.x); refz = __builtin_amdgcn_readfirstlane(n.y & 0xffffu);
    }
    if (t < 2 * (kFRA + kFRB)) flagA[t] = 0u;
    unsigned long long nan_out = 0ull;             // lanes whose output held a NaN, or that staged a sky texel with a -0.0 channel (EXACT = false)
    const float phi_n = a.phi_normal;              // != 0 (launcher)
    const float inv_phi_c = hw_rcp(a.phi_colour) * kLog2e;     // log2(e) / PhiColour

    if (!second) {
        // ------------------------------------------------------------------ waves 0-3: staging + iteration 0 (step 1)
        const int col = t & (kFT0 - 1);
        const int gx = x0 - kFReach1 + col;                       // own column (may lie left of the frame in the first tile)
        const int oli = col + 2;
        const bool halo_wave = wig == 0;
        const bool has_halo = halo_wave && lane < 4;
        const int hx = lane < 2 ? gx - 2 : gx + kFT0 - 2;         // lanes 0,1: columns x0-6, x0-5; lanes 2,3: x0+124, x0+125
        const int hli = lane < 2 ? lane : kFT0 + lane;
        const bool own_ok = gx >= 0 && gx < g.W, halo_ok = has_halo && hx >= 0 && hx < g.W;
        const unsigned vo_c = own_ok ? (unsigned)gx * CB : kOob, vo_m = own_ok ? (unsigned)gx * 16u : kOob, vo_n = own_ok ? ((unsigned)gx << n_shift) + n_off : kOob;
        const unsigned vh_c = halo_ok ? (unsigned)hx * CB : kOob, vh_m = halo_ok ? (unsigned)hx * 16u : kOob, vh_n = halo_ok ? ((unsigned)hx << n_shift) + n_off : kOob;
        // the feedback texel belongs to the tile whose iteration-1 columns hold it
        const unsigned vo_fb = (own_ok && col >= kFReach1 && col < kFT0 - kFReach1) ? (unsigned)gx * CB : kOob;
        // ... and to the band whose iteration-1 rows hold it; the first / last band also own the 4 rows beyond the launch rows
        const int fb_lo = band == 0 ? -kFReach1 : j0, fb_hi = j1 == nrows ? nrows + kFReach1 : j1;

        typedef RawPx<ST, true> OwnPx;
        typedef RawPx<ST, false> HaloPx;
        struct Staged { OwnPx o; HaloPx h; };
        auto fetch = [&](int jn, Staged& st) __attribute__((always_inline)) {          // input rows jn, jn+1 (relative to g.yb): this wave's is jn + rg
            const int y = g.yb + jn + rg, yl = y - g.y0;
            const bool rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
            const int srow = rok ? yl * g.W : 0;
            const PlaneRsrc rs = plane_rsrc(a, npx, CB, n_shift, rok);
            raw_load<ST, true>(st.o, rs, vo_c, vo_m, vo_n, srow, n_shift, m_off);
            if (halo_wave) raw_load<ST, false>(st.h, rs, vh_c, vh_m, vh_n, srow, n_shift, m_off);
        };
        auto commit = [&](int sl, const Staged& st, int jn) __attribute__((always_inline)) {      // jn: as fetched (EXACT only: which texels lie outside the frame)
            int so = sl + rg; so = so >= kFRA ? so - kFRA : so;
            constexpr int noff = kFRA * kFWA * 8;
            const int at_o = so * kFWA + oli, at_h = so * kFWA + hli;
            bool row_out = false;
            if constexpr (EXACT) { const int y = g.yb + jn + rg; row_out = y < 0 || y >= g.H; }
            unsigned long long negzero = 0ull;              // (a texel with a -0.0 channel: the band is run again — commit_px, svgf_device.h)
            unsigned long long differs = commit_px<ST, true, EXACT>(st.o, lds_addr(aA) + at_o * 16, lds_addr(aL) + at_o * 8, noff, ref01, refz, true, &negzero, row_out || !own_ok);
            if constexpr (!EXACT) nan_out |= negzero;
            if (halo_wave) differs |= commit_px<ST, false, EXACT>(st.h, lds_addr(aA) + at_h * 16, lds_addr(aL) + at_h * 8, noff, ref01, refz, has_halo, nullptr, row_out || !halo_ok);
            if (lane == 0) flagA[so * 2 + wig] = differs != 0ull ? kFlagNormal : 0u;
        };
        float dq0 = 0.f, dq1 = 0.f;
        __syncthreads();                                          // the flags are zero
        // prologue: input rows j0-6 .. j0-1 (iteration 0 starts at row j0-4)
        {
            Staged s0, s1, s2;                                    // one round of memory latency (the tap loop's registers are free here)
            fetch(j0 - 6, s0); fetch(j0 - 4, s1); fetch(j0 - 2, s2);
            commit(0, s0, j0 - 6); commit(2, s1, j0 - 4); commit(4, s2, j0 - 2);
            dq0 = __uint_as_float(s1.o.zd.y); dq1 = __uint_as_float(s2.o.zd.y);
        }
        constexpr int PD = kFPrefetch;
        Staged q[PD];
#pragma unroll
        for (int d = 0; d + 1 < PD; d++) fetch(j0 + 2 * d, q[d]);   // the rows steps 0 .. PD-2 commit (K0 >= 5 steps: always needed)
        __syncthreads();

        const uint32_t refz_f = __float_as_uint(unpack_h2(refz).x);
        int slotA = 0, slotB = rg;                                // ring A slot of input row (step's first row - 2); ring B slot of this wave's row
        // cs: the rows the NEXT step needs (requested PD steps ago, committed at the end of this step); fs: requested now
        auto step = [&](int k, Staged& cs, Staged& fs) __attribute__((always_inline)) {
            const bool active = k < K0, more = k + 1 < K0;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            bool sky = true, redo = true;
            if (active) {
                if (k + PD < K0 && !(SVGF_FUSED_DIAG & 4)) fetch(j0 + 2 * (k + PD - 1), fs);
                int rowbase[5];
#pragma unroll
                for (int r = 0; r < 5; r++) { int sl = slotA + rg + r; sl = sl >= kFRA ? sl - kFRA : sl; rowbase[r] = sl * kFWA + col; }
                const int ci = rowbase[2] + 2;
                const TapCentre c = centre_setup<1>(aA[ci], aL[ci], aN[ci], dq0, inv_phi_c);
                sky = c.sky;
                const bool wave_has_surface = !(SVGF_FUSED_DIAG & 1) && wave_any(!sky);
                const bool uniform = !EXACT && !a.no_fastpath && !wave_any(lane < 2 * kFRA && flagA[lane < 2 * kFRA ? lane : 0] != 0u);
                // (EXACT: the exact form for every pixel of the second pass — the one-launch-per-iteration kernel keeps the first pass's value of the finite
                // ones (filter_px), which costs a second set of taps this kernel has no registers for.  Around a NaN texel the pair launch therefore rounds
                // as the exact form does: within the stage tolerance, not bit-identical to two launches there; the same around a texel with a -0.0 channel)
                o = filter_px<1, TD, kFRA * kFWA * 8, EXACT, false>(rows_from_index(aA, aL, rowbase), c, phi_n, wave_has_surface, uniform);
                if constexpr (!EXACT) nan_out |= lanes_where(__builtin_isunordered(o.x, o.w));
                // ring B record: the texel iteration 1 would load from the plane iteration 0 stores (:618 unclamped, in the storage type;
                // :586 imageLoad clamps — a NaN stays NaN, svgf_device.h)
                float4 q = o;
                if constexpr (ST == 1) { const float2 lo = unpack_h2(pack_h2(o.x, o.y)), hi = unpack_h2(pack_h2(o.z, o.w)); q = make_float4(lo.x, lo.y, hi.x, hi.y); }
                const f32x2 q01 = clamp01_pk((f32x2){q.x, q.y}), q23 = clamp01_pk((f32x2){q.z, q.w});
                if constexpr (!EXACT) q = make_float4(q01.x, q01.y, q23.x, q23.y);
                else {                                                              // (-0.0 kept: clamp01_ref; a pixel outside the frame: the sums' identity, commit_px)
                    q = make_float4(q.x == 0.0f ? q.x : q01.x, q.y == 0.0f ? q.y : q01.y, q.z == 0.0f ? q.z : q23.x, q.w == 0.0f ? q.w : q23.y);
                    const int yi = g.yb + j0 - kFReach1 + 2 * k + rg;
                    if (yi < 0 || yi >= g.H || !own_ok) q = make_float4(-0.0f, -0.0f, -0.0f, -0.0f);
                }
                const int bi = slotB * kFT0 + col;
                bA[bi] = (f32x4){q.x, q.y, q.z, q.w};
                bL[bi] = (f32x2){lum_exact(q.x, q.y, q.z), sky ? kSkyZ : c.lz.y};
                bN[bi] = (f32x2){__uint_as_float(c.n01), c.nz};
                bD[bi] = dq0;
                const bool differs = !sky && (c.n01 != ref01 || __float_as_uint(c.nz) != refz_f);
                const bool wave_differs = wave_any(differs);
                if (lane == 0) flagB[slotB * 2 + wig] = wave_differs ? kFlagNormal : 0u;
            }
            lds_barrier();                                        // every wave is done reading ring A's two oldest rows
            if (more && !(SVGF_FUSED_DIAG & 4)) {
                commit(slotA, cs, j0 + 2 * k);
                dq0 = dq1; dq1 = __uint_as_float(cs.o.zd.y);
                slotA += 2; if (slotA >= kFRA) slotA -= kFRA;
            }
            lds_barrier();
            if (active) {
                const int i = j0 - kFReach1 + 2 * k + rg, y = g.yb + i;                  // this wave's iteration-0 row
                if (i >= fb_lo && i < fb_hi && y >= 0 && y < g.H && !((SVGF_FUSED_DIAG & 8) && o.x != 12345.678f)) {          // scalar
                    const int srow = (y - g.y0) * g.W;
                    const __amdgpu_buffer_rsrc_t rs_fb = __builtin_amdgcn_make_buffer_rsrc(a.feedback ? a.feedback : a.out, 0, a.feedback ? (int)(npx * CB) : 0, 0x00020000);
                    if constexpr (ST == 0) {
                        const u32x4 raw = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(raw, rs_fb, sky || (EXACT && !redo) ? kOob : vo_fb, srow * CB, 0);       // :619-622 (not for sky)
                    } else {
                        const u32x2 raw = {pack_h2(o.x, o.y), pack_h2(o.z, o.w)};
                        __builtin_amdgcn_raw_buffer_store_b64(raw, rs_fb, sky || (EXACT && !redo) ? kOob : vo_fb, srow * CB, 0);
                    }
                }
            }
            slotB += 2; if (slotB >= kFRB) slotB -= kFRB;
        };
        for (int k = 0; k <= K0; k += PD) {
#pragma unroll
            for (int u = 0; u < PD; u++) if (k + u <= K0) step(k + u, q[u], q[(u + PD - 1) % PD]);
        }
    } else {
        // ------------------------------------------------------------------ waves 4-7: iteration 1 (step 2) from ring B
        const int c1 = wig * 64 + lane;                            // 0 .. 127, 120 of them are columns of the tile
        const bool col_ok = c1 < kFT1 && x0 + c1 < g.W;
        const int col = c1 < kFT1 ? c1 : kFT1 - 1;                 // lanes beyond the tile repeat its last column (nothing is stored)
        const unsigned vo_c = col_ok ? (unsigned)(x0 + c1) * CB : kOob;
        __syncthreads();
        __syncthreads();
        int slotB = 0;                                             // ring B slot of iteration-0 row (j - 4), j = the step's first row
        for (int k = 0; k <= K0; k++) {
            const bool active = k >= kFLag;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            bool redo = true;                                      // EXACT: only the texels whose first-pass result held a NaN are stored again (filter_px)
            if (active) {
                int rowbase[5];
#pragma unroll
                for (int r = 0; r < 5; r++) { int sl = slotB + rg + 2 * r; sl = sl >= kFRB ? sl - kFRB : sl; rowbase[r] = sl * kFT0 + col; }
                const int ci = rowbase[2] + kFReach1;
                const TapCentre c = centre_setup<2>(bA[ci], bL[ci], bN[ci], bD[ci], inv_phi_c);
                const bool wave_has_surface = !(SVGF_FUSED_DIAG & 2) && wave_any(!c.sky);
                const bool uniform = !EXACT && !a.no_fastpath && !wave_any(lane < 2 * kFRB && flagB[lane < 2 * kFRB ? lane : 0] != 0u);
                o = filter_px<2, TD, kFRB * kFT0 * 8, EXACT, false>(rows_from_index(bA, bL, rowbase), c, phi_n, wave_has_surface, uniform);
                if constexpr (!EXACT) nan_out |= lanes_where(__builtin_isunordered(o.x, o.w));
            }
            lds_barrier();
            if (active) {
                const int j = j0 + 2 * (k - kFLag) + rg;
                if (j < j1 && !((SVGF_FUSED_DIAG & 8) && o.x != 12345.678f)) {                                      // scalar
                    const int srow = (g.yb + j - g.y0) * g.W;
                    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(npx * CB), 0x00020000);
                    const unsigned so_c = EXACT && !redo ? kOob : vo_c;
                    if constexpr (ST == 0) {
                        const u32x4 raw = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, so_c, srow * CB, 0);                    // :618
                    } else {
                        const u32x2 raw = {pack_h2(o.x, o.y), pack_h2(o.z, o.w)};
                        __builtin_amdgcn_raw_buffer_store_b64(raw, rs_out, so_c, srow * CB, 0);
                    }
                }
                slotB += 2; if (slotB >= kFRB) slotB -= kFRB;
            }
            lds_barrier();
        }
    }
    return nan_out != 0ull;
}

template <int ST>
__global__ __launch_bounds__(512, 4) void atrous_fused12_kernel(Geo g, AtrousArgs a, int band_rows, int nbands, int xgroup, int xrot) {
    keep_nan_in_clamps();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // tile order: x tile fastest, XCD-aware groups (svgf_device.h); the frame is walked bottom-up: what the temporal launch wrote
    // last is still in the Infinity Cache when it is read first
    const int xtiles = (g.W + kFT1 - 1) / kFT1;
    const int ntiles = xtiles * nbands;
    int v = xcd_tile(xgroup, xrot);
    if (v >= ntiles) return;
    v = ntiles - 1 - v;
    const int x0 = (v % xtiles) * kFT1;
    const int band = v / xtiles;
    const int nrows = g.ye - g.yb;                 // iteration-1 rows
    const int j0 = band * band_rows;
    if (j0 >= nrows) return;
    const int j1 = min(nrows, j0 + band_rows);
    uint32_t* const nan_word = (uint32_t*)(smem + kFusedLds) - 1;    // the workgroup's "an output was NaN" word (last word of the allocation)
    if (threadIdx.x == 0) *nan_word = 0u;          // (ordered before the waves' stores below by the band's barriers)
    const bool nan_wave = fused_band<ST, false>(g, a, smem, x0, band, j0, j1, nrows);
    if (nan_wave && (threadIdx.x & 63) == 0) *nan_word = 1u;
    __syncthreads();
    if (*nan_word == 0u) return;                   // every frame without a NaN
    __syncthreads();
    (void)fused_band<ST, true>(g, a, smem, x0, band, j0, j1, nrows);
}

// The launch rows of Geo are ITERATION 1's rows; iteration 0 runs on them and kFReach1 rows beyond on either side (inside the
// frame), which is where the feedback plane is written.  The caller has checked that the planes hold kFReach1 + 2 rows around.
template <int ST>
hipError_t launch_atrous_fused12(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(atrous_fused12_kernel<ST>, kFusedLds, attr_done); e != hipSuccess) return e;
    const int nrows = g.ye - g.yb;
    const int xtiles = (g.W + kFT1 - 1) / kFT1;
    // Two workgroups per CU (LDS).  A band pays 12 extra input rows and 8 extra iteration-0 rows: bands are as long as two rounds
    // of resident workgroups allow (one round, 136 rows at 4K: +6 %; four, 34 rows: +3 %), and not shorter than 32 rows.
    int nbands = 2 * num_cus() * 2 / xtiles;
    if (nbands < 1) nbands = 1;
    int band = (nrows + nbands - 1) / nbands;
    if (band < 32) band = 32;
    band = (band + 1) / 2 * 2;
    nbands = (nrows + band - 1) / band;
    int xgroup;
    const dim3 grid = xcd_grid(xtiles * nbands, 16, xgroup);
    atrous_fused12_kernel<ST><<<grid, dim3(512), kFusedLds, s>>>(g, a, band, nbands, xgroup, 3);
    return hipGetLastError();
}

}  // namespace
}  // namespace svgf
