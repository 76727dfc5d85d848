// svgf_atrous_lds.h — one iteration of the wavelet filter (filter::FilterKernel, Filter.cuh:527-624) as an LDS-streaming kernel
// for CDNA4, steps 1..64.  (Rounds 2-4 kept an instrumented twin of this kernel and three alternative kernels under tools/variants/ for the
// ablations of DESIGN.md 3.3; their results live in profiles/r02_* .. r04_*, the code left the tree in round 5.)
//
// The reference gathers 25 taps x 3 textures per pixel.  For step S a pixel only ever reads pixels of its own row residue
// (y mod S), so a workgroup (4 waves) owns ONE residue of a band of rows and a 128-column block and STREAMS DOWN the band: a ring
// of 6 decimated rows x (128 + 4S) columns lives in LDS as fp32 records (svgf_device.h); every step waves 0-1 produce decimated
// row j and waves 2-3 row j+1 from the ring (one output per thread, svgf_atrous_taps.h), then the two oldest ring rows are
// replaced by the two rows requested from HBM at the start of the step.  Global loads are always full-width row segments
// (16 B per lane, coalesced) whatever the step; the y over-fetch is (band+4)/band and the x over-fetch (128+4S)/128 instead of the
// 25x gather of a per-pixel kernel.
//
// The centre's ddepth is the only per-pixel input that is not in the records: the thread that stages a pixel of its own column
// is the thread that filters it two steps later, so ddepth rides in a two-register queue.  Everything that is the same for all
// lanes of a wave — row offsets, ring slots, validity of a row — lives in scalar registers; staging a row costs no vector ALU.
//
// Uniform-normal fast path (bit-identical): nflag[slot][wave] = "a surface texel of this ring row staged by this wave differs
// from the workgroup's reference normal"; while no flag is set, n.n' is each centre's own |n|^2 (taps24<.., UNI>).
#pragma once
#include "svgf_atrous_taps.h"

namespace svgf {
namespace {

constexpr int kTX = 128;                 // columns of a workgroup: two waves per row, two rows per step
constexpr int kTapDepth = 3;             // LDS reads run this many taps ahead of the arithmetic
constexpr int kAtrousOversubscribe = 4;   // workgroups per resident slot of the chip the bands are cut for ...
constexpr int kAtrousMinBand = 8;         // ... but no band shorter than this many (decimated) rows: 4 more are fetched per band.  (Round 4 swept
                                          // 2x / 4x and 8 / 12 / 16 rows on an 8K/8 strip, 1080p and 4K, interleaved on one device: 4x and 8 rows are
                                          // best or equal everywhere, profiles/r04_small_experiments.txt)
// Resident waves per SIMD the kernel of step S is compiled for (registers: 95 -> five).  At step 16 the ring (37 KB) allows four
// workgroups per CU anyway, at step 32 (49 KB) three, at step 64 (74 KB) two.
constexpr int atrous_waves(int S) { return S <= 8 ? 5 : S == 16 ? 4 : S == 32 ? 3 : 2; }

// LDS layout of a workgroup, as byte addresses (no generic pointers: an address-space cast of a pointer the compiler cannot see
// through costs a null check per use): colour records (16 B x kRing x WL), {luminance, depth} records (8 B x ..), normal records
// (8 B x ..), the ring rows' flag words [kRing][8], the reference normal {(nx,ny) bits, nz bits}, the "an output was NaN" word.
template <int S, int TX> struct AtrousLds {
    static constexpr int WL = TX + 4 * S;
    uint32_t a;                                                         // colour records
    __device__ __forceinline__ uint32_t l() const { return a + kRing * WL * 16; }
    static constexpr int NOFF = kRing * WL * 8;                         // bytes from a pixel's {luminance, depth} record to its normal record
    __device__ __forceinline__ uint32_t flag(int i) const { return l() + 2 * NOFF + 4 * i; }
    __device__ __forceinline__ uint32_t nref(int i) const { return flag(kRing * 8 + i); }
    // nref(3): the waves that have finished; nref(4..11): what the end of a signalling workgroup needs (stashed by thread 0 at the start);
    // nref(12..13): AtrousArgs::path_stats
    static constexpr size_t bytes = (size_t)kRing * WL * kRecBytes + (kRing * 8 + 14) * sizeof(uint32_t);
};
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_load(uint32_t addr) { return *(const lds_u32*)(uintptr_t)addr; }
__device__ __forceinline__ void lds_store(uint32_t addr, uint32_t v) { *(lds_u32*)(uintptr_t)addr = v; }

// One workgroup streaming down its band: decimated rows [j0, j1) of the residue whose row j is global row ybase + S*j, columns
// [x0, x0 + kTX).  EXACT = false is the product path.  -> the wave's report (`state` below).
//
// NaN.  The reference's clamp keeps a NaN texel (svgf_device.h), and so do the records here.  The reference's weight then stays FINITE —
// `max(weightLillum, 0.0)` in Filter.cuh:424 is CUDA's fmax, which drops the NaN — and the NaN reaches the sums through the channels
// that hold it (:608); a sky centre is copied whatever its taps hold (:554-558).  The fast taps fold |dl| / phi_l into one FMA of the
// exponent: a NaN luminance makes their weight, and with it ALL FOUR channels of the output, NaN (and copy a sky centre through
// weights that are exactly 0 — but 0 x NaN is not 0).  So any NaN in a pixel's window shows in its fast-path output — as a NaN in
// every channel, or (a NaN variance only) in the variance channel — and costs the product path ONE compare per output.  The
// kernel then runs the band again with EXACT = true: the general taps in the form that evaluates the luminance term as the
// reference does (taps24<.., kTapsNaN>) and a select for the sky centres.  A workgroup whose texels are all finite never gets there.
// WT: the band's stores are written THROUGH to memory (sc0 sc1) — the ranges that signal (RangePlan): their rows are read by another kernel (RCCL's) while
// this launch is still running, and making them visible with release fences instead means an L2 write-back per workgroup (measured: the one-launch
// iteration 0.17 ms SLOWER per frame than three launches, profiles/r05_small_experiments.txt)
template <int ST, int S, int TX, bool EXACT>
__device__ __forceinline__ uint32_t atrous_band(const Geo& g, const AtrousArgs& a, const AtrousLds<S, TX>& L, int x0, int j0, int j1, int ybase, bool wt = false) {
    constexpr int WL = TX + 4 * S;                 // staged columns per ring row
    constexpr int CB = ST == 0 ? 16 : 8;           // bytes per colour texel
    constexpr int NH = 4 * S;                      // halo pixels per ring row.  Steps 1-16: all staged by wave 0 of the row group (lanes
                                                   // 0..NH-1; spread over the waves, every wave paid the halo's ~20 VALU + 3 loads for a few
                                                   // lanes).  Steps 32 and 64: 128 / 256 of them — HP full passes on BOTH waves of the row group
    constexpr int WPR = TX / 64;                   // waves per row group
    constexpr int HP = NH <= 64 ? 1 : NH / (64 * WPR);    // halo passes per staging wave
    constexpr bool kBothWaves = NH > 64;
    static_assert(NH <= 64 || NH == 64 * WPR * HP, "halo passes must be whole waves");

    int t = threadIdx.x;
    // (the second pass derives its per-lane constants from a thread index the compiler cannot identify with the first pass's: shared
    // between the two passes they would stay live through the product loop, which has no register to spare — it spilled)
    if constexpr (EXACT) asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int col = t % TX;
    const int rg = __builtin_amdgcn_readfirstlane(t / TX);          // row group: wave-uniform -> scalar
    const int wig = __builtin_amdgcn_readfirstlane((t % TX) >> 6);  // wave index inside its row group

    // per-lane constants
    const int gx = x0 + col;                       // own column
    const bool halo_wave = kBothWaves || wig == 0; // scalar
    const GuideSel gs(a.guide != nullptr);
    const bool own_ok = gx < g.W;
    const unsigned vo_c = own_ok ? (unsigned)gx * CB : kOob, vo_m = own_ok ? (unsigned)gx * 16u : kOob, vo_n = own_ok ? ((unsigned)gx << gs.n_shift) + gs.n_off : kOob;
    // halo pixel of pass p: index hidx in [0, NH) along the row's 2S left + 2S right halo columns
    bool has_halo[HP];
    unsigned vh_c[HP], vh_m[HP], vh_n[HP];
    int hli[HP];
#pragma unroll
    for (int p = 0; p < HP; p++) {
        const int hidx = kBothWaves ? (wig * HP + p) * 64 + lane : lane;
        has_halo[p] = halo_wave && hidx < NH;      // this lane also stages one halo pixel per pass and row of its row group
        const int hx = (hidx < 2 * S) ? x0 - 2 * S + hidx : x0 + TX + hidx - 2 * S;
        hli[p] = (hidx < 2 * S) ? hidx : TX + hidx;
        const bool halo_ok = has_halo[p] && hx >= 0 && hx < g.W;
        vh_c[p] = halo_ok ? (unsigned)hx * CB : kOob; vh_m[p] = halo_ok ? (unsigned)hx * 16u : kOob; vh_n[p] = halo_ok ? ((unsigned)hx << gs.n_shift) + gs.n_off : kOob;
    }
    const unsigned npx = (unsigned)g.rows * (unsigned)g.W;
    constexpr int NOFF = AtrousLds<S, TX>::NOFF;
    const uint32_t colA = L.a + (uint32_t)col * 16u, colL = L.l() + (uint32_t)col * 8u;    // this column's records of ring row 0
    uint32_t haloA[HP], haloL[HP];                                                          // ... and of its halo pixel(s)
#pragma unroll
    for (int p = 0; p < HP; p++) { haloA[p] = L.a + (uint32_t)hli[p] * 16u; haloL[p] = L.l() + (uint32_t)hli[p] * 8u; }

    // A thread's share of one staged step: its own pixel of row (jn + rg), and a halo pixel on lanes < NH.  Buffer resources are
    // built where they are used (a scalar select of num_records) instead of being kept in SGPRs for the whole kernel.
    struct Staged { RawPx<ST, true> o; RawPx<ST, false> h[HP]; };
    auto fetch = [&](int jn, Staged& st) __attribute__((always_inline)) {
        const int y = ybase + S * (jn + rg), yl = y - g.y0;                             // scalar
        const bool rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
        const int srow = rok ? yl * g.W : 0;
        const PlaneRsrc rs = plane_rsrc(a, npx, CB, gs.n_shift, rok);
        raw_load<ST, true>(st.o, rs, vo_c, vo_m, vo_n, srow, gs.n_shift, gs.m_off);
        if (halo_wave) {
#pragma unroll
            for (int p = 0; p < HP; p++) raw_load<ST, false>(st.h[p], rs, vh_c[p], vh_m[p], vh_n[p], srow, gs.n_shift, gs.m_off);
        }
    };
    uint32_t ref01 = 0, refz = 0;
    // What the wave reports of its band (EXACT = false), ONE scalar register (the kernel has none to spare: one more and the S = 2 .. 8 kernels spill):
    //   bit 31      an output held a NaN, or an own texel holds a -0.0 channel (commit_px): the band is run again;
    //   bits 0-15   steps in which the wave filtered a surface pixel, bits 16-30 those of them on the uniform-normal path (svgf_path_stats_enable).
    uint32_t state = 0u;
    auto rerun_if = [&](unsigned long long lanes) { state |= (((uint32_t)lanes | (uint32_t)(lanes >> 32)) != 0u) ? 0x80000000u : 0u; };
    // jn: the decimated row the step fetched (fetch(jn, st)); EXACT only: which texels lie outside the frame
    auto commit = [&](int sl, const Staged& st, int jn) __attribute__((always_inline)) {
        int so = sl + rg; so = so >= kRing ? so - kRing : so;                           // scalar
        bool row_out = false;
        if constexpr (EXACT) { const int y = ybase + S * (jn + rg); row_out = y < 0 || y >= g.H; }
        // (own pixel: column + 2S of the ring row — a scalar added to the column's address, no register of its own)
        unsigned long long negzero = 0ull;
        unsigned long long differs = commit_px<ST, true, EXACT>(st.o, colA + (uint32_t)(so * (WL * 16) + 2 * S * 16), colL + (uint32_t)(so * (WL * 8) + 2 * S * 8), NOFF, ref01, refz,
                                                                true, &negzero, row_out || !own_ok);
        if constexpr (!EXACT) rerun_if(negzero);
        if (halo_wave) {
#pragma unroll
            for (int p = 0; p < HP; p++)
                differs |= commit_px<ST, false, EXACT>(st.h[p], haloA[p] + (uint32_t)(so * (WL * 16)), haloL[p] + (uint32_t)(so * (WL * 8)), NOFF, ref01, refz, has_halo[p], nullptr,
                                                       row_out || vh_c[p] == kOob);
        }
        if (lane == 0) lds_store(L.flag(so * 8 + wig), differs != 0ull ? kFlagNormal : 0u);   // a ring slot is always staged by the same waves
    };

    // ddepth of this thread's next two centres (rows j+rg and two rows further).  A staged row becomes a centre two steps after it
    // is committed; its ddepth is taken over at commit time (never at fetch time: that would wait for the prefetch).
    float dq0 = 0.f, dq1 = 0.f;
    // prologue: two ring rows at a time, three dependent rounds of memory latency.  (All six rows requested at once measure the
    // same at 4K and 1 % slower at 1080p: profiles/r03_small_experiments.txt.)  Rows j0, j0+1 (always inside the frame) go first:
    // thread 0's pixel of row j0 is the workgroup's reference normal.
    if (t < kRing * 8) lds_store(L.flag(0) + 4 * t, 0u);
#pragma unroll 1
    for (int rr = 0; rr < kRing; rr += kRS) {
        const int r = rr == 0 ? 2 : (rr == 2 ? 0 : rr);
        Staged st;
        fetch(j0 - 2 + r, st);
        if (rr == 0) {
            if (t == 0) { lds_store(L.nref(0), st.o.n.x); lds_store(L.nref(1), st.o.n.y & 0xffffu); }
            __syncthreads();
            ref01 = lds_load(L.nref(0)); refz = lds_load(L.nref(1));
        }
        commit(r, st, j0 - 2 + r);
        if (r == 2) dq0 = __uint_as_float(st.o.zd.y);
        if (r == 4) dq1 = __uint_as_float(st.o.zd.y);
    }
    __syncthreads();

    const float phi_n = a.phi_normal;              // != 0 (launcher)
    const float inv_phi_c = hw_rcp(a.phi_colour) * kLog2e;  // log2(e) / PhiColour (wave-uniform)
    // the uniform-normal path's exponent bases, once per workgroup: wave-uniform values (scalar registers)
    UniBase ref_base = uni_base(ref01, unpack_h2(refz).x, phi_n);
#pragma unroll
    for (int k = 0; k < 5; k++) ref_base.e[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ref_base.e[k])));
    int slot0 = 0;
    Staged cs;                                     // the rows the NEXT step needs: requested at the start of a step, committed at its end
    for (int j = j0; j < j1; j += kRS) {
        const bool more = (j + kRS) < j1;
        if (more) fetch(j + kRS + 2, cs);

        // this thread's centre is ring row 2+rg, its taps ring rows rg .. rg+4, columns oli-2S .. oli+2S
        TapRows rows;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            int sl = slot0 + rg + r; sl = sl >= kRing ? sl - kRing : sl;                 // scalar
            rows.a[r] = colA + (uint32_t)(sl * (WL * 16)); rows.l[r] = colL + (uint32_t)(sl * (WL * 8));
        }
        const uint32_t ca = rows.a[2] + 2 * S * 16, cl = rows.l[2] + 2 * S * 8;
        const TapCentre c = centre_setup<S>(lds_read_a(ca), lds_read_l(cl), lds_read_l(cl + NOFF), dq0, inv_phi_c);
        const bool sky = c.sky;
        const bool wave_has_surface = wave_any(!sky);
        // (a reference normal that holds a NaN: no uniform form — its exponent bases are NaN for EVERY centre of the workgroup, the sky texels
        // that are to be copied included, and the second pass, which looks at the general taps, would not see what the first had stored)
        const bool uniform = !EXACT && !a.no_fastpath && ref_base.e[0] == ref_base.e[0] &&
                             !wave_any(lane < kRing * 8 && lds_load(L.flag(0) + 4 * (lane < kRing * 8 ? lane : 0)) != 0u);
        bool redo = true;                          // EXACT: this lane's first-pass result held a NaN — only those texels are stored again
        // (&state: svgf_path_stats_enable's two counts, added where filter_px branches anyway — one scalar add per step, no register of their own)
        const float4 o = filter_px<S, kTapDepth, NOFF, EXACT>(rows, c, phi_n, wave_has_surface, uniform, &ref_base, EXACT ? &redo : nullptr, EXACT ? nullptr : &state);
        if constexpr (!EXACT) rerun_if(lanes_where(__builtin_isunordered(o.x, o.w)));
        else redo = redo | has_negzero(make_float4(c.A.x, c.A.y, c.A.z, c.A.w));            // the sign of a zero (commit_px): only such a centre can come out -0.0

        // Output values now, their stores AFTER the ring refill: hipcc's vmcnt bookkeeping cannot tell that the rows committed
        // below were fetched long before this step's stores, so stores issued first would be waited for.
        if (more) {
            lds_barrier();                         // every wave is done reading the two oldest ring rows
            commit(slot0, cs, j + kRS + 2);
            dq0 = dq1; dq1 = __uint_as_float(cs.o.zd.y);                                 // row j+4+rg: the centre two steps on
            slot0 += kRS; if (slot0 >= kRing) slot0 -= kRing;
            lds_barrier();
        }
        if (j + rg < j1) {                                                               // scalar
            const int srow = (ybase + S * (j + rg) - g.y0) * g.W;
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(npx * CB), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_fb = __builtin_amdgcn_make_buffer_rsrc(a.feedback ? a.feedback : a.out, 0, a.feedback ? (int)(npx * CB) : 0, 0x00020000);
            // columns outside the frame carry the out-of-range offset: the store is dropped by the range check
            const unsigned so_c = EXACT && !redo ? kOob : vo_c;                                             // (EXACT = false: vo_c, a constant)
            constexpr int kWT = 1 | 16;                                                                     // sc0 | sc1: system-scope write-through
            if constexpr (ST == 0) {
                const u32x4 raw = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                if (wt) {                                                                                   // (scalar)
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, so_c, srow * CB, kWT);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_fb, sky ? kOob : so_c, srow * CB, kWT);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, so_c, srow * CB, 0);               // :618
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_fb, sky ? kOob : so_c, srow * CB, 0);   // :619-622 (not for sky)
                }
            } else {
                const u32x2 raw = {pack_h2(o.x, o.y), pack_h2(o.z, o.w)};
                if (wt) {
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_out, so_c, srow * CB, kWT);
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_fb, sky ? kOob : so_c, srow * CB, kWT);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_out, so_c, srow * CB, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_fb, sky ? kOob : so_c, srow * CB, 0);
                }
            }
        }
    }
    return state;
}

// What the kernel knows of its row ranges.  One range (every launch but the strip driver's): rows [g.yb, g.ye).  Several (AtrousRanges): the
// first `first_blocks` workgroup ids serve the first ranges' `first_tiles` tiles (XCD-aware order of their own), the others the last range.
struct RangePlan {
    int nranges, nfirst;
    int yb[3], ye[3], band[3], nbands[3], tiles_end[3];      // tiles_end: cumulative over the first ranges; [nranges - 1]: the last range's own count
    int first_blocks, first_tiles, xgroup_first, xgroup;
    unsigned long long* signal; unsigned* arrivals; unsigned long long value;
};

template <int ST, int S, int TX>
__global__ __launch_bounds__(TX * kRS, atrous_waves(S)) void atrous_lds_kernel(Geo g, AtrousArgs a, RangePlan rp) {
    keep_nan_in_clamps();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AtrousLds<S, TX> L{lds_addr(smem)};
    // tile order v = (residue, band, x tile), x fastest.  Step 1 walks the frame bottom-up: what the temporal launch wrote last is
    // still in the 256 MB Infinity Cache when it is read first (-4.5 % for that launch).  Later steps sweep the frame once per row
    // residue; of the six direction patterns tried for them, only "step 8 backwards too" measured better than all forwards (-1.5 to
    // -3 % for that launch, -0.7 % for step 16 after it; steps 2 and 4 backwards are 1-4 % slower: profiles/r03_small_experiments.txt)
    const int xtiles = (g.W + TX - 1) / TX;
    const bool first = (int)blockIdx.x < rp.first_blocks;         // (scalar) a tile of the ranges that signal
    // (the range's numbers are SELECTED, never indexed: a run-time index into a by-value argument makes the compiler copy it to scratch)
    int v, r = rp.nranges - 1;
    auto pick3 = [&](int f0, int f1, int f2) { return r == 0 ? f0 : r == 1 ? f1 : f2; };
#define pick(f) pick3(rp.f[0], rp.f[1], rp.f[2])
    if (first) {
        v = xcd_tile_of((int)blockIdx.x, rp.xgroup_first, 3);
        if (v >= rp.first_tiles) return;                          // padding
        r = 0;
        if (rp.nfirst > 1 && v >= rp.tiles_end[0]) { r = 1; v -= rp.tiles_end[0]; }
    } else {
        const int ntiles = pick(tiles_end);
        v = xcd_tile_of((int)blockIdx.x - rp.first_blocks, rp.xgroup, 3);
        if (v >= ntiles) return;                                  // padding of the last groups
        if (S == 1 || S == 8) v = ntiles - 1 - v;
    }
    const int yb = pick(yb), nrows = pick(ye) - yb, nbands = pick(nbands), band_rows = pick(band);
#undef pick
    const int x0 = (v % xtiles) * TX;
    const int band = (v / xtiles) % nbands;
    const int rv = v / (xtiles * nbands);          // row residue (relative to yb) this workgroup owns
    const int nj = (nrows - rv + S - 1) / S;       // decimated rows of this residue
    const int j0 = band * band_rows;
    if (threadIdx.x == 0) {                        // (ordered before the waves' stores below by the band's barriers)
        lds_store(L.nref(2), 0u); lds_store(L.nref(3), 0u);
        // what the end of the workgroup needs to signal, kept in LDS: in scalar registers through the band it spilled the product loop's
        // (the kernel sits at its SGPR and VGPR limits; tests/test_kernel_budgets.py)
        lds_store(L.nref(4), first ? 1u : 0u);
        lds_store(L.nref(12), (uint32_t)(uintptr_t)a.path_stats); lds_store(L.nref(13), (uint32_t)((uintptr_t)a.path_stats >> 32));
        if (first) {
            lds_store(L.nref(5), (uint32_t)(uintptr_t)rp.arrivals); lds_store(L.nref(6), (uint32_t)((uintptr_t)rp.arrivals >> 32));
            lds_store(L.nref(7), (uint32_t)(uintptr_t)rp.signal); lds_store(L.nref(8), (uint32_t)((uintptr_t)rp.signal >> 32));
            lds_store(L.nref(9), (uint32_t)rp.value); lds_store(L.nref(10), (uint32_t)(rp.value >> 32));
            lds_store(L.nref(11), (uint32_t)rp.first_tiles);
        }
    }
    if (j0 >= nj) __syncthreads();
    else {
        const int j1 = min(nj, j0 + band_rows);
        const int ybase = yb + rv;                 // global row of decimated index j: ybase + S*j
        const uint32_t report = atrous_band<ST, S, TX, false>(g, a, L, x0, j0, j1, ybase, first);
        if ((report >> 31) != 0u && (threadIdx.x & 63) == 0) lds_store(L.nref(2), 1u);
        __syncthreads();
        if ((lds_load(L.nref(12)) | lds_load(L.nref(13))) != 0u) {      // diagnostics (svgf_path_stats_enable); off: two LDS words read per wave
            unsigned long long* stats = (unsigned long long*)((uintptr_t)lds_load(L.nref(12)) | ((uintptr_t)lds_load(L.nref(13)) << 32));
            if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u && (report & 0xffffu) != 0u) {
                (void)__hip_atomic_fetch_add(stats, (unsigned long long)(report & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(stats + 1, (unsigned long long)((report >> 16) & 0x7fffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lds_load(L.nref(2)) != 0u) {           // (no frame without a NaN or a -0.0 texel gets here)
            __syncthreads();                       // (the band's prologue writes the flag words again)
            (void)atrous_band<ST, S, TX, true>(g, a, L, x0, j0, j1, ybase, lds_load(L.nref(4)) != 0u);
        }
    }
    // The rows a neighbour waits for: every wave's stores are made visible (release: vmcnt(0) + L2 write-back) and the wave counts itself in
    // LDS; the last wave counts the workgroup in device memory, and the last workgroup publishes the value the communication stream waits
    // for (hipStreamWaitValue64).  No thread index, no argument is used here: they would have to live through the band.
    if (lds_load(L.nref(4)) != 0u) {
        // this wave's stores were written through (atrous_band, WT): once they have been acknowledged they are in memory — no release fence, which
        // on this part is a write-back of the XCD's whole L2, per workgroup, under the interior tiles' feet.
        // This is OUTSIDE the compiler's memory model and rests on three hardware facts: (1) a store with sc0 sc1 is written through to
        // memory, never left dirty in this XCD's L2; (2) s_waitcnt vmcnt(0) returns only when the memory side has acknowledged the wave's stores;
        // (3) whoever reads these rows is a kernel launched LATER on THIS device — RCCL's send kernel behind hipStreamWaitValue64, whose launch
        // invalidates the caches it reads through; no other device reads them in place.  Checked bit for bit on one device over RCCL's own kernels
        // and over the mailbox (tests/test_gpu_strips_mailbox.py, tests/fuzz_parity.py); on real peers `bench.py --gpus N` checks the strips
        // against the one-GPU frame itself and keeps the three-launch schedule unless that check passes.  Hence an opt-in
        // (svgf_strips_set_edge_first, default 0): the default schedule orders the exchange behind an event, inside the memory model.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (lane == 0 && __hip_atomic_fetch_add((lds_u32*)(uintptr_t)L.nref(3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u == (unsigned)(TX * kRS / 64)) {
            unsigned* arrivals = (unsigned*)((uintptr_t)lds_load(L.nref(5)) | ((uintptr_t)lds_load(L.nref(6)) << 32));
            const unsigned n = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n + 1u == lds_load(L.nref(11))) {
                unsigned long long* signal = (unsigned long long*)((uintptr_t)lds_load(L.nref(7)) | ((uintptr_t)lds_load(L.nref(8)) << 32));
                __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(signal, (unsigned long long)lds_load(L.nref(9)) | ((unsigned long long)lds_load(L.nref(10)) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// bands of one range: (x tiles) x (S residues) x (bands) is kAtrousOversubscribe times the resident slots of the chip (LDS: 160 KiB per CU;
// registers: atrous_waves(S) waves per SIMD): workgroups that take a fast path (all sky, uniform normals) make room for others
// instead of idling until the slowest one of a single round finishes (A/B on one device: 2x -3..5 %, 4x another -1.5 %, 6x worse).
template <int S, int TX>
inline void cut_bands(int nrows, int xtiles, size_t lds, int& band, int& nbands) {
    const int per_cu_lds = (int)((160 * 1024) / lds), per_cu_waves = atrous_waves(S) * 4 / (TX * kRS / 64);      // workgroups of TX * kRS / 64 waves
    const int per_cu = per_cu_lds < per_cu_waves ? per_cu_lds : per_cu_waves;
    const int njmax = (nrows + S - 1) / S;
    nbands = per_cu * num_cus() * kAtrousOversubscribe / (xtiles * S);
    if (nbands < 1) nbands = 1;
    band = (njmax + nbands - 1) / nbands;
    if (band < kAtrousMinBand) band = kAtrousMinBand;
    band = (band + kRS - 1) / kRS * kRS;
    nbands = (njmax + band - 1) / band;
}

template <int ST, int S, int TX>
hipError_t launch_atrous_lds_tx(const Geo& g, const AtrousArgs& a, hipStream_t s, const AtrousRanges* ranges = nullptr) {
    constexpr size_t lds = AtrousLds<S, TX>::bytes;
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(atrous_lds_kernel<ST, S, TX>, lds, attr_done); e != hipSuccess) return e;
    const int xtiles = (g.W + TX - 1) / TX;
    RangePlan rp{};
    if (!ranges) { rp.nranges = 1; rp.nfirst = 0; rp.yb[0] = g.yb; rp.ye[0] = g.ye; }
    else {
        rp.nranges = ranges->n; rp.nfirst = ranges->nfirst;
        for (int r = 0; r < ranges->n; r++) { rp.yb[r] = ranges->yb[r]; rp.ye[r] = ranges->ye[r]; }
        rp.signal = ranges->signal; rp.arrivals = ranges->arrivals; rp.value = ranges->value;
    }
    if (rp.nranges < 1 || rp.nranges > 3 || rp.nfirst < 0 || rp.nfirst > 2 || rp.nfirst >= rp.nranges + (rp.nfirst ? 1 : 0)) return hipErrorInvalidValue;
    int total_first = 0;
    for (int r = 0; r < rp.nranges; r++) {
        cut_bands<S, TX>(rp.ye[r] - rp.yb[r], xtiles, lds, rp.band[r], rp.nbands[r]);
        const int nt = rp.ye[r] > rp.yb[r] ? xtiles * rp.nbands[r] * S : 0;
        if (r < rp.nfirst) { total_first += nt; rp.tiles_end[r] = total_first; } else rp.tiles_end[r] = nt;
    }
    const bool has_last = rp.nranges > rp.nfirst;                      // (a launch may consist of first ranges only)
    const int last_tiles = has_last ? rp.tiles_end[rp.nranges - 1] : 0;
    rp.first_tiles = total_first;
    unsigned blocks = 0;
    if (total_first > 0) { const dim3 gf = xcd_grid(total_first, 1, rp.xgroup_first); rp.first_blocks = (int)gf.x; blocks += gf.x; }
    else { rp.first_blocks = 0; rp.xgroup_first = 1; }
    if (!has_last) { rp.nranges += 1; rp.yb[rp.nranges - 1] = rp.ye[rp.nranges - 1] = 0; rp.band[rp.nranges - 1] = kAtrousMinBand; rp.nbands[rp.nranges - 1] = 1; rp.tiles_end[rp.nranges - 1] = 0; rp.xgroup = 1; }
    if (last_tiles > 0) { const dim3 gl = xcd_grid(last_tiles, S <= 2 ? 16 : (S == 16 ? 2 : 1), rp.xgroup); blocks += gl.x; }     // groups per XCD: A/B per step on one device (4K)
    else rp.xgroup = 1;
    if (!blocks) return hipSuccess;
    atrous_lds_kernel<ST, S, TX><<<dim3(blocks), dim3(TX * kRS), lds, s>>>(g, a, rp);
    return hipGetLastError();
}

// (The tile width is a template parameter because 64-column two-wave workgroups were tried for small launches — 1080p frames, strips —
// where they cut twice as many tiles in x and so allow bands twice as tall: parity-green and 3-5 % SLOWER at 1080p, on an 8K/8 strip
// and at 4K, profiles/r04_small_experiments.txt block 6.)
template <int ST, int S>
hipError_t launch_atrous_lds(const Geo& g, const AtrousArgs& a, hipStream_t s, const AtrousRanges* r) { return launch_atrous_lds_tx<ST, S, kTX>(g, a, s, r); }

template <int ST>
hipError_t launch_atrous_lds_step(const Geo& g, const AtrousArgs& a, hipStream_t s, const AtrousRanges* r = nullptr) {
    switch (a.step) {
        case 1: return launch_atrous_lds<ST, 1>(g, a, s, r);
        case 2: return launch_atrous_lds<ST, 2>(g, a, s, r);
        case 4: return launch_atrous_lds<ST, 4>(g, a, s, r);
        case 8: return launch_atrous_lds<ST, 8>(g, a, s, r);
        case 16: return launch_atrous_lds<ST, 16>(g, a, s, r);
        case 32: return launch_atrous_lds<ST, 32>(g, a, s, r);
        case 64: return launch_atrous_lds<ST, 64>(g, a, s, r);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace
}  // namespace svgf
