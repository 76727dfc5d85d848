// atrous_lds_instrumented.h — the LDS-streaming a-trous kernel as it stood at the end of round 2, WITH every measurement switch
// (tap pipeline depth, waves per SIMD, outputs per thread, streaming-only / arithmetic-only modes, cost probes, cache policy bits,
// in-kernel s_memtime stamps, getenv launch-geometry hooks) and the two alternative kernels measured against it
// (atrous_ws.h: specialised loader / compute waves; atrous_r4.h: four rows per step).  NOT part of the product: svgf_kernels.hip
// includes this file instead of svgf_atrous_lds.h only when the library is built with -DSVGF_DIAG (tools/stamps.py, tools/abn.sh,
// tools/sweep_launch.py build such twins under build/).  profiles/r02_atrous_ablations.txt was measured with it.
// The text lives in a namespace of its own (svgf::{anonymous}::r02): it brings its own copies of the staging helpers.
#pragma once
#ifdef SVGF_P2
#include "atrous_p2.h"        // two output pixels per thread: another kernel altogether
#else
#include "../../svgf_amd/csrc/svgf_device.h"

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>

// measurement switches of the a-trous kernel (tools/abn.sh builds twins of the library with other values)
#ifndef SVGF_COLOUR_LD_AUX
#define SVGF_COLOUR_LD_AUX 0        // cache policy bits of the a-trous colour loads / stores and G-buffer loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef SVGF_COLOUR_ST_AUX
#define SVGF_COLOUR_ST_AUX 0
#endif
#ifndef SVGF_GB_LD_AUX
#define SVGF_GB_LD_AUX 0
#endif
#ifndef SVGF_REVERSE_MASK
#define SVGF_REVERSE_MASK 1         // bit i set: the iteration with step 2^i walks the frame bottom-up.  Step 1 does: what the temporal launch wrote last is
                                    // still in the 256 MB Infinity Cache when it is read first (-4.5 % for that launch; for the later steps, whose
                                    // row residues sweep the frame several times, the order makes no difference or hurts: tools/abn.sh)
#endif

namespace svgf {
namespace {
namespace r02 {

// ------------------------------------------------------------------ a-trous (LDS streaming) ---
// Filter.cuh:527-624 re-designed for CDNA4.  For step S a pixel only ever reads pixels of its own row residue
// (y mod S), so a workgroup owns ONE residue of a band of rows and a TX-pixel-wide column block (TX = 128: 4 waves,
// 4 workgroups per CU; the 256-column form is kept for diagnostics), and streams down the band: a ring of
// kRing = kRS+4 decimated rows (tile + 2S halo columns each side) lives in LDS as fp32 records; every step the
// workgroup produces kRS = 2 vertically adjacent decimated rows from the ring (waves 0-1 row j, waves 2-3 row j+1:
// one output per thread), while the rows of the next two steps are already in flight from HBM into registers.
// Global loads are always full-width row segments (16 B per lane, coalesced) whatever the step; the y over-fetch is
// (band+4)/band and the x over-fetch (TX+4S)/TX instead of the 25x gather of a per-pixel kernel.
//
// LDS record per pixel (32 B in three planes): A = {r,g,b,variance} clamped (imageLoad :78-83), L = {luminance,
// depth (sky -> 1e30)}, N = {(nx,ny) as packed halfs, nz as float}.  The centre's ddepth is the only other per-pixel
// input: the thread that stages a pixel of its own column is the thread that later filters it, so ddepth rides in
// a two-register queue instead of LDS.
//
// Everything that is the same for all lanes of a wave — row offsets, ring slots, validity of a row — is kept in
// scalar registers: planes are addressed as buffer resources with a per-thread constant byte offset (voffset)
// plus a per-step scalar row offset (soffset), so staging a row costs no vector ALU at all.  With 4 waves
// sharing a SIMD every vector instruction outside the tap loop costs as much as inside it.
// Pixels outside the frame (or the strip) come back as all-zero texels from the buffer range check (a row
// outside the frame is loaded through a zero-length resource): depth 0 = sky sentinel and a zero normal give
// weight exactly 0, which is what skipping the tap (:579,584) does.
constexpr int kRS = 2;                   // decimated rows produced per step
constexpr int kRing = kRS + 4;
constexpr int kXcds = 8;                 // MI355X: 8 accelerator dies, workgroup id i is dispatched to XCD i % 8
constexpr int kRecBytes = 32;            // LDS bytes per staged pixel
#ifndef SVGF_NARROW_MAX_STEP
#define SVGF_NARROW_MAX_STEP 16
#endif
constexpr int kNarrowMaxStep = SVGF_NARROW_MAX_STEP;           // steps up to this one use 128-column workgroups
constexpr int kDefaultKR = 1;            // outputs per thread of the kernel the library launches (see atrous_lds_kernel)
constexpr unsigned kOob = 0xFFFFFF00u;   // byte offset no plane reaches (planes are < 4 GiB)

#ifdef SVGF_STAMPS
// In-kernel phase stamps (a diagnostic twin of the library only, tools/stamps.py; the product build has none of
// this, and the stamps' own waits slow that twin down: read its shares, not its run time).
// Every wave adds its sums to a slot of its own (blockIdx, wave): no atomics — 100 000 waves adding to the same sixteen words at
// their exits slowed the instrumented launch twelve-fold and stalled everybody's memory instructions.
constexpr int kStampSlots = 1 << 18;
__device__ unsigned long long g_stamp_log[(size_t)kStampSlots * 16];
__device__ __forceinline__ void stamp_add(int wave, int i, unsigned long long v) {
    const unsigned slot = (blockIdx.x * 8u + (unsigned)wave) & (unsigned)(kStampSlots - 1);
    g_stamp_log[(size_t)slot * 16 + i] += v;
}
// Where the waves really run: workgroups resident on the CU when a workgroup starts (sum in stamp 9), and a histogram of the
// SIMD each wave of a workgroup lands on.
#ifdef SVGF_STAMPS_PLACEMENT
__device__ unsigned g_cu_resident[4096];
__device__ unsigned g_simd_hist[8 * 4];
#endif
__device__ __forceinline__ unsigned hw_cu_key(unsigned& simd) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    simd = (hw >> 4) & 3u;
    return ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u)) & 4095u;
}
__device__ __forceinline__ unsigned stamp_enter(int wave, int lane) {        // -> the CU's key (for stamp_leave)
#ifdef SVGF_STAMPS_PLACEMENT        // (its atomics double the instrumented launch: a build of its own, SVGF_STAMPS_FLAGS=-DSVGF_STAMPS_PLACEMENT)
    unsigned simd;
    const unsigned key = hw_cu_key(simd);
    if (lane == 0) {
        atomicAdd(&g_simd_hist[(wave & 7) * 4 + simd], 1u);
        if (wave == 0) stamp_add(0, 9, atomicAdd(&g_cu_resident[key], 1u));
    }
    return key;
#else
    return 0u;
#endif
}
__device__ __forceinline__ void stamp_leave(int wave, int lane, unsigned key) {
#ifdef SVGF_STAMPS_PLACEMENT
    if (lane == 0 && wave == 0) atomicSub(&g_cu_resident[key], 1u);
#endif
}
#define SVGF_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_acc[i] += t_ - stamp_t; stamp_t = t_; } while (0)
#else
#define SVGF_STAMP(i) do { } while (0)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;   // explicitly in LDS (a volatile access through a generic pointer would be a flat load)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// One staged pixel as it comes off the planes: colour (16 B fp32 / 8 B fp16), {depth, ddepth} (ddepth only for
// pixels of the thread's own column: DZ), normal.
template <int ST, bool DZ> struct RawPx;
template <> struct RawPx<0, true> { u32x4 c; u32x2 zd; u32x2 n; };
template <> struct RawPx<1, true> { u32x2 c; u32x2 zd; u32x2 n; };
template <> struct RawPx<0, false> { u32x4 c; unsigned zd; u32x2 n; };
template <> struct RawPx<1, false> { u32x2 c; unsigned zd; u32x2 n; };

struct PlaneRsrc { __amdgpu_buffer_rsrc_t colour, motion, normal; };

// voff_c / voff_n: the lane's constant byte offsets into the colour(+motion) and normal planes (kOob for a
// column outside the frame); srow: the row's scalar element offset yl*W.
template <int ST, bool DZ>
__device__ __forceinline__ void raw_load(RawPx<ST, DZ>& r, const PlaneRsrc& rs, unsigned voff_c, unsigned voff_m, unsigned voff_n, int srow, unsigned n_shift) {
    constexpr int cb = ST == 0 ? 16 : 8;
    if constexpr (ST == 0) r.c = __builtin_amdgcn_raw_buffer_load_b128(rs.colour, voff_c, srow * cb, SVGF_COLOUR_LD_AUX);
    else r.c = __builtin_amdgcn_raw_buffer_load_b64(rs.colour, voff_c, srow * cb, SVGF_COLOUR_LD_AUX);
#ifndef SVGF_GUIDE_B128
#define SVGF_GUIDE_B128 0           // 1: the guide texel with one 16-byte load instead of two 8-byte loads (measured slower: tools/abn.sh)
#endif
    if (SVGF_GUIDE_B128 && n_shift == 4u) {
        // guide plane: ONE 16-byte texel {depth, ddepth, (nx,ny), (nz,-)} per pixel (voff_m is its offset)
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs.motion, voff_m, srow * 16, SVGF_GB_LD_AUX);
        if constexpr (DZ) r.zd = (u32x2){t.x, t.y}; else r.zd = t.x;
        r.n = (u32x2){t.z, t.w};
        return;
    }
#ifdef SVGF_DIAG_SKIP_MOTION
    if constexpr (DZ) r.zd = (u32x2){0x40a00000u, 0x3c23d70au}; else r.zd = 0x40a00000u;               // bandwidth probe only
#else
    if constexpr (DZ) r.zd = __builtin_amdgcn_raw_buffer_load_b64(rs.motion, voff_m, srow * 16, SVGF_GB_LD_AUX);      // {depth, ddepth}
    else r.zd = __builtin_amdgcn_raw_buffer_load_b32(rs.motion, voff_m, srow * 16, SVGF_GB_LD_AUX);                   // depth
#endif
#ifdef SVGF_DIAG_SKIP_NORMAL
    r.n = (u32x2){0x3c00u, 0xbc00u};      // bandwidth probe only (results are wrong)
#else
    r.n = __builtin_amdgcn_raw_buffer_load_b64(rs.normal, voff_n, srow << n_shift, SVGF_GB_LD_AUX);
#endif
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float med01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }   // = min(max(v,0),1) for non-NaN v

template <int ST, bool DZ>
__device__ __forceinline__ bool commit_px(const RawPx<ST, DZ>& r, f32x4* recA, f32x2* recL, f32x2* recN, int at, uint32_t ref01, uint32_t refz) {
    float4 c;
    if constexpr (ST == 0) c = make_float4(__uint_as_float(r.c.x), __uint_as_float(r.c.y), __uint_as_float(r.c.z), __uint_as_float(r.c.w));
    else { float2 lo = unpack_h2(r.c.x), hi = unpack_h2(r.c.y); c = make_float4(lo.x, lo.y, hi.x, hi.y); }
    c = make_float4(med01(c.x), med01(c.y), med01(c.z), med01(c.w));    // imageLoad, :586
    float z;
    if constexpr (DZ) z = __uint_as_float(r.zd.x); else z = __uint_as_float(r.zd);
    if (z == 0.0f) z = kSkyZ;                                           // GetDepth, :199-207
    recA[at] = (f32x4){c.x, c.y, c.z, c.w};
    recL[at] = (f32x2){lum_exact(c.x, c.y, c.z), z};
    recN[at] = (f32x2){__uint_as_float(r.n.x), unpack_h2(r.n.y).x};
    // a texel without depth (sky, or outside the frame) has weight 0 through the depth term whatever its normal
    return z != kSkyZ && (r.n.x != ref01 || (r.n.y & 0xffffu) != refz);
}

// log2 of the kernel weight K[|xx|]*K[|yy|] (:540,582), folded into the exponent
__device__ __forceinline__ constexpr float klog2(int axx, int ayy) {
    return (axx + ayy == 1) ? -0.5849624872207642f       // 1 * 2/3
         : (axx == 1 && ayy == 1) ? -1.1699249744415283f // 2/3 * 2/3
         : (axx + ayy == 2) ? -2.5849626064300537f       // 1 * 1/6
         : (axx + ayy == 3) ? -3.1699249744415283f       // 2/3 * 1/6
         : -5.169925212860107f;                          // 1/6 * 1/6
}
__device__ __forceinline__ constexpr int kernel_class(int axx, int ayy) {   // index of klog2's five values
    return (axx + ayy == 1) ? 0 : (axx == 1 && ayy == 1) ? 1 : (axx + ayy == 2) ? 2 : (axx + ayy == 3) ? 3 : 4;
}
__device__ __forceinline__ constexpr int len_class(int xx, int yy) {    // |(xx,yy)| in {1, sqrt2, 2, sqrt5, 2sqrt2}
    const int l2 = xx * xx + yy * yy;
    return l2 == 1 ? 0 : l2 == 2 ? 1 : l2 == 4 ? 2 : l2 == 5 ? 3 : 4;
}
// (nx,ny).(nx',ny') of two packed-half pairs: exact products, one rounding of their sum (v_dot2_f32_f16 with a
// zero addend; the builtin would pick the accumulating v_dot2c form and spend a v_mov on the zero).  hipcc does not
// look inside asm statements, so the three wait states a non-dot VALU needs before it may read (or overwrite) a
// dot result on gfx940+ are part of the statement; with 4 waves per SIMD they cost no VALU issue.
__device__ __forceinline__ float dot2_h2(uint32_t a, uint32_t b) {
    float d;
    asm("v_dot2_f32_f16 %0, %1, %2, 0\n\ts_nop 2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// MODE is a diagnostic knob (SVGF_DIAG builds only): 0 = the kernel, 1 = streaming only (no tap arithmetic),
// 2 = arithmetic only (no global prefetch / ring refill after the prologue).
//
// Every step produces kRS = 2 decimated rows of the workgroup's TX columns.  KR = outputs per thread:
//   KR = 1: 2*TX threads, thread t owns column t % TX and row group t / TX (waves 0-3 row j, waves 4-7 row j+1);
//           ~105 VGPRs, 4 waves per SIMD with the two workgroups per CU the ring's LDS footprint allows.
//   KR = 2: TX threads, each thread produces rows j and j+1 of its column and shares the 20 taps the two outputs
//           have in common (30 LDS record pairs per 2 outputs instead of 50): on CDNA4 an LDS read's data return
//           occupies the SIMD's register-file write path for ~16 cycles per ds_read_b128 and delays vector ALU
//           issue by as much (tools/ubench/tap_lds.hip), so LDS bytes per output are paid for like instructions.
#ifndef SVGF_NO_FASTPATH
#define SVGF_NO_FASTPATH 0          // 1: measure the kernel as it runs on geometry without planar regions (tools/ab.sh)
#endif
#ifndef SVGF_TAP_DEPTH
#define SVGF_TAP_DEPTH 3            // > 0: taps as one rolling pipeline, LDS reads this many taps ahead (see tap_roll); 0: a ring row at a time
#endif
#ifndef SVGF_MIN_WAVES
#define SVGF_MIN_WAVES 0            // != 0: waves per SIMD the register allocation is asked to leave room for, every step (KR = 1); 0: per step below
#endif
#ifndef SVGF_WAVES_S1
#define SVGF_WAVES_S1 5             // steps 1 and 2
#endif
#ifndef SVGF_WAVES_S4
#define SVGF_WAVES_S4 5             // steps 4 and 8
#endif
#ifndef SVGF_WAVES_S16
#define SVGF_WAVES_S16 4            // step 16: the ring (37 KB) allows four workgroups per CU anyway
#endif
// Resident waves per SIMD the kernel of step S is compiled for: more workgroups per CU keep more loads in flight while others
// are in their tap phase (5 per CU: -4.5 % per launch at S <= 8, tools/abn.sh W5).  The tap pipeline is one tap shorter per
// step of occupancy beyond 5 (registers).
constexpr int cfg_waves(int S) { return SVGF_MIN_WAVES ? SVGF_MIN_WAVES : (S <= 2 ? SVGF_WAVES_S1 : S <= 8 ? SVGF_WAVES_S4 : SVGF_WAVES_S16); }
constexpr int cfg_tap_depth(int S) { return (SVGF_TAP_DEPTH > 0 && cfg_waves(S) >= 6) ? 1 : SVGF_TAP_DEPTH; }
#ifndef SVGF_KR2_WAVES
#define SVGF_KR2_WAVES 2            // the same for KR = 2
#endif
#ifndef SVGF_PREFETCH_STEPS
#define SVGF_PREFETCH_STEPS 1       // ring rows are requested this many steps before the step that needs them (1 = at the start of the step whose
                                    // end commits them: 15 registers per step of depth, and depth 2 or 3 measured no faster: tools/abn.sh PF1..PF3)
#endif
#ifndef SVGF_PROLOGUE_ALL
#define SVGF_PROLOGUE_ALL 0         // 1: the six ring rows of a workgroup's prologue requested at once
#endif
#ifndef SVGF_FORCE_MODE
#define SVGF_FORCE_MODE 0           // diagnostic builds: 1 = streaming only, 2 = arithmetic only (see MODE), for the kernels the library launches
#endif
#ifndef SVGF_WAVE_TILE
#define SVGF_WAVE_TILE 0            // 1: single-wave workgroups (64 columns, KR = 2): no barrier, no sibling wave to wait for
#endif
template <int ST, int S, int TX, int KR, int MODE = 0>
__global__ __launch_bounds__(TX * (kRS / KR), KR == 1 ? cfg_waves(S) : SVGF_KR2_WAVES) void atrous_lds_kernel(Geo g, AtrousArgs a, int band_rows, int nbands, int xgroup, int xrot, int band_fastest) {
    constexpr int WL = TX + 4 * S;                 // staged columns per ring row
    constexpr int CB = ST == 0 ? 16 : 8;           // bytes per colour texel
    constexpr int NH = 4 * S;                      // halo pixels per ring row: all staged by wave 0 of the row group (lanes 0..NH-1);
                                                   // spread over the waves, every wave paid the halo's ~20 VALU + 3 loads for a few lanes
    constexpr int NR = KR + 4;                     // ring rows a thread reads
    static_assert(NH >= 1 && NH <= 64, "halo does not fit one wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* recA = (f32x4*)smem;
    f32x2* recL = (f32x2*)(recA + kRing * WL);     // 8-byte records, contiguous: conflict-free ds_read_b64 (64 banks)
    f32x2* recN = recL + kRing * WL;
    // Uniform-normal fast path: on planar geometry every texel of the ring carries the same normal bits; then
    // n.n' is the centre's own |n|^2 for every tap and the dot product, its log2 and an FMA (22 of a tap's ~59 VALU
    // cycles) leave the tap loop — with bit-identical results.  nflag[slot][wave] = "a texel of this ring row staged
    // by this wave differs from the workgroup's reference normal" (depth-0 texels do not count: their weight is 0).
    uint32_t* nflag = (uint32_t*)(recN + kRing * WL);              // [kRing][8]
    uint32_t* nref = nflag + kRing * 8;                            // {(nx,ny) bits, nz bits}

#ifdef SVGF_STAMPS
    unsigned long long stamp_entry, stamp_real0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_entry), "=s"(stamp_real0) :: "memory");
#endif
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int col = t % TX;
    const int rg = __builtin_amdgcn_readfirstlane(t / TX);          // row group: wave-uniform -> scalar
    const int wig = __builtin_amdgcn_readfirstlane((t % TX) >> 6);  // wave index inside its row group
    // XCD-aware tile order.  The dispatcher deals consecutive workgroup ids to the 8 XCDs in turn, and each XCD has
    // its own L2: with a plain (x, y) grid the two tiles that share a 2S-column halo, or two bands that share four
    // ring rows, always sit on different XCDs and every halo texel comes from memory twice.  Here tile order is
    // v = (residue, band, x tile) with x fastest, cut into groups of `xgroup` consecutive tiles, and group k goes to
    // XCD k % 8: neighbours inside a group run on one XCD at about the same time and share their halos in its L2.
    const int xtiles = (g.W + TX - 1) / TX;
    const int ntiles = xtiles * nbands * S;
    const int wid = blockIdx.x >> 3;               // index among the workgroups of this XCD
    const int round = wid / xgroup;                // the XCD's round-th group; rotated so that an XCD's groups come from different parts of the frame
    int v = (round * kXcds + ((blockIdx.x + xrot * round) & (kXcds - 1))) * xgroup + wid % xgroup;
    if (v >= ntiles) return;                       // padding of the last groups
    if ((SVGF_REVERSE_MASK / S) & 1) v = ntiles - 1 - v;
    // band_fastest: tile order (residue, x tile, band) instead — an XCD's consecutive workgroups walk down one column
    // of tiles (every band halo shared, and each XCD's share of the frame is a set of vertical strips)
    const int x0 = (band_fastest ? (v / nbands) % xtiles : v % xtiles) * TX;
    const int band = band_fastest ? v % nbands : (v / xtiles) % nbands;
    const int rv = v / (xtiles * nbands);          // row residue (relative to g.yb) this workgroup owns
    const int nrows = g.ye - g.yb;
    const int nj = (nrows - rv + S - 1) / S;       // decimated rows of this residue
    const int j0 = band * band_rows;
    if (j0 >= nj) return;
    const int j1 = min(nj, j0 + band_rows);
    const int ybase = g.yb + rv;                   // global row of decimated index j: ybase + S*j

    // per-lane constants
    const int gx = x0 + col;                       // own column
    const int oli = col + 2 * S;                   // its LDS column
    const bool halo_wave = wig == 0;               // scalar
    const bool has_halo = halo_wave && lane < NH;  // this lane also stages one halo pixel per row of its row group
    const int hh = lane;                           // 0 .. 4S-1
    const int hx = (hh < 2 * S) ? x0 - 2 * S + hh : x0 + TX + hh - 2 * S;
    const int hli = (hh < 2 * S) ? hh : TX + hh;
    const bool own_ok = gx < g.W, halo_ok = has_halo && hx >= 0 && hx < g.W;
    // Depth / normal source: the G-buffer's motion plane (16-B texels, {depth, ddepth} at +8) and normal plane (8-B texels), or
    // the frame's guide plane (16-B texels: {depth, ddepth} at +0, normal at +8): the same two loads, 16 instead of 24 bytes of
    // lines per pixel.  Everything here is a scalar select.
#ifdef SVGF_NO_GUIDE_CODE
    constexpr bool guided = false;                                   // measurement: the kernel as it was before the guide plane
    constexpr unsigned m_off = 8u, n_off = 0u, n_shift = 3u;
#else
    const bool guided = a.guide != nullptr;
    const unsigned m_off = guided ? 0u : 8u, n_off = guided ? 8u : 0u, n_shift = guided ? 4u : 3u;
#endif
    const unsigned vo_c = own_ok ? (unsigned)gx * CB : kOob, vo_m = own_ok ? (unsigned)gx * 16u + m_off : kOob, vo_n = own_ok ? ((unsigned)gx << n_shift) + n_off : kOob;
    const unsigned vh_c = halo_ok ? (unsigned)hx * CB : kOob, vh_m = halo_ok ? (unsigned)hx * 16u + m_off : kOob, vh_n = halo_ok ? ((unsigned)hx << n_shift) + n_off : kOob;

    // Buffer resources are built where they are used (base pointer + a num_records word chosen by a scalar select) instead
    // of being kept in 32 SGPRs for the whole kernel; a row outside the frame gets num_records = 0: every load returns 0.
    const unsigned npx = (unsigned)g.rows * (unsigned)g.W;
    auto plane_rsrc = [&](bool rok) __attribute__((always_inline)) {
        PlaneRsrc r;
        r.colour = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, rok ? (int)(npx * CB) : 0, 0x00020000);
        r.motion = __builtin_amdgcn_make_buffer_rsrc(guided ? (void*)a.guide : (void*)a.motion, 0, rok ? (int)(npx * 16u) : 0, 0x00020000);
        r.normal = __builtin_amdgcn_make_buffer_rsrc(guided ? (void*)a.guide : (void*)a.normal, 0, rok ? (int)(npx << n_shift) : 0, 0x00020000);
        return r;
    };

    // A thread's share of one staged step: KR rows (jn + rg*KR + k): own pixel, and a halo pixel on lanes < NH
    typedef RawPx<ST, true> OwnPx;
    typedef RawPx<ST, false> HaloPx;
    struct Staged { OwnPx o[KR]; HaloPx h[KR]; };
    auto fetch = [&](int jn, Staged& st) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const int y = ybase + S * (jn + rg * KR + k), yl = y - g.y0;                    // scalar
            const bool rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
            const int srow = rok ? yl * g.W : 0;
            const PlaneRsrc rs = plane_rsrc(rok);
            raw_load<ST, true>(st.o[k], rs, vo_c, vo_m, vo_n, srow, n_shift);
            if (halo_wave) raw_load<ST, false>(st.h[k], rs, vh_c, vh_m, vh_n, srow, n_shift);
        }
    };
    uint32_t ref01 = 0, refz = 0;
    auto commit = [&](int sl, const Staged& st) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KR; k++) {
            int so = sl + rg * KR + k; so = so >= kRing ? so - kRing : so;                   // scalar
            bool differs = commit_px<ST, true>(st.o[k], recA, recL, recN, so * WL + oli, ref01, refz);
            if (halo_wave) { if (has_halo) differs = commit_px<ST, false>(st.h[k], recA, recL, recN, so * WL + hli, ref01, refz) || differs; }
            const bool wave_differs = __ballot(differs) != 0ull;
            if (lane == 0) nflag[so * 8 + wig] = wave_differs ? 1u : 0u;                     // a ring slot is always staged by the same waves
        }
    };

    // ddepth of this thread's next two centres per output row (rows j0+rg*KR+k and two rows further).  A staged row
    // becomes a centre two steps after it is committed; its ddepth is taken over at commit time (never at fetch
    // time: that would wait for the prefetch it was issued with).
    float dq0[KR], dq1[KR];
    // prologue: two ring rows at a time (requesting all six at once measured the same: the launch is one resident
    // round, so the first memory latency is paid once per kernel either way).  Rows j0, j0+1 (always inside the
    // frame) go first: thread 0's pixel of row j0 is the workgroup's reference normal.
#ifdef SVGF_STAGGER
    // measurement: de-phase the workgroups that start together on a CU (consecutive ids of an XCD), in units of 64*SVGF_STAGGER cycles
    for (int q = (int)((blockIdx.x >> 3) & 3u) * SVGF_STAGGER; q > 0; q--) __builtin_amdgcn_s_sleep(1);
#endif
#ifdef SVGF_STAMPS
    const unsigned stamp_key = stamp_enter(t >> 6, lane);
#endif
    if (t < kRing * 8) nflag[t] = 0u;
#if SVGF_PROLOGUE_ALL
    {
        // the whole ring requested at once: ONE round of memory latency per workgroup instead of three (a slot runs four
        // workgroups per launch; the registers of the tap loop are free here)
        Staged st0, st1, st2;
        fetch(j0, st0);
        fetch(j0 - 2, st1);
        fetch(j0 + 2, st2);
        if (t == 0) { nref[0] = st0.o[0].n.x; nref[1] = st0.o[0].n.y & 0xffffu; }
        __syncthreads();
        ref01 = nref[0]; refz = nref[1];
        commit(2, st0);
        commit(0, st1);
        commit(4, st2);
#pragma unroll
        for (int k = 0; k < KR; k++) { dq0[k] = __uint_as_float(st0.o[k].zd.y); dq1[k] = __uint_as_float(st2.o[k].zd.y); }
    }
#else
#pragma unroll 1
    for (int rr = 0; rr < kRing; rr += kRS) {
        const int r = rr == 0 ? 2 : (rr == 2 ? 0 : rr);
        Staged st;
        fetch(j0 - 2 + r, st);
        if (rr == 0) {
            if (t == 0) { nref[0] = st.o[0].n.x; nref[1] = st.o[0].n.y & 0xffffu; }
            __syncthreads();
            ref01 = nref[0]; refz = nref[1];
        }
        commit(r, st);
#pragma unroll
        for (int k = 0; k < KR; k++) {
            if (r == 2) dq0[k] = __uint_as_float(st.o[k].zd.y);
            if (r == 4) dq1[k] = __uint_as_float(st.o[k].zd.y);
        }
    }
#endif
    __syncthreads();

    const float phi_n = a.phi_normal;              // != 0 (launcher)
    int slot0 = 0;
    // a workgroup of ONE wave needs no barrier: the LDS operations of a wave execute in order
    auto wg_barrier = [&]() __attribute__((always_inline)) { if constexpr (TX * (kRS / KR) > 64) lds_barrier(); else asm volatile("" ::: "memory"); };
#ifdef SVGF_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t;
    unsigned long long stamp_cnt[3] = {0, 0, 0};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t) :: "memory");
    const unsigned long long stamp_first = stamp_t;
#endif

    // One step: produce decimated rows j and j+1 from the ring.  `cs` holds the rows the NEXT step needs (fetched
    // during the previous step, landed by now; committed at the end of this one), `fs` receives the rows of the
    // step after that — in flight during this step's arithmetic.
    auto step = [&](int j, Staged& cs, Staged& fs) __attribute__((always_inline)) {
        const bool more = MODE != 2 && (j + kRS) < j1;
        const bool more2 = MODE != 2 && (j + SVGF_PREFETCH_STEPS * kRS) < j1;
        if (more2) fetch(j + SVGF_PREFETCH_STEPS * kRS + 2, fs);
        SVGF_STAMP(0);                             // fetch issue

        // this thread's centres are ring rows 2+rg*KR+k, its taps ring rows rg*KR .. rg*KR+KR+3; columns oli-2S .. oli+2S
        int rowbase[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) { int sl = slot0 + rg * KR + r; sl = sl >= kRing ? sl - kRing : sl; rowbase[r] = sl * WL + col; }   // scalar + lane constant

        f32x4 cA[KR];
        f32x2 lzc[KR], srg[KR], sbv[KR];
        float ncz[KR], il[KR], iz[KR][5], sw[KR];
        uint32_t nc01[KR];
        bool any_surface = false;
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const f32x4 A = recA[rowbase[2 + k] + 2 * S];
            const f32x2 L = recL[rowbase[2 + k] + 2 * S], N = recN[rowbase[2 + k] + 2 * S];
            const f32x4 B = {L.x, L.y, N.x, N.y};
            cA[k] = A;
            const float cdz = B.y == kSkyZ ? 0.0f : dq0[k];                                  // GetDepth: sky -> ddepth 0
            lzc[k] = (f32x2){B.x, B.y};                                                      // centre luminance, depth
            ncz[k] = B.w;
            nc01[k] = __float_as_uint(B.z);
            const float phi_l = a.phi_colour * sqrtf(fmaxf(0.0f, 1e-10f + A.w));             // :562
            il[k] = fminf(hw_rcp(phi_l), 1e30f) * kLog2e;
            const float izb = hw_rcp(fmaxf(cdz, 1e-6f) * (float)S) * kLog2e;                 // :563
            iz[k][0] = izb; iz[k][1] = izb * 0.70710678118654752f; iz[k][2] = izb * 0.5f;
            iz[k][3] = izb * 0.44721359549995794f; iz[k][4] = izb * 0.35355339059327376f;
            // accumulators, packed by channel pairs: (r,g) and (b,variance) advance with one v_pk_fma_f32 each
            sw[k] = 1.0f;                                                                     // :567
            srg[k] = (f32x2){A.x, A.y}; sbv[k] = (f32x2){A.z, A.w};                           // :568
            any_surface = any_surface || B.y != kSkyZ;
        }

        // a wave whose centres are all sky (a band of cleared texels) has nothing to filter (:554-558)
        const bool wave_has_surface = __ballot(any_surface) != 0ull;
        // every surface texel of the ring has the reference normal -> n.n' is each centre's own |n|^2
        const bool uniform_normals = !a.no_fastpath && __ballot(lane < kRing * 8 && nflag[lane < kRing * 8 ? lane : 0] != 0u) == 0ull;
        // One ring row at a time (5 taps = 10 x ds_read_b128 in flight; KR = 2 reads the next row before it
        // consumes the current one).  The empty asm statements pin that order: left alone, instruction selection
        // sinks all arithmetic below all LDS reads of the unrolled loop (256 VGPRs + scratch spills).
        f32x4 tA[KR][5];
        f32x2 tL[KR][5], tN[KR][5];
        // uni: the normal records are not read at all when the ring's normals are uniform
        auto load_row = [&](int r, int buf, bool uni) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 5; c++) {
                if (KR == 1 && r == 2 && c == 2) continue;                                        // the centre itself: already in registers
                tA[buf][c] = recA[rowbase[r] + c * S];
                // volatile: keeps these as single ds_read_b64 (2 LDS cycles each); merged into ds_read2_b64 they take 8
                tL[buf][c] = ((const volatile lds_f32x2*)recL)[rowbase[r] + c * S];
                if (!uni) tN[buf][c] = ((const volatile lds_f32x2*)recN)[rowbase[r] + c * S];
            }
        };
        auto tap_rows = [&](auto uni_tag) __attribute__((always_inline)) {
            constexpr bool UNI = decltype(uni_tag)::value;
            // UNI: exponent of the normal term + kernel weight, per kernel-weight class, from the centre's own |n|^2 (the
            // same expression the general path evaluates per tap, so the results are bit-identical)
            float ebase[KR][5];
            if constexpr (UNI) {
#pragma unroll
                for (int k = 0; k < KR; k++) {
                    const float lg = hw_log2(clamp01(fmaf(ncz[k], ncz[k], dot2_h2(nc01[k], nc01[k]))));
                    ebase[k][0] = fmaf(lg, phi_n, klog2(0, 1)); ebase[k][1] = fmaf(lg, phi_n, klog2(1, 1)); ebase[k][2] = fmaf(lg, phi_n, klog2(0, 2));
                    ebase[k][3] = fmaf(lg, phi_n, klog2(1, 2)); ebase[k][4] = fmaf(lg, phi_n, klog2(2, 2));
                }
            }
            constexpr bool kDouble = true;        // KR = 2: double-buffer the ring rows (measured: 6 % faster than not)
            if (KR == 2 && kDouble) load_row(0, 0, UNI);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int buf = (KR == 2 && kDouble) ? (r & 1) : 0;
                if (KR == 2 && kDouble) { if (r + 1 < NR) load_row(r + 1, buf ^ 1, UNI); } else load_row(r, 0, UNI);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int xx = -2; xx <= 2; xx++) {
                    const f32x4 A = tA[buf][xx + 2];
                    const f32x2 L = tL[buf][xx + 2];
#pragma unroll
                    for (int k = 0; k < KR; k++) {
                        const int yy = r - 2 - k;
                        if (yy < -2 || yy > 2 || (xx == 0 && yy == 0)) continue;             // compile time; centre: weight 1, already in
                        const int axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                        const f32x2 dlz = L - lzc[k];
                        float e;
                        if constexpr (UNI) {
                            e = ebase[k][kernel_class(axx, ayy)];
                        } else {
                            const f32x2 N = tN[buf][xx + 2];
                            const float d = clamp01(fmaf(N.y, ncz[k], dot2_h2(__float_as_uint(N.x), nc01[k])));
                            e = fmaf(hw_log2(d), phi_n, klog2(axx, ayy));
                        }
                        e = fmaf(-fabsf(dlz.x), il[k], e);
                        e = fmaf(-fabsf(dlz.y), iz[k][len_class(xx, yy)], e);
                        const float w = hw_exp2(e);
                        const f32x2 ww = {w, w * w};                                          // weights of (b, variance): :604-608
                        sw[k] += w;                                                           // :607
                        srg[k] = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, srg[k]);
                        sbv[k] = __builtin_elementwise_fma(ww, (f32x2){A.z, A.w}, sbv[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < KR; k++) asm volatile("" : "+v"(sw[k]), "+v"(srg[k]), "+v"(sbv[k]) :: "memory");
            }
        };
        // KR = 1, SVGF_TAP_DEPTH = D > 0: the 24 taps as ONE rolling software pipeline — the LDS reads of tap t+D are issued
        // before tap t is consumed, across row boundaries (row-at-a-time, every ring row started with no read in flight:
        // five LDS round trips per step exposed to the wave), and only D+1 taps' records are live instead of a row's five.
        auto tap_roll = [&](auto uni_tag) __attribute__((always_inline)) {
            constexpr bool UNI = decltype(uni_tag)::value;
            constexpr int D = cfg_tap_depth(S) > 0 ? cfg_tap_depth(S) : 1;
            constexpr int NT = 5 * NR;                       // records of the thread's NR ring rows, row-major
            float ebase[KR][5];
            if constexpr (UNI) {
#pragma unroll
                for (int k = 0; k < KR; k++) {
                    const float lg = hw_log2(clamp01(fmaf(ncz[k], ncz[k], dot2_h2(nc01[k], nc01[k]))));
                    ebase[k][0] = fmaf(lg, phi_n, klog2(0, 1)); ebase[k][1] = fmaf(lg, phi_n, klog2(1, 1)); ebase[k][2] = fmaf(lg, phi_n, klog2(0, 2));
                    ebase[k][3] = fmaf(lg, phi_n, klog2(1, 2)); ebase[k][4] = fmaf(lg, phi_n, klog2(2, 2));
                }
            }
            f32x4 qA[NT];
            f32x2 qL[NT], qN[NT];
            auto skip = [](int t) constexpr { return KR == 1 && t == 12; };                       // KR = 1: the centre itself is no tap
            auto issue = [&](int t) __attribute__((always_inline)) {
                if (skip(t)) return;
                const int r = t / 5, c = t % 5;
#ifdef SVGF_DIAG_NO_LDS_TAPS
                qA[t] = cA[0] * (float)(t + 1); qL[t] = lzc[0] * (float)(t + 2); qN[t] = (f32x2){__uint_as_float(nc01[0]), ncz[0]};      // cost probe only
                (void)r; (void)c;
#else
                qA[t] = recA[rowbase[r] + c * S];
                qL[t] = ((const volatile lds_f32x2*)recL)[rowbase[r] + c * S];
                if (!UNI) qN[t] = ((const volatile lds_f32x2*)recN)[rowbase[r] + c * S];
#endif
            };
#pragma unroll
            for (int t = 0; t < D; t++) issue(t);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t + D < NT) issue(t + D);
                asm volatile("" ::: "memory");
                if (skip(t)) continue;
                const int r = t / 5, xx = t % 5 - 2;
                const f32x4 A = qA[t];
                const f32x2 L = qL[t];
#pragma unroll
                for (int k = 0; k < KR; k++) {
                    const int yy = r - 2 - k;
                    if (yy < -2 || yy > 2 || (xx == 0 && yy == 0)) continue;                     // compile time
                    const int axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                    const f32x2 dlz = L - lzc[k];
                    float e;
                    if constexpr (UNI) {
                        e = ebase[k][kernel_class(axx, ayy)];
                    } else {
                        const f32x2 N = qN[t];
                        const float d = clamp01(fmaf(N.y, ncz[k], dot2_h2(__float_as_uint(N.x), nc01[k])));
                        e = fmaf(hw_log2(d), phi_n, klog2(axx, ayy));
                    }
                    e = fmaf(-fabsf(dlz.x), il[k], e);
                    e = fmaf(-fabsf(dlz.y), iz[k][len_class(xx, yy)], e);
#ifdef SVGF_DIAG_NO_EXP
                    const float w = e * 0.001f;                                               // cost probe only
#else
                    const float w = hw_exp2(e);
#endif
                    const f32x2 ww = {w, w * w};
                    sw[k] += w;
                    srg[k] = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, srg[k]);
                    sbv[k] = __builtin_elementwise_fma(ww, (f32x2){A.z, A.w}, sbv[k]);
                }
#pragma unroll
                for (int k = 0; k < KR; k++) asm volatile("" : "+v"(sw[k]), "+v"(srg[k]), "+v"(sbv[k]) :: "memory");
            }
        };
        if (MODE != 1 && wave_has_surface) {
#ifdef SVGF_DIAG_TAPS_TWICE
            if (uniform_normals && !SVGF_NO_FASTPATH) tap_roll(std::true_type{}); else tap_roll(std::false_type{});      // cost probe only (results are wrong)
#endif
            if constexpr (SVGF_TAP_DEPTH > 0) {
                if (uniform_normals && !SVGF_NO_FASTPATH) tap_roll(std::true_type{}); else tap_roll(std::false_type{});
            } else {
                if (uniform_normals && !SVGF_NO_FASTPATH) tap_rows(std::true_type{}); else tap_rows(std::false_type{});
            }
        }
#ifdef SVGF_STAMPS
        stamp_cnt[0]++; if (uniform_normals) stamp_cnt[1]++; if (!wave_has_surface) stamp_cnt[2]++;
#endif
        SVGF_STAMP(1);                             // centre setup + tap loop

        // Output values now, their stores AFTER the ring refill: hipcc's vmcnt bookkeeping cannot tell that the rows
        // committed below were fetched long before this step's stores, so stores issued first would be waited for.
        float4 o[KR];
#pragma unroll
        for (int k = 0; k < KR; k++) {
            if (lzc[k].y == kSkyZ) {
                o[k] = make_float4(cA[k].x, cA[k].y, cA[k].z, cA[k].w);                            // :554-558
            } else {
                const float inv = hw_rcp(sw[k]);                                                   // sw >= 1
                o[k] = make_float4(srg[k].x * inv, srg[k].y * inv, sbv[k].x * inv, sbv[k].y * (inv * inv));   // :615
            }
        }
        SVGF_STAMP(2);                             // epilogue
        if (more) {
            // Raw barriers: __syncthreads() would also wait for vmcnt(0), i.e. for the prefetch issued at the
            // start of this step — exactly the latency the two-step prefetch exists to hide.  Only this wave's
            // LDS reads/writes have to be done.
            wg_barrier();                          // every wave is done reading the kRS oldest ring rows
            SVGF_STAMP(3);                         // barrier 1
            commit(slot0, cs);
#pragma unroll
            for (int k = 0; k < KR; k++) { dq0[k] = dq1[k]; dq1[k] = __uint_as_float(cs.o[k].zd.y); }   // rows j+4+..: the centres two steps on
            slot0 += kRS; if (slot0 >= kRing) slot0 -= kRing;
            SVGF_STAMP(4);                         // wait for the staged rows + convert + LDS writes
            wg_barrier();
            SVGF_STAMP(5);                         // barrier 2
        }
#pragma unroll
        for (int k = 0; k < KR; k++) {
            if (j + rg * KR + k < j1) {                                                            // scalar
                const int srow = (ybase + S * (j + rg * KR + k) - g.y0) * g.W;
                const bool sky = lzc[k].y == kSkyZ;
                const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(npx * CB), 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_fb = __builtin_amdgcn_make_buffer_rsrc(a.feedback ? a.feedback : a.out, 0, a.feedback ? (int)(npx * CB) : 0, 0x00020000);
                // columns outside the frame carry the out-of-range offset: the store is dropped by the range check
#ifdef SVGF_DIAG_SKIP_STORE
                if (o[k].x != 12345.678f) continue;                                                // cost probe only: (almost) never stores
#endif
                if constexpr (ST == 0) {
                    const u32x4 raw = {__float_as_uint(o[k].x), __float_as_uint(o[k].y), __float_as_uint(o[k].z), __float_as_uint(o[k].w)};
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, vo_c, srow * CB, SVGF_COLOUR_ST_AUX);                   // :618
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_fb, sky ? kOob : vo_c, srow * CB, 0);      // :619-622 (not for sky)
                } else {
                    const u32x2 raw = {pack_h2(o[k].x, o[k].y), pack_h2(o[k].z, o[k].w)};
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_out, vo_c, srow * CB, SVGF_COLOUR_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_fb, sky ? kOob : vo_c, srow * CB, 0);
                }
            }
        }
    };

    // two register sets take turns as "commit next" / "fetch for the step after" (no copies between steps)
    // SVGF_PREFETCH_STEPS register sets take turns as "commit next" / "fetch for the step after ..." (no copies between steps):
    // rows requested at the start of a step are committed at the end of the step SVGF_PREFETCH_STEPS - 1 steps later
    constexpr int PD = SVGF_PREFETCH_STEPS;
    Staged q[PD];
#pragma unroll
    for (int d = 1; d < PD; d++) if (MODE != 2 && j0 + d * kRS < j1) fetch(j0 + d * kRS + 2, q[d - 1]);
    for (int j = j0; j < j1; j += PD * kRS) {
#pragma unroll
        for (int u = 0; u < PD; u++) if (j + u * kRS < j1) step(j + u * kRS, q[u], q[(u + PD - 1) % PD]);
    }
#ifdef SVGF_STAMPS
    if ((t & 63) == 0) {
        const int w_ = t >> 6;
        for (int i = 0; i < 6; i++) stamp_add(w_, i, stamp_acc[i]);
        stamp_add(w_, 6, stamp_first - stamp_entry);          // prologue: entry -> first step
        stamp_add(w_, 7, stamp_t - stamp_entry);              // lifetime of the wave
        unsigned long long stamp_real1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_real1) :: "memory");
        stamp_add(w_, 13, stamp_real0); stamp_add(w_, 14, stamp_real1);   // absolute 100 MHz times: the launch's occupancy over time (svgf_diag_stamp_log, ONE launch)
        stamp_add(w_, 8, 1ull);
        stamp_add(w_, 10, stamp_cnt[0]); stamp_add(w_, 11, stamp_cnt[1]); stamp_add(w_, 12, stamp_cnt[2]);
        stamp_leave(w_, 0, stamp_key);
    }
#endif
}

#ifdef SVGF_DIAG
inline int diag_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#endif

// Per-device launch facts, cached without a lock: contexts on different devices (or host threads) may launch concurrently.
constexpr int kMaxDevices = 64;
inline int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 ? dev : 0;
}
inline int num_cus() {
    static std::atomic<int> cus[kMaxDevices];
    const int dev = current_device();
    int n = dev < kMaxDevices ? cus[dev].load(std::memory_order_relaxed) : 0;
    if (n <= 0) {
        n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (n <= 0) n = 256;
        if (dev < kMaxDevices) cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, device) pair: set once per device.  Setting it twice is
// harmless, so a relaxed flag per device is enough for concurrent first launches.
template <typename K>
hipError_t allow_dynamic_lds(K kernel, size_t bytes, std::atomic<unsigned long long>& done) {
    const int dev = current_device();
    const unsigned long long bit = dev < kMaxDevices ? 1ull << dev : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}

template <int ST, int S, int TX, int KR, int MODE = 0>
hipError_t launch_atrous_lds(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    constexpr int WL = TX + 4 * S;
    constexpr size_t lds = (size_t)kRing * WL * kRecBytes + (kRing * 8 + 2) * sizeof(uint32_t);
    constexpr int threads = TX * (kRS / KR);
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(atrous_lds_kernel<ST, S, TX, KR, MODE>, lds, attr_done); e != hipSuccess) return e;
    // One round of workgroups: bands are sized so that (x tiles) x (S residues) x (bands) fills the resident
    // slots of the chip once (LDS: 160 KiB per CU; registers: 4 / 2 waves per SIMD) instead of leaving a partial round.
    constexpr int per_cu_lds = (int)((160 * 1024) / lds), per_cu_waves = (KR == 1 ? 4 * cfg_waves(S) : 4 * SVGF_KR2_WAVES) / (threads / 64);
    constexpr int per_cu = per_cu_lds < per_cu_waves ? per_cu_lds : per_cu_waves;
    const int nrows = g.ye - g.yb;
    const int njmax = (nrows + S - 1) / S;
    const int xtiles = (g.W + TX - 1) / TX;
    // 128-column workgroups: four times as many workgroups as resident slots, so that workgroups that take a fast
    // path (all sky, uniform normals) make room for others instead of idling until the slowest one of a single round
    // finishes (A/B on one device: 2x -3..5 %, 4x another -1.5 %, 6x worse; no gain for the 256-column kernels)
#ifndef SVGF_OVERSUB
#define SVGF_OVERSUB 4
#endif
#ifndef SVGF_BAND_SLOTS_PER_CU
#define SVGF_BAND_SLOTS_PER_CU per_cu
#endif
    int slots = (SVGF_BAND_SLOTS_PER_CU) * num_cus() * (TX <= 128 ? SVGF_OVERSUB : 1);
#ifdef SVGF_DIAG
    slots = diag_env("SVGF_ATROUS_SLOTS", slots);
#endif
    int nbands = slots / (xtiles * S);
    if (nbands < 1) nbands = 1;
    int band = (njmax + nbands - 1) / nbands;
#ifndef SVGF_MIN_BAND
#define SVGF_MIN_BAND 8
#endif
    int min_band = SVGF_MIN_BAND;
#ifdef SVGF_DIAG
    if (const int only = diag_env("SVGF_ATROUS_ONLY_STEP", 0); only == 0 || only == S) {   // tune one step at a time
        slots = diag_env("SVGF_ATROUS_SLOTS_S", slots);
        nbands = slots / (xtiles * S);
        if (nbands < 1) nbands = 1;
        band = (njmax + nbands - 1) / nbands;
        min_band = diag_env("SVGF_ATROUS_MIN_BAND", min_band);
    }
#endif
    if (band < min_band) band = min_band;
    band = (band + kRS - 1) / kRS * kRS;
    nbands = (njmax + band - 1) / band;
    // m groups per XCD, 8 m groups in all (so that every XCD gets the same number of tiles)
    int xm = S <= 2 ? 16 : (S == 16 ? 2 : 1);              // A/B per step on one device (4K): tools/xgroup.sh
#ifdef SVGF_DIAG
    xm = diag_env("SVGF_ATROUS_XM", xm);
    if (xm < 1) xm = 1;
#endif
    int xgroup = (xtiles * nbands * S + kXcds * xm - 1) / (kXcds * xm);
#ifdef SVGF_DIAG
    xgroup = diag_env("SVGF_ATROUS_XGROUP", xgroup);
    if (xgroup < 1) xgroup = 1;
#endif
    const int ngroups = (xtiles * nbands * S + xgroup - 1) / xgroup;
    const dim3 grid((unsigned)((ngroups + kXcds - 1) / kXcds) * kXcds * xgroup);
#ifdef SVGF_DIAG
    static bool told = false;
    if (!told) {
        told = true;
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)atrous_lds_kernel<ST, S, TX, KR, MODE>, threads, lds);
        fprintf(stderr, "[svgf diag] atrous_lds<ST=%d,S=%d,TX=%d,KR=%d,MODE=%d>: lds %zu B, occupancy %d blocks/CU (planned %d), grid %u (x tiles %d, bands %d, xgroup %d), band %d\n", ST, S, TX, KR, MODE, lds, nb, per_cu, grid.x, xtiles, nbands, xgroup, band);
    }
#endif
    int xrot = 3, xorder = 0;
#ifdef SVGF_DIAG
    xorder = diag_env("SVGF_ATROUS_XORDER", xorder);
    xrot = diag_env("SVGF_ATROUS_XROT", xrot);
#endif
    atrous_lds_kernel<ST, S, TX, KR, MODE><<<grid, dim3(threads), lds, s>>>(g, a, band, nbands, xgroup, xrot, xorder);
    return hipGetLastError();
}

#ifndef SVGF_WAVE_SPECIALISED
#define SVGF_WAVE_SPECIALISED 0     // 1: steps 1-16 through atrous_ws_kernel (compute waves + loader waves, no barriers): a measured alternative,
#endif                              // parity-green and ~10 % slower than atrous_lds_kernel (svgf_atrous_ws.h); not in the product build
#if SVGF_WAVE_SPECIALISED
#include "atrous_ws.h"
#endif
#ifndef SVGF_ROWS4
#define SVGF_ROWS4 0                // bit mask of steps (1, 2, 4, 8, 16) launched through atrous_r4_kernel (four rows per step, 8-wave workgroups):
#endif                              // a measured alternative (svgf_atrous_r4.h), not in the product build
#if SVGF_ROWS4
#include "atrous_r4.h"
#endif

template <int ST, int KR, int MODE>
hipError_t launch_atrous_lds_step_kr(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    // 128-column workgroups (4 waves, 4 per CU) for every step: smaller tiles hit the uniform-normal fast path more
    // often and balance better across the chip; with the oversubscribed grid below they beat 256 columns at every
    // step (A/B on one device: 0.80 vs 0.82 ms per 4K frame), although the 4S-column halo costs 1.5x staging at S = 16.
    bool narrow = a.step <= kNarrowMaxStep;
#ifdef SVGF_DIAG
    narrow = diag_env("SVGF_ATROUS_TX", narrow ? 128 : 256) == 128;
#endif
#ifdef SVGF_KR2_TX128
    if (MODE == 0) switch (a.step) {                 // measurement: two outputs per thread on 128-column workgroups (2 waves)
        case 1: return launch_atrous_lds<ST, 1, 128, 2, SVGF_FORCE_MODE>(g, a, s);
        case 2: return launch_atrous_lds<ST, 2, 128, 2, SVGF_FORCE_MODE>(g, a, s);
        case 4: return launch_atrous_lds<ST, 4, 128, 2, SVGF_FORCE_MODE>(g, a, s);
        case 8: return launch_atrous_lds<ST, 8, 128, 2, SVGF_FORCE_MODE>(g, a, s);
        case 16: return launch_atrous_lds<ST, 16, 128, 2, SVGF_FORCE_MODE>(g, a, s);
        default: return hipErrorInvalidValue;
    }
#endif
    if (SVGF_WAVE_TILE && MODE == 0) switch (a.step) {
        case 1: return launch_atrous_lds<ST, 1, 64, 2, SVGF_FORCE_MODE>(g, a, s);
        case 2: return launch_atrous_lds<ST, 2, 64, 2, SVGF_FORCE_MODE>(g, a, s);
        case 4: return launch_atrous_lds<ST, 4, 64, 2, SVGF_FORCE_MODE>(g, a, s);
        case 8: return launch_atrous_lds<ST, 8, 64, 2, SVGF_FORCE_MODE>(g, a, s);
        case 16: return launch_atrous_lds<ST, 16, 64, 2, SVGF_FORCE_MODE>(g, a, s);
        default: return hipErrorInvalidValue;
    }
#ifdef SVGF_TX64_MAX_STEP
    if (KR == 1 && MODE == 0 && a.step <= SVGF_TX64_MAX_STEP) switch (a.step) {      // measurement: 64-column workgroups of two waves
        case 1: return launch_atrous_lds<ST, 1, 64, 1, SVGF_FORCE_MODE>(g, a, s);
        case 2: return launch_atrous_lds<ST, 2, 64, 1, SVGF_FORCE_MODE>(g, a, s);
        case 4: return launch_atrous_lds<ST, 4, 64, 1, SVGF_FORCE_MODE>(g, a, s);
        case 8: return launch_atrous_lds<ST, 8, 64, 1, SVGF_FORCE_MODE>(g, a, s);
        default: break;
    }
#endif
#if SVGF_ROWS4
    if (KR == 1 && MODE == 0 && narrow && (a.step & SVGF_ROWS4)) switch (a.step) {
        case 1: return launch_atrous_r4<ST, 1>(g, a, s);
        case 2: return launch_atrous_r4<ST, 2>(g, a, s);
        case 4: return launch_atrous_r4<ST, 4>(g, a, s);
        case 8: return launch_atrous_r4<ST, 8>(g, a, s);
        case 16: return launch_atrous_r4<ST, 16>(g, a, s);
        default: break;
    }
#endif
#if SVGF_WAVE_SPECIALISED
    if (KR == 1 && MODE == 0 && narrow) switch (a.step) {
        case 1: return launch_atrous_ws<ST, 1>(g, a, s);
        case 2: return launch_atrous_ws<ST, 2>(g, a, s);
        case 4: return launch_atrous_ws<ST, 4>(g, a, s);
        case 8: return launch_atrous_ws<ST, 8>(g, a, s);
        case 16: return launch_atrous_ws<ST, 16>(g, a, s);
        default: return hipErrorInvalidValue;
    }
#endif
    if (KR == 1 && MODE == 0 && narrow) switch (a.step) {
        case 1: return launch_atrous_lds<ST, 1, 128, 1, SVGF_FORCE_MODE>(g, a, s);
        case 2: return launch_atrous_lds<ST, 2, 128, 1, SVGF_FORCE_MODE>(g, a, s);
        case 4: return launch_atrous_lds<ST, 4, 128, 1, SVGF_FORCE_MODE>(g, a, s);
        case 8: return launch_atrous_lds<ST, 8, 128, 1, SVGF_FORCE_MODE>(g, a, s);
        case 16: return launch_atrous_lds<ST, 16, 128, 1, SVGF_FORCE_MODE>(g, a, s);
        default: return hipErrorInvalidValue;
    }
    switch (a.step) {
        case 1: return launch_atrous_lds<ST, 1, 256, KR, MODE>(g, a, s);
        case 2: return launch_atrous_lds<ST, 2, 256, KR, MODE>(g, a, s);
        case 4: return launch_atrous_lds<ST, 4, 256, KR, MODE>(g, a, s);
        case 8: return launch_atrous_lds<ST, 8, 256, KR, MODE>(g, a, s);
        case 16: return launch_atrous_lds<ST, 16, 256, KR, MODE>(g, a, s);
        default: return hipErrorInvalidValue;
    }
}

template <int ST>
hipError_t launch_atrous_lds_step(const Geo& g, const AtrousArgs& a, hipStream_t s) {
#ifdef SVGF_DIAG
    const int kr = diag_env("SVGF_ATROUS_KR", kDefaultKR);
    const int mode = diag_env("SVGF_ATROUS_MODE", 0);
    if (ST == 0 && mode == 1) return kr == 2 ? launch_atrous_lds_step_kr<0, 2, 1>(g, a, s) : launch_atrous_lds_step_kr<0, 1, 1>(g, a, s);
    if (ST == 0 && mode == 2) return kr == 2 ? launch_atrous_lds_step_kr<0, 2, 2>(g, a, s) : launch_atrous_lds_step_kr<0, 1, 2>(g, a, s);
    if (kr != kDefaultKR) return kr == 2 ? launch_atrous_lds_step_kr<ST, 2, 0>(g, a, s) : launch_atrous_lds_step_kr<ST, 1, 0>(g, a, s);
#endif
    return launch_atrous_lds_step_kr<ST, kDefaultKR, 0>(g, a, s);
}


}  // namespace r02
using r02::launch_atrous_lds_step;
}  // namespace

#ifdef SVGF_STAMPS
// the raw per-wave log: r02::kStampSlots x 16 words (slot = blockIdx * 8 + wave)
extern "C" int svgf_diag_stamp_log(unsigned long long* out, unsigned long long words) {
    const unsigned long long all = (unsigned long long)r02::kStampSlots * 16;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(r02::g_stamp_log), (words < all ? words : all) * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int svgf_diag_stamps(unsigned long long* out, int reset) {
    std::vector<unsigned long long> h((size_t)r02::kStampSlots * 16);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(r02::g_stamp_log), h.size() * sizeof(h[0])) != hipSuccess) return -1;
    for (int i = 0; i < 16; i++) out[i] = 0;
    for (size_t k = 0; k < h.size(); k++) out[k & 15] += h[k];
#ifdef SVGF_STAMPS_PLACEMENT
    unsigned hist[32];
    if (hipMemcpyFromSymbol(hist, HIP_SYMBOL(r02::g_simd_hist), sizeof(hist)) == hipSuccess) {
        fprintf(stderr, "[svgf stamps] waves per (wave of the workgroup, SIMD):");
        for (int w = 0; w < 8; w++) fprintf(stderr, "  w%d: %u %u %u %u", w, hist[w * 4], hist[w * 4 + 1], hist[w * 4 + 2], hist[w * 4 + 3]);
        fprintf(stderr, "\n");
    }
    if (reset) { unsigned zh[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(r02::g_simd_hist), zh, sizeof(zh)); }
#endif
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(r02::g_stamp_log)) != hipSuccess || hipMemset(p, 0, h.size() * sizeof(h[0])) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace svgf
#endif  // SVGF_P2
