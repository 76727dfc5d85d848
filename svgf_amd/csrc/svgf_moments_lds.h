// svgf_moments_lds.h — the spatial estimate of young pixels (filter::FilterMoments, Filter.cuh:430-525) as an LDS-streaming kernel
// for frames in which (nearly) every pixel is young (history < 4: the first three frames of a sequence, after a reset, a resize, a
// camera cut).  The 7x7 window is served from an 8-row LDS ring exactly like the a-trous kernel's 5x5 window (svgf_atrous_lds.h):
// a workgroup streams down a band, 128 columns x 2 rows per step, one output per thread, rows fetched one step ahead — instead of
// 49 x 4 gathers per pixel through L1.
//
// Records are RAW (this stage does not clamp, :450,479), 36 B per staged pixel in four planes:
//   A = {r, g, b, m1}    L = {luminance, depth (0 -> 1e30)}    N = {(nx,ny) half bits, nz}    C = m2
// The 49 taps are one rolling software pipeline (the LDS reads of tap t+D are issued before tap t is consumed).  Three forms of the
// tap, chosen per wave and step from what the waves that staged the eight ring rows reported (mflag; "exact" is sticky per workgroup):
//   uniform  no texel of the ring differs from the workgroup's reference normal: n.n' is the reference normal's own |n|^2 for every
//            tap, so the normal term of the exponent is ONE value per workgroup (a scalar register) and the taps read no normal
//            record, run no v_dot2 / v_log: 9 vector instructions per tap instead of 13.  Same expressions on the same bits as the
//            general form: bit-identical;
//   general  n.n' per tap;
//   exact    a texel of the ring holds a NaN or inf (colour, moments, depth or normal) AND the pixel's sums came out NaN in the general form: the pixel is
//            evaluated again with the luminance and depth terms the way the reference has them —
//            max(|dl| / phi_l, 0.0) is CUDA's fmax, which drops a NaN (:424), so the weight stays finite and the NaN reaches the sums
//            through the channels that hold it (:498-499) — and a zero-normal centre takes no shortcut (its weights are exactly 0,
//            and 0 x NaN is NaN).
// Texels outside the frame come back all-zero from the buffer range check: depth 0 -> sentinel and a zero normal give weight exactly
// 0, which equals skipping the tap (:473); their zero normal differs from any reference normal, so the border waves run the general
// form.
#pragma once
#include "svgf_atrous_taps.h"

namespace svgf {
namespace {

constexpr int kMR = 3;                       // window radius (the reference's, :465)
constexpr int kMRing = kRS + 2 * kMR;        // 8 ring rows
constexpr int kMTX = 128;                    // columns of a workgroup (two waves per row; round 4: 256 -> 128 and four resident rounds of
                                             // workgroups instead of one: -13 % per launch, profiles/r04_small_experiments.txt block 5)
constexpr int kMTapDepth = 3;                // LDS reads run this many taps ahead of the arithmetic
constexpr int kMRounds = 4;                  // workgroups per resident slot of the chip the bands are cut for
constexpr int kMRecBytes = 36;

__device__ __forceinline__ constexpr int len_class7(int xx, int yy) {   // |(xx,yy)|^2 in {1,2,4,5,8,9,10,13,18}
    const int l2 = xx * xx + yy * yy;
    return l2 == 1 ? 0 : l2 == 2 ? 1 : l2 == 4 ? 2 : l2 == 5 ? 3 : l2 == 8 ? 4 : l2 == 9 ? 5 : l2 == 10 ? 6 : l2 == 13 ? 7 : 8;
}

struct MomCentre { float lc, zc, ncz; uint32_t nc01; float il; float iz[9]; };
struct MomSums { float sw, sm2; f32x2 srg, sbm; };

// e0: the uniform form's exponent of the normal term (wave-uniform)
template <int MODE>
__device__ __forceinline__ void moments_taps49(const f32x4* recA, const f32x2* recL, const f32x2* recN, const float* recC,
                                               const int (&rowbase)[2 * kMR + 1], const MomCentre& c, float phi_n, float e0, MomSums& s) {
    constexpr int NW = 2 * kMR + 1, NT = NW * NW, D = kMTapDepth;
    f32x4 qA[NT];
    f32x2 qL[NT], qN[NT];
    float qC[NT];
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int at = rowbase[t / NW] + t % NW;
        qA[t] = recA[at];
        qL[t] = ((const volatile lds_f32x2*)recL)[at];                                   // single ds_read_b64 (see taps24)
        if (MODE != kTapsUniform) qN[t] = ((const volatile lds_f32x2*)recN)[at];
        qC[t] = recC[at];
    };
#pragma unroll
    for (int t = 0; t < D; t++) issue(t);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t + D < NT) issue(t + D);
        asm volatile("" ::: "memory");
        const int yy = t / NW - kMR, xx = t % NW - kMR;
        const f32x4 A = qA[t];
        const f32x2 L = qL[t];
        float e;
        if constexpr (MODE == kTapsUniform) {
            e = e0;
        } else {
            const f32x2 N = qN[t];
            const float d = clamp01(fmaf(N.y, c.ncz, dot2_h2(__float_as_uint(N.x), c.nc01)));
            e = hw_log2(d) * phi_n;
        }
        if constexpr (MODE == kTapsNaN) e -= fmaxf(fabsf(L.x - c.lc) * c.il, 0.0f);      // fmax(NaN, 0) = 0, :424
        else e = fmaf(-fabsf(L.x - c.lc), c.il, e);
        if (xx != 0 || yy != 0) {                                                        // phiDepth == 0 -> wZ = 0 at the centre, :420
            if constexpr (MODE == kTapsNaN) e -= fmaxf(fabsf(L.y - c.zc) * c.iz[len_class7(xx, yy)], 0.0f);   // fmax(NaN, 0) = 0, :424 (a NaN depth)
            else e = fmaf(-fabsf(L.y - c.zc), c.iz[len_class7(xx, yy)], e);
        }
        const float w = hw_exp2(e);
        s.sw += w;                                                                       // :497-499
        s.srg = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, s.srg);
        s.sbm = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.z, A.w}, s.sbm);
        s.sm2 = fmaf(w, qC[t], s.sm2);
        asm volatile("" : "+v"(s.sw), "+v"(s.srg), "+v"(s.sbm), "+v"(s.sm2) :: "memory");
    }
}

template <int ST>
__global__ __launch_bounds__(kMTX* kRS, 4) void moments_lds_kernel(Geo g, MomentsArgs a, int band_rows) {
    constexpr int TX = kMTX, WL = TX + 2 * kMR, CB = ST == 0 ? 16 : 8, MB = ST == 0 ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* recA = (f32x4*)smem;
    f32x2* recL = (f32x2*)(recA + kMRing * WL);
    f32x2* recN = recL + kMRing * WL;
    float* recC = (float*)(recN + kMRing * WL);
    uint32_t* mflag = (uint32_t*)(recC + kMRing * WL);     // [kMRing][4]: what each wave of a row group reported of its ring row (two of the four words are used),
    constexpr int kBadWord = kMRing * 4;                   // then the workgroup's sticky "a NaN / inf was staged" word
    uint32_t* nref = mflag + kBadWord + 1;                 // the workgroup's reference normal {(nx,ny) bits, nz half bits}

    const int t = threadIdx.x, lane = t & 63, col = t % TX;
    const int rg = __builtin_amdgcn_readfirstlane(t / TX);
    const int wig = __builtin_amdgcn_readfirstlane((t % TX) >> 6);
    const int x0 = blockIdx.x * TX;
    const int nrows = g.ye - g.yb;
    const int j0 = blockIdx.y * band_rows;
    if (j0 >= nrows) return;
    const int j1 = min(nrows, j0 + band_rows);

    const int gx = x0 + col, oli = col + kMR;
    const bool halo_wave = wig == 0;
    const bool has_halo = halo_wave && lane < 2 * kMR;      // six halo pixels per row: lanes 0-5 of the row group's first wave
    const int hx = (lane < kMR) ? x0 - kMR + lane : x0 + TX + lane - kMR;
    const int hli = (lane < kMR) ? lane : TX + lane;
    const bool own_ok = gx < g.W, halo_ok = has_halo && hx >= 0 && hx < g.W;
    const unsigned vo = own_ok ? (unsigned)gx : 0u, vh = halo_ok ? (unsigned)hx : 0u;
    const unsigned vo_c = own_ok ? vo * CB : kOob, vo_mo = own_ok ? vo * MB : kOob, vo_m = own_ok ? vo * 16u + 8u : kOob, vo_n = own_ok ? vo * 8u : kOob, vo_h = own_ok ? vo : kOob;
    const unsigned vh_c = halo_ok ? vh * CB : kOob, vh_mo = halo_ok ? vh * MB : kOob, vh_m = halo_ok ? vh * 16u + 8u : kOob, vh_n = halo_ok ? vh * 8u : kOob;

    const unsigned npx = (unsigned)g.rows * (unsigned)g.W;
    auto mk = [](const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rs_out = mk(a.out, npx * CB), rs_h = mk(a.hist, npx), rs_m = mk(a.motion, npx * 16u);

    struct Px { u32x4 c; u32x2 mo; unsigned z; u32x2 n; };
    // a row outside the frame / the strip is read through zero-length resources: every texel comes back zero
    auto load_px = [&](Px& p, bool rok, int srow, unsigned o_c, unsigned o_mo, unsigned o_m, unsigned o_n) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rc = mk(a.colour, rok ? npx * CB : 0u), rmo = mk(a.mom, rok ? npx * MB : 0u),
                                     rm = mk(a.motion, rok ? npx * 16u : 0u), rn = mk(a.normal, rok ? npx * 8u : 0u);
        if constexpr (ST == 0) { p.c = __builtin_amdgcn_raw_buffer_load_b128(rc, o_c, srow * CB, 0); p.mo = __builtin_amdgcn_raw_buffer_load_b64(rmo, o_mo, srow * MB, 0); }
        else { const u32x2 c2 = __builtin_amdgcn_raw_buffer_load_b64(rc, o_c, srow * CB, 0); p.c = (u32x4){c2.x, c2.y, 0u, 0u}; p.mo = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(rmo, o_mo, srow * MB, 0), 0u}; }
        p.z = __builtin_amdgcn_raw_buffer_load_b32(rm, o_m, srow * 16, 0);
        p.n = __builtin_amdgcn_raw_buffer_load_b64(rn, o_n, srow * 8, 0);
    };
    auto row_of = [&](int j, bool& rok) __attribute__((always_inline)) {      // scalar: decimated == actual rows here
        const int y = g.yb + j, yl = y - g.y0;
        rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
        return rok ? yl * g.W : 0;
    };
    struct Staged { Px o, h; };
    auto fetch = [&](int jn, Staged& st) __attribute__((always_inline)) {     // rows jn, jn+1: this wave's is jn+rg
        bool rok; const int srow = row_of(jn + rg, rok);
        load_px(st.o, rok, srow, vo_c, vo_mo, vo_m, vo_n);
        if (halo_wave) load_px(st.h, rok, srow, vh_c, vh_mo, vh_m, vh_n);
    };
    uint32_t ref01 = 0, refz = 0;
    // -> the texel's normal is not the reference normal; bad |= its luminance or one of its moments is NaN / inf
    auto commit_one = [&](const Px& p, int at, bool& bad) __attribute__((always_inline)) -> bool {
        float4 c; float2 m;
        if constexpr (ST == 0) { c = make_float4(__uint_as_float(p.c.x), __uint_as_float(p.c.y), __uint_as_float(p.c.z), __uint_as_float(p.c.w)); m = make_float2(__uint_as_float(p.mo.x), __uint_as_float(p.mo.y)); }
        else { const float2 lo = unpack_h2(p.c.x), hi = unpack_h2(p.c.y); c = make_float4(lo.x, lo.y, hi.x, hi.y); m = unpack_h2(p.mo.x); }
        float z = __uint_as_float(p.z);
        if (z == 0.0f) z = kSkyZ;                                                   // GetDepth, :199-207
        const float lum = lum_exact(c.x, c.y, c.z);
        recA[at] = (f32x4){c.x, c.y, c.z, m.x};                                     // raw loads, :479-480
        recL[at] = (f32x2){lum, z};
        recN[at] = (f32x2){__uint_as_float(p.n.x), unpack_h2(p.n.y).x};
        recC[at] = m.y;
        // ... or its depth is NaN / inf, or a component of its normal is (half exponent 31): the general form would turn the weight NaN where the
        // reference drops the term (`max(weightZ, 0.0)`, :424) or reads saturate(NaN) as 0 (:419)
        bad = bad | !(fabsf(fmaf(z, 0.0f, lum + (m.x + m.y))) < __builtin_inff());  // (an overflowing sum of finite values: the exact form, needlessly)
        bad = bad | ((((p.n.x & 0x7c007c00u) + 0x04000400u) & 0x80008000u) != 0u) | ((p.n.y & 0x7c00u) == 0x7c00u);
        return (p.n.x != ref01) | ((p.n.y & 0xffffu) != refz);
    };
    auto commit = [&](int sl, const Staged& st) __attribute__((always_inline)) {
        int so = sl + rg; so = so >= kMRing ? so - kMRing : so;
        bool bad = false;
        bool differs = commit_one(st.o, so * WL + oli, bad);
        if (halo_wave) { if (has_halo) differs = commit_one(st.h, so * WL + hli, bad) | differs; }
        if (wave_any(bad)) { if (lane == 0) mflag[kBadWord] = 1u; }                  // rare, sticky
        const bool wave_differs = wave_any(differs);
        if (lane == 0) mflag[so * 4 + wig] = wave_differs ? kFlagNormal : 0u;       // a ring slot is always staged by the same waves
    };
    // the centre's own history byte and ddepth come straight from the planes, one step ahead (L2 hits)
    struct Centre { unsigned h; unsigned dz; };
    auto fetch_centre = [&](int j, Centre& c) __attribute__((always_inline)) {
        bool rok; const int srow = row_of(j + rg, rok);
        if (rok) { c.h = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_h, vo_h, srow, 0); c.dz = __builtin_amdgcn_raw_buffer_load_b32(rs_m, own_ok ? vo * 16u + 12u : kOob, srow * 16, 0); }
        else { c.h = 255u; c.dz = 0u; }
    };

    // prologue: ring rows 0..7 = rows j0-3 .. j0+4.  The pair that holds row j0 (ring rows 2, 3 = rows j0-1, j0) is committed first:
    // the pixel (x0, j0) is inside the frame, and its normal is the workgroup's reference normal.
    if (t <= kBadWord) mflag[t] = 0u;
    {   // all eight ring rows requested at once — ONE round of memory latency instead of four (the tap loop's registers are free here; -1 % per launch)
        Staged s0, s1, s2, s3;
        fetch(j0 - kMR + 2, s0); fetch(j0 - kMR + 0, s1); fetch(j0 - kMR + 4, s2); fetch(j0 - kMR + 6, s3);
        if (t == TX) { nref[0] = s0.o.n.x; nref[1] = s0.o.n.y & 0xffffu; }     // ring row 3 = row j0 is staged by row group 1: its thread of column 0
        __syncthreads();
        ref01 = nref[0]; refz = nref[1];
        commit(2, s0); commit(0, s1); commit(4, s2); commit(6, s3);
    }
    Centre cen, cen_next;
    fetch_centre(j0, cen);
    __syncthreads();

    const float phi_n = a.phi_normal;                      // != 0 (launcher)
    const float il = hw_rcp(a.phi_colour) * kLog2e;        // :460
    // the uniform form's normal term, once per workgroup (the expressions of the general tap on the reference normal's bits)
    const float ref_nz = unpack_h2(refz).x;
    const float e0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(hw_log2(clamp01(fmaf(ref_nz, ref_nz, dot2_h2(ref01, ref01)))) * phi_n)));
    int slot0 = 0;
    for (int j = j0; j < j1; j += kRS) {
        const bool more = (j + kRS) < j1;
        Staged fs;
        if (more) { fetch(j + kRS + kMR, fs); fetch_centre(j + kRS, cen_next); }       // rows j+5, j+6 enter the ring next step

        int rowbase[2 * kMR + 1];
#pragma unroll
        for (int r = 0; r <= 2 * kMR; r++) { int sl = slot0 + rg + r; sl = sl >= kMRing ? sl - kMRing : sl; rowbase[r] = sl * WL + col; }
        const int ci = rowbase[kMR] + kMR;
        const f32x2 cL = recL[ci], cN = recN[ci];
        MomCentre c;
        c.lc = cL.x; c.zc = cL.y; c.ncz = cN.y; c.nc01 = __float_as_uint(cN.x); c.il = il;
        const float h = (float)cen.h;                                               // :442
        const float dzc = c.zc == kSkyZ ? 0.0f : __uint_as_float(cen.dz);
        const float izb = hw_rcp(fmaxf(dzc, 1e-8f) * 3.0f) * kLog2e;                // :461
        c.iz[0] = izb; c.iz[1] = izb * 0.70710678118654752f; c.iz[2] = izb * 0.5f; c.iz[3] = izb * 0.44721359549995794f; c.iz[4] = izb * 0.35355339059327376f;
        c.iz[5] = izb * 0.33333333333333333f; c.iz[6] = izb * 0.31622776601683794f; c.iz[7] = izb * 0.27735009811261456f; c.iz[8] = izb * 0.23570226039551584f;
        // every wave's word of every ring row (lanes 0 .. kBadWord-1) and the workgroup's sticky word (lane kBadWord): one read, one compare
        const unsigned long long flagged = __builtin_amdgcn_ballot_w64(lane <= kBadWord && mflag[lane <= kBadWord ? lane : 0] != 0u);
        const bool uniform = !a.no_fastpath && (flagged & ((1ull << kBadWord) - 1ull)) == 0ull;
        // ... or PhiColour is 0 (the GUI's range starts there, GUI.cpp:992): |dl| / 0 is inf — or NaN for the taps of the centre's own luminance, the
        // centre itself among them — which `max(., 0.0)` = fmax turns into "no term" (:424): every pixel's fused-exponent sums are NaN then
        const bool exact = (flagged >> kBadWord) != 0ull || !(il < __builtin_inff());
        MomSums s{0.0f, 0.0f, {0.f, 0.f}, {0.f, 0.f}};
        // a cleared sky texel (zero normal): every weight is exactly 0, the result (0,0,0,0) while the window is finite (see moments_pixel)
        const bool zero_normal = !exact && ((c.nc01 & 0x7fff7fffu) == 0u) && (c.ncz == 0.0f);
        const bool need = (h < 4.0f) && !zero_normal && (j + rg < j1);
        if (wave_any(need)) {
            if (uniform) moments_taps49<kTapsUniform>(recA, recL, recN, recC, rowbase, c, phi_n, e0, s);
            else moments_taps49<kTapsGeneral>(recA, recL, recN, recC, rowbase, c, phi_n, e0, s);
            if (exact) {
                // A workgroup that has staged a non-finite texel: the pixels whose sums came out NaN — every pixel with a NaN (or inf - inf, 0 x inf)
                // in its window, and no other — are evaluated again the reference's way; the others keep the bits every other workgroup would
                // give them, however the frame is cut into tiles, bands or strips (round 4 took the exact form for the whole workgroup).
                const bool redo = need && (__builtin_isunordered(s.sw, s.sm2) | __builtin_isunordered(s.srg.x, s.srg.y) | __builtin_isunordered(s.sbm.x, s.sbm.y));
                if (wave_any(redo)) {
                    MomSums s2{0.0f, 0.0f, {0.f, 0.f}, {0.f, 0.f}};
                    moments_taps49<kTapsNaN>(recA, recL, recN, recC, rowbase, c, phi_n, e0, s2);
                    if (redo) s = s2;
                }
            }
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f * (4.0f / h));
        if (!zero_normal) {
            const float sw = fmaxf(s.sw, 1e-6f);                                    // :505
            const float inv = 1.0f / sw;
            const float m1 = s.sbm.y * inv, m2 = s.sm2 * inv;
            o = make_float4(s.srg.x * inv, s.srg.y * inv, s.sbm.x * inv, (m2 - m1 * m1) * (4.0f / h));   // :507-516
        }
        if (more) {
            lds_barrier();
            commit(slot0, fs);
            slot0 += kRS; if (slot0 >= kMRing) slot0 -= kMRing;
            lds_barrier();
        }
        if (j + rg < j1) {
            const int srow = (g.yb + j + rg - g.y0) * g.W;
            if (h < 4.0f) {
                if constexpr (ST == 0) __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)}, rs_out, vo_c, srow * CB, 0);
                else __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_h2(o.x, o.y), pack_h2(o.z, o.w)}, rs_out, vo_c, srow * CB, 0);
            } else if (!a.cold_only) {                                              // :521 copy
                const __amdgpu_buffer_rsrc_t rs_c = mk(a.colour, npx * CB);
                if constexpr (ST == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_amdgcn_raw_buffer_load_b128(rs_c, vo_c, srow * CB, 0), rs_out, vo_c, srow * CB, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_amdgcn_raw_buffer_load_b64(rs_c, vo_c, srow * CB, 0), rs_out, vo_c, srow * CB, 0);
            }
        }
        cen = cen_next;
    }
}

template <int ST>
hipError_t launch_moments_lds(const Geo& g, const MomentsArgs& a, hipStream_t s) {
    constexpr int WL = kMTX + 2 * kMR;
    constexpr size_t lds = (size_t)kMRing * WL * kMRecBytes + (kMRing * 4 + 1 + 2) * sizeof(uint32_t);
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(moments_lds_kernel<ST>, lds, attr_done); e != hipSuccess) return e;
    const int nrows = g.ye - g.yb, xtiles = (g.W + kMTX - 1) / kMTX;
    constexpr int per_cu = (int)((160 * 1024) / lds);
    int nbands = kMRounds * per_cu * num_cus() / xtiles;                        // resident rounds x workgroups per CU (LDS: 38.6 KB each, four per CU)
    if (nbands < 1) nbands = 1;
    int band = (nrows + nbands - 1) / nbands;
    if (band < 8) band = 8;
    band = (band + kRS - 1) / kRS * kRS;
    nbands = (nrows + band - 1) / band;
    moments_lds_kernel<ST><<<dim3(xtiles, nbands), dim3(kMTX * kRS), lds, s>>>(g, a, band);
    return hipGetLastError();
}

}  // namespace
}  // namespace svgf
