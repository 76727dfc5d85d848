// svgf_atrous_taps.h — one pixel of the edge-avoiding à-trous wavelet (filter::FilterKernel, Filter.cuh:527-624) served from LDS
// records: the code both streaming kernels (svgf_atrous_lds.h, svgf_atrous_fused.h) run per output pixel, so that one launch per
// iteration and iterations 0 + 1 in one launch give the same bits.
//
// The edge-stopping weight (computeWeight, :407-427) times the kernel weight (:540,582) is ONE v_exp_f32 of a fused exponent:
//   w = exp2( log2 K  +  phi_n * log2 sat(n.n')  -  |dl| * log2(e)/phi_l  -  |dz| * log2(e)/(phi_z |offset|) )
// 8 vector instructions per tap on the uniform-normal path (v_pk_add of {dl, dz}, two FMAs with -|x| modifiers, v_exp, w^2, sum w,
// two v_pk_fma for the four channels), 13 on the general path (v_dot2_f32_f16, clamp-FMA, v_log, FMA more).
#pragma once
#include "svgf_device.h"

namespace svgf {
namespace {

// LDS BYTE addresses of a thread's five tap rows (its own column's record in each): A = the colour records (16 B), L = the
// {luminance, depth} records (8 B); a row's normal record sits NOFF bytes behind its L record (same stride).  Byte addresses, built as
// (per-lane constant) + (per-row scalar): one v_add per row and plane per step — index arithmetic (row * width + column, then x16 and
// x8) cost three.  Tap (row r, column cc) is an immediate offset from there.
struct TapRows { uint32_t a[5], l[5]; };
__device__ __forceinline__ f32x4 lds_read_a(uint32_t addr) { return *(const lds_f32x4*)(uintptr_t)addr; }
// volatile: keeps these as single ds_read_b64 (2 LDS cycles each); merged into ds_read2_b64 they take 8
__device__ __forceinline__ f32x2 lds_read_l(uint32_t addr) { return *(const volatile lds_f32x2*)(uintptr_t)addr; }
// the same from record indices (row * width + column) into planes recA / recL
__device__ __forceinline__ TapRows rows_from_index(const f32x4* recA, const f32x2* recL, const int (&rowbase)[5]) {
    TapRows t;
    const uint32_t ba = lds_addr(recA), bl = lds_addr(recL);
#pragma unroll
    for (int r = 0; r < 5; r++) { t.a[r] = ba + (uint32_t)rowbase[r] * 16u; t.l[r] = bl + (uint32_t)rowbase[r] * 8u; }
    return t;
}

// What a thread keeps of its centre pixel (the set-up of :543-568).
struct TapCentre {
    f32x4 A;             // clamped colour + variance
    f32x2 lz;            // luminance, depth; the depth of a SKY centre (sentinel 1e30) is replaced by -1e30, see centre_setup
    uint32_t n01;        // (nx, ny) half bits
    float nz;
    float il;            // log2(e) / phi_l
    float iz[5];         // log2(e) / (phi_z * |offset|) per offset length class
    bool sky;            // GetDepth() == sentinel: the pixel is copied (:554-558)
};
// A, L, N: the centre's LDS records; ddepth: its depth derivative as stored; S: the iteration's step; k_colour = log2(e) / PhiColour.
// A sky centre is copied, not filtered (:554-558).  Instead of selecting between the filtered value and the copy afterwards (four
// v_cndmask), its depth enters the taps as -1e30: every tap — surface or sky (+1e30) — is then at least 1e30 away in depth, its
// weight exp2(-1e30 * iz) is exactly 0, the sums stay {1, colour} and the normalisation multiplies by rcp(1) = 1: the copy, bit for bit.
template <int S>
__device__ __forceinline__ TapCentre centre_setup(f32x4 A, f32x2 L, f32x2 N, float ddepth, float k_colour) {
    TapCentre c;
    c.sky = L.y == kSkyZ;
    c.A = A; c.lz = (f32x2){L.x, c.sky ? -kSkyZ : L.y}; c.n01 = __float_as_uint(N.x); c.nz = N.y;
    c.il = inv_phi_l_log2e(A.w, k_colour);                                               // :562
    // :563: log2(e) / (max(ddepth, 1e-6) * S); (GetDepth gives a sky texel ddepth 0: irrelevant now, any positive value kills its taps)
    const float izb = hw_rcp(fmaxf(ddepth, 1e-6f) * ((float)S * 0.6931471805599453f));
    const f32x2 i12 = (f32x2){izb, izb} * (f32x2){0.70710678118654752f, 0.5f}, i34 = (f32x2){izb, izb} * (f32x2){0.44721359549995794f, 0.35355339059327376f};
    c.iz[0] = izb; c.iz[1] = i12.x; c.iz[2] = i12.y; c.iz[3] = i34.x; c.iz[4] = i34.y;
    return c;
}

// UNI: the exponent of the normal term + kernel weight per kernel-weight class, from a normal's own |n|^2
struct UniBase { float e[5]; };
__device__ __forceinline__ UniBase uni_base(uint32_t n01, float nz, float phi_n) {
    UniBase u;
    const float lg = hw_log2(clamp01(fmaf(nz, nz, dot2_h2(n01, n01))));
    u.e[0] = fmaf(lg, phi_n, klog2(0, 1)); u.e[1] = fmaf(lg, phi_n, klog2(1, 1)); u.e[2] = fmaf(lg, phi_n, klog2(0, 2));
    u.e[3] = fmaf(lg, phi_n, klog2(1, 2)); u.e[4] = fmaf(lg, phi_n, klog2(2, 2));
    return u;
}

// MODE: kTapsGeneral, kTapsUniform (the uniform-normal path), or kTapsNaN — the general path in the form that treats a NaN the way
// the reference does (a band that is run again because a NaN showed in its output, svgf_atrous_lds.h): a NaN luminance
// difference counts as 0 — `max(weightLillum, 0.0)` in :424 is CUDA's fmax, which drops the NaN — so the weight stays finite and the
// NaN reaches the sums only through the channels that hold it (:608); a NaN depth difference (a NaN depth in the G-buffer, or inf - inf)
// counts as 0 the same way (`max(weightZ, 0.0)`, :424); saturate(n.n') of a NaN is 0 (:419).  The other two paths fold
// |dl| / phi_l and |dz| / phi_z into FMAs of the exponent, which makes the weight — and with it all four channels — NaN.
constexpr int kTapsGeneral = 0, kTapsUniform = 1, kTapsNaN = 2;
template <int CS, int D, int MODE, int NOFF>
__device__ __forceinline__ void taps24(const TapRows& rows, const TapCentre& c, float phi_n, float& sw, f32x2& srg, f32x2& sbv, const UniBase* shared_base) {
    // shared_base: the workgroup's reference normal's values, computed once (every surface centre of a uniform wave carries exactly
    // those normal bits, so they are what uni_base(c.n01, c.nz) would give); null: per pixel
    constexpr bool UNI = MODE == kTapsUniform;
    UniBase ub;
    if constexpr (UNI) ub = shared_base ? *shared_base : uni_base(c.n01, c.nz, phi_n);
    const float (&ebase)[5] = ub.e;
    constexpr int NT = 25;
    f32x4 qA[NT];
    f32x2 qL[NT], qN[NT];
    auto issue = [&](int t) __attribute__((always_inline)) {
        if (t == 12) return;                                                             // the centre itself is no tap (:584)
        const int r = t / 5, cc = t % 5;
        qA[t] = lds_read_a(rows.a[r] + cc * CS * 16);
        qL[t] = lds_read_l(rows.l[r] + cc * CS * 8);
        if (!UNI) qN[t] = lds_read_l(rows.l[r] + cc * CS * 8 + NOFF);
    };
#pragma unroll
    for (int t = 0; t < D; t++) issue(t);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t + D < NT) issue(t + D);
        asm volatile("" ::: "memory");
        if (t == 12) continue;
        const int yy = t / 5 - 2, xx = t % 5 - 2;
        const int axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
        const f32x4 A = qA[t];
        const f32x2 dlz = qL[t] - c.lz;
        float e;
        if constexpr (UNI) {
            e = ebase[kernel_class(axx, ayy)];
        } else {
            const f32x2 N = qN[t];
            float d = fmaf(N.y, c.nz, dot2_h2(__float_as_uint(N.x), c.n01));
            if constexpr (MODE == kTapsNaN) d = clamp01_hw(d != d ? 0.0f : d);           // (the wave runs with keep_nan_in_clamps())
            else d = clamp01(d);
            e = fmaf(hw_log2(d), phi_n, klog2(axx, ayy));
        }
        if constexpr (MODE == kTapsNaN) e -= fmaxf(fabsf(dlz.x) * c.il, 0.0f);          // fmax(NaN, 0) = 0, :424
        else e = fmaf(-fabsf(dlz.x), c.il, e);
        if constexpr (MODE == kTapsNaN) e -= fmaxf(fabsf(dlz.y) * c.iz[len_class(xx, yy)], 0.0f);   // fmax(NaN, 0) = 0, :424
        else e = fmaf(-fabsf(dlz.y), c.iz[len_class(xx, yy)], e);
        const float w = hw_exp2(e);
        const f32x2 ww = {w, w * w};                                                     // weights of (b, variance): :604-608
        sw += w;                                                                         // :607
        srg = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, srg);           // accumulators packed by channel pairs:
        sbv = __builtin_elementwise_fma(ww, (f32x2){A.z, A.w}, sbv);                     // (r,g) and (b,variance), one v_pk_fma_f32 each
        asm volatile("" : "+v"(sw), "+v"(srg), "+v"(sbv) :: "memory");
    }
}

// One pixel: taps + normalisation (:554-558,567-568,615).  `wave_has_surface` (a wave whose centres are all sky has nothing to
// filter) and `uniform` are wave-uniform.
//
// EXACT: the band is run again because a NaN showed in its output (svgf_atrous_lds.h).  In the second pass a pixel takes the exact form's
// value only if its FAST result holds a NaN — the test that sent the band here (general taps: the uniform form computes the same bits);
// a pixel whose window is finite keeps its first-pass bits whichever workgroup, band or strip it falls into next to a NaN (ADVICE r04:
// the whole band used to take the exact form's rounding of the luminance term, and strips / row ranges then differed from the whole frame
// in the last bit around a NaN texel).  Two ways to say that: *fast_bad receives the per-lane test and the exact value is returned for EVERY
// lane — the caller stores only where the test holds, the other texels are in memory from the first pass (no register kept across the
// second set of taps: the streaming kernel has none to spare) — or, with fast_bad == nullptr, the merged value is returned (the pair
// launch, whose iteration-0 result goes into an LDS ring, not to memory).
// KEEP_FAST = false (the pair launch, which has no register for a second set of taps either): the exact form for every pixel of the second pass,
// as in round 4 — around a NaN texel its finite pixels then round as the exact form does (within the stage tolerance).
template <int CS, int D, int NOFF, bool EXACT = false, bool KEEP_FAST = true>
__device__ __forceinline__ float4 filter_px(const TapRows& rows, const TapCentre& c, float phi_n, bool wave_has_surface, bool uniform, const UniBase* shared_base = nullptr,
                                            bool* fast_bad = nullptr, uint32_t* path_count = nullptr) {
    float sw = 1.0f;                                                                     // :567
    f32x2 srg = {c.A.x, c.A.y}, sbv = {c.A.z, c.A.w};                                    // :568
    if constexpr (EXACT && !KEEP_FAST) {
        if (wave_has_surface) {
            taps24<CS, D, kTapsNaN, NOFF>(rows, c, phi_n, sw, srg, sbv, nullptr);
            if (c.sky) { sw = 1.0f; srg = (f32x2){c.A.x, c.A.y}; sbv = (f32x2){c.A.z, c.A.w}; }      // copied (:554-558): weights exactly 0, but 0 x NaN is not
        }
        const float inv = hw_rcp(sw);
        return make_float4(srg.x * inv, srg.y * inv, sbv.x * inv, sbv.y * (inv * inv));
    }
    // path_count (svgf_path_stats_enable, the caller's scalar report word): + 1 for a step that filters a surface pixel, + 0x10000 more on the uniform path
    if (wave_has_surface) {
        if (!EXACT && uniform) { taps24<CS, D, kTapsUniform, NOFF>(rows, c, phi_n, sw, srg, sbv, shared_base); if (path_count) *path_count += 0x10001u; }
        else { taps24<CS, D, kTapsGeneral, NOFF>(rows, c, phi_n, sw, srg, sbv, nullptr); if (path_count) *path_count += 1u; }
    }
    const float inv = hw_rcp(sw);                                                        // sw >= 1 (a sky centre: exactly 1, and the sums are its colour)
    const float4 o = make_float4(srg.x * inv, srg.y * inv, sbv.x * inv, sbv.y * (inv * inv));  // :615
    if constexpr (!EXACT) return o;
    else {
        const bool bad = __builtin_isunordered(o.x, o.w);
        if (fast_bad) *fast_bad = bad;
        float sw2 = 1.0f;
        f32x2 srg2 = {c.A.x, c.A.y}, sbv2 = {c.A.z, c.A.w};
        if (wave_has_surface && (fast_bad || wave_any(bad))) {
            taps24<CS, D, kTapsNaN, NOFF>(rows, c, phi_n, sw2, srg2, sbv2, nullptr);
            // a sky centre is copied (:554-558): its taps' weights are exactly 0, but 0 x NaN is not
            if (c.sky) { sw2 = 1.0f; srg2 = (f32x2){c.A.x, c.A.y}; sbv2 = (f32x2){c.A.z, c.A.w}; }
        }
        const float inv2 = hw_rcp(sw2);
        const float4 o2 = make_float4(srg2.x * inv2, srg2.y * inv2, sbv2.x * inv2, sbv2.y * (inv2 * inv2));
        if (fast_bad) return o2;
        return bad ? o2 : o;
    }
}

}  // namespace
}  // namespace svgf
