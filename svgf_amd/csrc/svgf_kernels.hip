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
#include <atomic>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

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

constexpr float kSkyZ = 1e30f;                     // GetDepth sentinel, Filter.cuh:204
constexpr float kLog2e = 1.4426950408889634f;

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 unpack_h2(uint32_t u) {
    half2_t h = __builtin_bit_cast(half2_t, u);
    return make_float2((float)h.x, (float)h.y);
}
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {   // round-to-nearest-even, as __float2half
    half2_t h;
    h.x = (_Float16)a;
    h.y = (_Float16)b;
    return __builtin_bit_cast(uint32_t, h);
}

// Storage traits: ST = 0 fp32 (float4/float2), ST = 1 fp16 (half4/half2, Filter.cuh:15-16).
template <int ST> struct Store;
template <> struct Store<0> {
    __device__ static __forceinline__ float4 ld4(const void* p, size_t i) { return ((const float4*)p)[i]; }
    __device__ static __forceinline__ void st4(void* p, size_t i, float4 v) { ((float4*)p)[i] = v; }
    __device__ static __forceinline__ float2 ld2(const void* p, size_t i) { return ((const float2*)p)[i]; }
    __device__ static __forceinline__ void st2(void* p, size_t i, float2 v) { ((float2*)p)[i] = v; }
};
template <> struct Store<1> {
    __device__ static __forceinline__ float4 ld4(const void* p, size_t i) {
        uint2 r = ((const uint2*)p)[i];
        float2 a = unpack_h2(r.x), b = unpack_h2(r.y);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    __device__ static __forceinline__ void st4(void* p, size_t i, float4 v) {
        ((uint2*)p)[i] = make_uint2(pack_h2(v.x, v.y), pack_h2(v.z, v.w));
    }
    __device__ static __forceinline__ float2 ld2(const void* p, size_t i) { return unpack_h2(((const uint32_t*)p)[i]); }
    __device__ static __forceinline__ void st2(void* p, size_t i, float2 v) { ((uint32_t*)p)[i] = pack_h2(v.x, v.y); }
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float4 clamp01(float4 v) { return make_float4(clamp01(v.x), clamp01(v.y), clamp01(v.z), clamp01(v.w)); }

// (z, dz) of a motion texel; depth 0 = sky sentinel (Filter.cuh:199-207)
__device__ __forceinline__ void depth_of(float4 m, float& z, float& dz) {
    z = m.z; dz = m.w;
    if (z == 0.0f) { z = kSkyZ; dz = 0.0f; }
}
__device__ __forceinline__ float3 normal_of(uint2 n) {
    float2 a = unpack_h2(n.x), b = unpack_h2(n.y);
    return make_float3(a.x, a.y, b.x);
}
// One guide texel (16 B): what the wavelet iterations and the NEXT frame's reprojection test read of a G-buffer texel:
// {depth, ddepth} as stored (raw: depth_of() is applied by the reader), (nx, ny) half bits, (nz, instance ID) half bits.
__device__ __forceinline__ uint4 guide_texel(float4 motion, uint2 normal, uint2 uv) {
    return make_uint4(__float_as_uint(motion.z), __float_as_uint(motion.w), normal.x, (normal.y & 0xffffu) | (uv.y & 0xffff0000u));
}
// glm::dot order; exact (no contraction) — used by threshold tests
__device__ __forceinline__ float dot3_exact(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float dot3_fma(float3 a, float3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// CalculateLuminance, Filter.cuh:260-263, in the reference's operation order: where the temporal variance is 0 the
// a-trous weights amplify a one-ulp luminance difference ~1e4 times (Tolerance), so no FMA here
__device__ __forceinline__ float lum_exact(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
__device__ __forceinline__ float mix_exact(float x, float y, float a) { return x * (1.0f - a) + y * a; }   // glm::mix

__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Edge-stopping weight, computeWeight Filter.cuh:407-427:
//   w = exp(-max(|dl|/phi_l,0) - max(|dz|/phi_z,0)) * pow(saturate(n.n'), phi_n)
// evaluated as exp2( phi_n*log2(sat(n.n')) - (max(|dl|*il,0) + |dz|*iz)*log2(e) ) with il = 1/phi_l,
// iz = 1/phi_z precomputed per pixel; phi_n == 0 drops the normal term (pow(x,0) = 1 even at x = 0).
__device__ __forceinline__ float edge_weight(float dl_abs, float il, float dz_abs, float iz, float ndot, float phi_n) {
    const float d = clamp01(ndot);                                    // NaN -> 0 like saturate()
    const float ln = (phi_n == 0.0f) ? 0.0f : phi_n * hw_log2(d);
    const float wl = fmaxf(dl_abs * il, 0.0f);                        // NaN (0*inf at phi_l = 0) -> 0 like fmax() in :424
    const float e = fmaf(-kLog2e, wl + dz_abs * iz, ln);
    return hw_exp2(e);
}

constexpr int kBX = 64, kBY = 4;                                      // one wave = 64 consecutive pixels of one row

// ------------------------------------------------------------------ temporal ------------------
// Filter.cuh:359-404 + LoadPreviousData :225-258.  All previous-frame gathers are issued before the
// accept/reject tests are evaluated (one round of latency instead of the reference's chain of seven
// dependent fetches); rejected pixels simply discard them.
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void temporal_kernel(Geo g, TemporalArgs a) {
    if (a.young_list && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && threadIdx.y == 0) *a.young_count_next = 0u;
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
    const int qx = x + (int)mc.x, qy = y + (int)mc.y;                 // :232, truncation toward zero
    bool ok = qx >= 0 && qx < g.W && qy >= 0 && qy < g.H;             // :235
    const int ql = qy - g.y0;
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

    const float4 c = clamp01(Store<ST>::ld4(a.radiance, idx));        // :370 imageLoad
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
    const float4 pc = clamp01(Store<ST>::ld4(a.prev_colour, q));      // :254 imageLoad
    const int hp = a.hist_prev[q];                                    // :255
    const float2 pm = Store<ST>::ld2(a.mom_prev, q);                  // :256

    float zc, dzc, zp, dzp;
    depth_of(mc, zc, dzc);
    depth_of(mp, zp, dzp);
    ok = ok && !(fabsf(zp - zc) > a.depth_thr);                       // :242
    if (a.mesh_id_test) {                                             // :245-247 (intended test)
        const int idc = (int)unpack_h2(uc_raw.y).y, idp = (int)unpack_h2(up_raw.y).y;
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
    if (!a.sparse_colour || h < 4 || zc == kSkyZ) Store<ST>::st4(a.colour_out, idx, clamp01(o));
    Store<ST>::st2(a.mom_cur, idx, m);                                // :402
    // Frame-driver fusion: for history >= 4 FilterMoments only copies this pixel into the filter buffer
    // (:521; store(load(x)) == x in both storage types), so it is written from here and the moments launch
    // touches nothing but the history byte of such pixels (-32 B/px of traffic in steady state).
    // A young pixel whose normal is exactly (0,0,0) — the G-buffer's cleared sky texels, which never pass the normal
    // test and stay young for ever — filters to exactly (0,0,0,0) when PhiNormal > 0 (see moments_pixel): written here too.
    const bool young = h < 4;
    const bool zero_young = young && a.sky_zero && ((nc_raw.x & 0x7fff7fffu) | (nc_raw.y & 0x7fffu)) == 0u;
    if (a.passthrough_out) {
        if (!young) Store<ST>::st4(a.passthrough_out, idx, clamp01(o));
        else if (zero_young) Store<ST>::st4(a.passthrough_out, idx, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    // ... and the moments launch is told where the remaining young pixels are: their indices are appended to a list, one
    // atomic per wave that holds any (disocclusions are sparse: frame borders under a pan, silhouettes).  The list is dense in
    // young pixels, so the moments launch costs what they cost, however they are spread over the frame.
    if (a.young_list) {
        const bool listed = young && !zero_young;
        const unsigned long long m = __ballot(listed);
        const bool whole_wave = m == ~0ull;                 // all 64 pixels of the segment: flagged, not listed
        if (m != 0ull && !whole_wave) {
            const int lane = threadIdx.x, first = __builtin_ctzll(m);
            unsigned base = 0;
            if (lane == first) base = atomicAdd(a.young_count, (unsigned)__builtin_popcountll(m));
            base = __shfl(base, first);
            if (listed) a.young_list[base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)idx;
        }
        if (threadIdx.x == 0) a.young_flags[(size_t)(y - g.y0) * ((g.W + kBX - 1) / kBX) + blockIdx.x] = whole_wave ? 1 : 0;
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
    // A centre whose normal is exactly (0,0,0) — the G-buffer's cleared sky texels — has n.n' = 0 for every tap,
    // so with phi_normal > 0 every weight is exp(..)*pow(0,phi_n) = 0: the sums stay 0, sumW clamps to 1e-6 and the
    // result is exactly (0,0,0,0) (:505-516).  Same value, none of the 49 taps.
    if (((nraw.x & 0x7fff7fffu) | (nraw.y & 0x7fffu)) == 0u && a.phi_normal > 0.0f) {
        Store<ST>::st4(a.out, idx, make_float4(0.f, 0.f, 0.f, 0.f * (4.0f / h)));
        return;
    }
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
            hq[k] = a.sparse_colour ? (int)a.hist[p[k]] : 0;
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
    if (((cc.n01 & 0x7fff7fffu) | (cc.n2 & 0x7fffu)) == 0u && a.phi_normal > 0.0f) {          // see moments_pixel
        Store<ST>::st4(a.out, idx, make_float4(0.f, 0.f, 0.f, 0.f * (4.0f / h)));
        return;
    }
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

// Steady state inside the frame driver: the temporal launch already copied every pixel with history >= 4 and listed the young
// ones that need the estimate (disocclusions: sparse).  A small grid walks the list; neighbouring entries are neighbouring
// pixels (a wave of the temporal launch appends its young pixels together), so a wave's gathers stay local.
template <int ST>
__global__ __launch_bounds__(256) void moments_young_kernel(Geo g, MomentsArgs a, int list_blocks) {
    if ((int)blockIdx.x < list_blocks) {
        const unsigned n = *a.young_count;
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned)list_blocks * 256u) {
            const uint32_t p = a.young_list[i];
            const int x = (int)(p % (uint32_t)g.W), y = g.y0 + (int)(p / (uint32_t)g.W);
            if (y >= g.yb && y < g.ye) moments_pixel<ST>(g, a, x, y);         // the moments rows may be a sub-range of the temporal rows
        }
        return;
    }
    // the segments whose 64 pixels are all young (one flag each): scanned 64 flags per wave-load, interleaved over the waves so
    // that a patch of them is shared out
    const int lane = threadIdx.x & 63;
    const int nseg = (g.W + kBX - 1) / kBX;
    const int first = (g.yb - g.y0) * nseg, last = (g.ye - g.y0) * nseg;             // flag range of the launch rows
    const int nwaves = ((int)gridDim.x - list_blocks) * 4, wave = ((int)blockIdx.x - list_blocks) * 4 + (threadIdx.x >> 6);
    for (int k = 0; first + k * 64 * nwaves + wave < last; k++) {
        const int sidx = first + (k * 64 + lane) * nwaves + wave;
        unsigned long long m = __ballot(sidx < last && a.young_flags[sidx] != 0);
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int seg = first + (k * 64 + b) * nwaves + wave, x = (seg % nseg) * kBX + lane, y = g.y0 + seg / nseg;
            if (x < g.W) moments_pixel<ST>(g, a, x, y);
        }
    }
}

// ------------------------------------------------------------------ a-trous (direct) ----------
// Filter.cuh:527-624, one thread per pixel, taps gathered straight from global memory (L1/L2).
// Kept as the simple variant (SVGF_VARIANT_DIRECT) the LDS-tiled kernel is A/B-tested against.
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void atrous_direct_kernel(Geo g, AtrousArgs a) {
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    const float4 c = clamp01(Store<ST>::ld4(a.in, idx));              // :543
    float zc, dzc;
    depth_of(a.motion[idx], zc, dzc);                                 // :552
    if (zc == kSkyZ) { Store<ST>::st4(a.out, idx, c); return; }       // :554-558
    const float3 nc = normal_of(a.normal[idx]);
    const float lc = lum_exact(c.x, c.y, c.z);
    const float il = hw_rcp(a.phi_colour * sqrtf(fmaxf(0.0f, 1e-10f + c.w)));   // :562
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
            const float4 q = clamp01(Store<ST>::ld4(a.in, p));        // :586
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
#include "svgf_atrous_ws.h"
#endif
#ifndef SVGF_ROWS4
#define SVGF_ROWS4 0                // bit mask of steps (1, 2, 4, 8, 16) launched through atrous_r4_kernel (four rows per step, 8-wave workgroups):
#endif                              // a measured alternative (svgf_atrous_r4.h), not in the product build
#if SVGF_ROWS4
#include "svgf_atrous_r4.h"
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

#include "svgf_atrous_fused.h"

// ------------------------------------------------------------------ moments (LDS streaming, cold frames) ----
// Filter.cuh:430-525 for frames in which (nearly) every pixel is young (history < 4: the first three frames of a
// sequence): the 7x7 window is served from an 8-row LDS ring exactly like the à-trous kernel's 5x5 window
// (streaming down a band, 256 columns x 2 rows per step, one output per thread, rows fetched one step ahead), instead
// of 49 x 4 gathers per pixel through L1.  Records are RAW (this stage does not clamp, :450,479):
// A = {r,g,b,m1}, B = {luminance, depth, (nx,ny) halfs, nz}, C = m2.  Pixels with history >= 4 are a copy (:521).
constexpr int kMR = 3;                       // window radius (the reference's, :465)
constexpr int kMRing = kRS + 2 * kMR;        // 8 ring rows
constexpr int kMTX = 256;

__device__ __forceinline__ constexpr int len_class7(int xx, int yy) {   // |(xx,yy)|^2 in {1,2,4,5,8,9,10,13,18}
    const int l2 = xx * xx + yy * yy;
    return l2 == 1 ? 0 : l2 == 2 ? 1 : l2 == 4 ? 2 : l2 == 5 ? 3 : l2 == 8 ? 4 : l2 == 9 ? 5 : l2 == 10 ? 6 : l2 == 13 ? 7 : 8;
}

template <int ST>
__global__ __launch_bounds__(kMTX* kRS, 4) void moments_lds_kernel(Geo g, MomentsArgs a, int band_rows) {
    constexpr int TX = kMTX, WL = TX + 2 * kMR, CB = ST == 0 ? 16 : 8, MB = ST == 0 ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* recA = (f32x4*)smem;
    f32x4* recB = recA + kMRing * WL;
    float* recC = (float*)(recB + kMRing * WL);

    const int t = threadIdx.x, lane = t & 63, col = t % TX;
    const int rg = __builtin_amdgcn_readfirstlane(t / TX);
    const int wig = __builtin_amdgcn_readfirstlane((t % TX) >> 6);
    const int x0 = blockIdx.x * TX;
    const int nrows = g.ye - g.yb;
    const int j0 = blockIdx.y * band_rows;
    if (j0 >= nrows) return;
    const int j1 = min(nrows, j0 + band_rows);

    const int gx = x0 + col, oli = col + kMR;
    const bool has_halo = wig == 0 && lane < 2 * kMR;       // six halo pixels per row: lanes 0-5 of the row group's first wave
    const int hx = (lane < kMR) ? x0 - kMR + lane : x0 + TX + lane - kMR;
    const int hli = (lane < kMR) ? lane : TX + lane;
    const bool own_ok = gx < g.W, halo_ok = has_halo && hx >= 0 && hx < g.W;
    const unsigned vo = own_ok ? (unsigned)gx : 0u, vh = halo_ok ? (unsigned)hx : 0u;
    const unsigned vo_c = own_ok ? vo * CB : kOob, vo_mo = own_ok ? vo * MB : kOob, vo_m = own_ok ? vo * 16u + 8u : kOob, vo_n = own_ok ? vo * 8u : kOob, vo_h = own_ok ? vo : kOob;
    const unsigned vh_c = halo_ok ? vh * CB : kOob, vh_mo = halo_ok ? vh * MB : kOob, vh_m = halo_ok ? vh * 16u + 8u : kOob, vh_n = halo_ok ? vh * 8u : kOob;

    const unsigned npx = (unsigned)g.rows * (unsigned)g.W;
    auto mk = [](const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rs_c = mk(a.colour, npx * CB), rs_mo = mk(a.mom, npx * MB), rs_m = mk(a.motion, npx * 16u),
                                 rs_n = mk(a.normal, npx * 8u), rs_h = mk(a.hist, npx), rs_out = mk(a.out, npx * CB);
    const __amdgpu_buffer_rsrc_t rz_c = mk(a.colour, 0), rz_mo = mk(a.mom, 0), rz_m = mk(a.motion, 0), rz_n = mk(a.normal, 0);

    struct Px { u32x4 c; u32x2 mo; unsigned z; u32x2 n; };
    auto load_px = [&](Px& p, bool rok, int srow, unsigned o_c, unsigned o_mo, unsigned o_m, unsigned o_n) __attribute__((always_inline)) {
        if (rok) {
            if constexpr (ST == 0) { p.c = __builtin_amdgcn_raw_buffer_load_b128(rs_c, o_c, srow * CB, 0); p.mo = __builtin_amdgcn_raw_buffer_load_b64(rs_mo, o_mo, srow * MB, 0); }
            else { const u32x2 c2 = __builtin_amdgcn_raw_buffer_load_b64(rs_c, o_c, srow * CB, 0); p.c = (u32x4){c2.x, c2.y, 0u, 0u}; p.mo = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(rs_mo, o_mo, srow * MB, 0), 0u}; }
            p.z = __builtin_amdgcn_raw_buffer_load_b32(rs_m, o_m, srow * 16, 0);
            p.n = __builtin_amdgcn_raw_buffer_load_b64(rs_n, o_n, srow * 8, 0);
        } else {
            if constexpr (ST == 0) { p.c = __builtin_amdgcn_raw_buffer_load_b128(rz_c, o_c, 0, 0); p.mo = __builtin_amdgcn_raw_buffer_load_b64(rz_mo, o_mo, 0, 0); }
            else { const u32x2 c2 = __builtin_amdgcn_raw_buffer_load_b64(rz_c, o_c, 0, 0); p.c = (u32x4){c2.x, c2.y, 0u, 0u}; p.mo = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(rz_mo, o_mo, 0, 0), 0u}; }
            p.z = __builtin_amdgcn_raw_buffer_load_b32(rz_m, o_m, 0, 0);
            p.n = __builtin_amdgcn_raw_buffer_load_b64(rz_n, o_n, 0, 0);
        }
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
        load_px(st.h, rok, srow, vh_c, vh_mo, vh_m, vh_n);
    };
    auto commit_one = [&](const Px& p, int at) __attribute__((always_inline)) {
        float4 c; float2 m;
        if constexpr (ST == 0) { c = make_float4(__uint_as_float(p.c.x), __uint_as_float(p.c.y), __uint_as_float(p.c.z), __uint_as_float(p.c.w)); m = make_float2(__uint_as_float(p.mo.x), __uint_as_float(p.mo.y)); }
        else { const float2 lo = unpack_h2(p.c.x), hi = unpack_h2(p.c.y); c = make_float4(lo.x, lo.y, hi.x, hi.y); m = unpack_h2(p.mo.x); }
        float z = __uint_as_float(p.z);
        if (z == 0.0f) z = kSkyZ;                                                   // GetDepth, :199-207
        recA[at] = (f32x4){c.x, c.y, c.z, m.x};                                     // raw loads, :479-480
        recB[at] = (f32x4){lum_exact(c.x, c.y, c.z), z, __uint_as_float(p.n.x), unpack_h2(p.n.y).x};
        recC[at] = m.y;
    };
    auto commit = [&](int sl, const Staged& st) __attribute__((always_inline)) {
        int so = sl + rg; so = so >= kMRing ? so - kMRing : so;
        commit_one(st.o, so * WL + oli);
        if (has_halo) commit_one(st.h, so * WL + hli);
    };
    // the centre's own history byte and ddepth come straight from the planes, one step ahead (L2 hits)
    struct Centre { unsigned h; unsigned dz; };
    auto fetch_centre = [&](int j, Centre& c) __attribute__((always_inline)) {
        bool rok; const int srow = row_of(j + rg, rok);
        if (rok) { c.h = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_h, vo_h, srow, 0); c.dz = __builtin_amdgcn_raw_buffer_load_b32(rs_m, own_ok ? vo * 16u + 12u : kOob, srow * 16, 0); }
        else { c.h = 255u; c.dz = 0u; }
    };

#pragma unroll 1
    for (int r = 0; r < kMRing; r += kRS) {               // ring rows 0..7 = rows j0-3 .. j0+4
        Staged st;
        fetch(j0 - kMR + r, st);
        commit(r, st);
    }
    Centre cen, cen_next;
    fetch_centre(j0, cen);
    __syncthreads();

    const float phi_n = a.phi_normal;                      // != 0 (launcher)
    const float il = hw_rcp(a.phi_colour) * kLog2e;        // :460
    int slot0 = 0;
    for (int j = j0; j < j1; j += kRS) {
        const bool more = (j + kRS) < j1;
        Staged fs;
        if (more) { fetch(j + kRS + kMR, fs); fetch_centre(j + kRS, cen_next); }       // rows j+5, j+6 enter the ring next step

        int rowbase[2 * kMR + 1];
#pragma unroll
        for (int r = 0; r <= 2 * kMR; r++) { int sl = slot0 + rg + r; sl = sl >= kMRing ? sl - kMRing : sl; rowbase[r] = sl * WL + col; }
        const f32x4 cB = recB[rowbase[kMR] + kMR];
        const float lc = cB.x, zc = cB.y, ncz = cB.w;
        const uint32_t nc01 = __float_as_uint(cB.z);
        const float h = (float)cen.h;                                               // :442
        const float dzc = zc == kSkyZ ? 0.0f : __uint_as_float(cen.dz);
        const float izb = hw_rcp(fmaxf(dzc, 1e-8f) * 3.0f) * kLog2e;                // :461
        const float iz[9] = {izb, izb * 0.70710678118654752f, izb * 0.5f, izb * 0.44721359549995794f, izb * 0.35355339059327376f,
                             izb * 0.33333333333333333f, izb * 0.31622776601683794f, izb * 0.27735009811261456f, izb * 0.23570226039551584f};
        float sw = 0.0f, sm2 = 0.0f;
        f32x2 srg = {0.f, 0.f}, sbm = {0.f, 0.f};
        const bool zero_normal = ((nc01 & 0x7fff7fffu) == 0u) && (ncz == 0.0f);     // cleared sky texel: every weight is 0 (see moments_kernel)
        const bool need = (h < 4.0f) && !zero_normal && (j + rg < j1);
        if (__ballot(need) != 0ull) {
#pragma unroll
            for (int r = 0; r <= 2 * kMR; r++) {
                const int yy = r - kMR;
                f32x4 tA[2 * kMR + 1], tB[2 * kMR + 1];
                float tC[2 * kMR + 1];
#pragma unroll
                for (int k = 0; k <= 2 * kMR; k++) { tA[k] = recA[rowbase[r] + k]; tB[k] = recB[rowbase[r] + k]; tC[k] = recC[rowbase[r] + k]; }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int xx = -kMR; xx <= kMR; xx++) {
                    const f32x4 A = tA[xx + kMR], B = tB[xx + kMR];
                    const float d = clamp01(fmaf(B.w, ncz, dot2_h2(__float_as_uint(B.z), nc01)));
                    float e = hw_log2(d) * phi_n;
                    e = fmaf(-fabsf(B.x - lc), il, e);
                    if (xx != 0 || yy != 0) e = fmaf(-fabsf(B.y - zc), iz[len_class7(xx, yy)], e);   // phiDepth == 0 -> wZ = 0 at the centre, :420
                    const float w = hw_exp2(e);
                    sw += w;                                                         // :497-499
                    srg = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, srg);
                    sbm = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.z, A.w}, sbm);
                    sm2 = fmaf(w, tC[xx + kMR], sm2);
                }
                asm volatile("" : "+v"(sw), "+v"(srg), "+v"(sbm), "+v"(sm2) :: "memory");
            }
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f * (4.0f / h));
        if (!zero_normal) {
            sw = fmaxf(sw, 1e-6f);                                                  // :505
            const float inv = 1.0f / sw;
            const float m1 = sbm.y * inv, m2 = sm2 * inv;
            o = make_float4(srg.x * inv, srg.y * inv, sbm.x * inv, (m2 - m1 * m1) * (4.0f / h));   // :507-516
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
    constexpr size_t lds = (size_t)kMRing * WL * 36;
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(moments_lds_kernel<ST>, lds, attr_done); e != hipSuccess) return e;
    const int nrows = g.ye - g.yb, xtiles = (g.W + kMTX - 1) / kMTX;
    int nbands = 2 * num_cus() / xtiles;                  // one resident round: 2 workgroups per CU (LDS)
    if (nbands < 1) nbands = 1;
    int band = (nrows + nbands - 1) / nbands;
    if (band < 8) band = 8;
    band = (band + kRS - 1) / kRS * kRS;
    nbands = (nrows + band - 1) / band;
    moments_lds_kernel<ST><<<dim3(xtiles, nbands), dim3(kMTX * kRS), lds, s>>>(g, a, band);
    return hipGetLastError();
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

template <int ST>
__global__ __launch_bounds__(kBX* kBY) void taa_kernel(Geo g, const void* filtered, const void* history, void* out) {
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const float iw = 1.0f / (float)g.W, ih = 1.0f / (float)g.H;
    const float u = (float)x * iw, v = (float)y * ih;                    // :296
    const int sx[3] = {tex_coord(u - iw, g.W), tex_coord(u, g.W), tex_coord(u + iw, g.W)};
    const int sy[3] = {tex_coord(v - ih, g.H), tex_coord(v, g.H), tex_coord(v + ih, g.H)};
    auto at = [&](const void* img, int ix, int iy) { return clamp01(Store<ST>::ld4(img, (size_t)(sy[iy] - g.y0) * g.W + sx[ix])); };
    const float4 last = at(history, 1, 1);                                // :299
    const float mix = fminf(last.w, 0.5f);                                // :302
    const float4 c0 = at(filtered, 1, 1);                                 // :305
    float3 aa = make_float3(sqrtf(mix_exact(last.x * last.x, c0.x * c0.x, mix)), sqrtf(mix_exact(last.y * last.y, c0.y * c0.y, mix)),
                            sqrtf(mix_exact(last.z * last.z, c0.z * c0.z, mix)));   // :307-308
    float3 ya = enc_yuv(aa);                                              // :319
    // :310-317,320-335: plus-shaped and diagonal neighbourhoods
    float3 mn, mx, mnd, mxd;
    {
        const float3 y0_ = enc_yuv(make_float3(c0.x, c0.y, c0.z));
        mn = y0_; mx = y0_;
        const int px[4] = {2, 0, 1, 1}, py[4] = {1, 1, 2, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 c = at(filtered, px[k], py[k]);
            const float3 yk = enc_yuv(make_float3(c.x, c.y, c.z));
            mn = make_float3(fminf(mn.x, yk.x), fminf(mn.y, yk.y), fminf(mn.z, yk.z));
            mx = make_float3(fmaxf(mx.x, yk.x), fmaxf(mx.y, yk.y), fmaxf(mx.z, yk.z));
        }
        mnd = mn; mxd = mx;
        const int dx[4] = {2, 0, 2, 0}, dy[4] = {2, 2, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 c = at(filtered, dx[k], dy[k]);
            const float3 yk = enc_yuv(make_float3(c.x, c.y, c.z));
            mnd = make_float3(fminf(mnd.x, yk.x), fminf(mnd.y, yk.y), fminf(mnd.z, yk.z));
            mxd = make_float3(fmaxf(mxd.x, yk.x), fmaxf(mxd.y, yk.y), fmaxf(mxd.z, yk.z));
        }
    }
    mn = make_float3(mix_exact(mn.x, mnd.x, 0.5f), mix_exact(mn.y, mnd.y, 0.5f), mix_exact(mn.z, mnd.z, 0.5f));
    mx = make_float3(mix_exact(mx.x, mxd.x, 0.5f), mix_exact(mx.y, mxd.y, 0.5f), mix_exact(mx.z, mxd.z, 0.5f));
    ya = make_float3(fminf(fmaxf(ya.x, mn.x), mx.x), fminf(fmaxf(ya.y, mn.y), mx.y), fminf(fmaxf(ya.z, mn.z), mx.z));   // :338
    // :277-285; pow(x, 0.5) = sqrt(x), NaN for negative x
    float r = sqrtf((ya.x * 1.0f + ya.y * 0.0f) + ya.z * 1.13983f);
    float gg = sqrtf((ya.x * 1.0f + ya.y * -0.39465f) + ya.z * -0.58060f);
    float b = sqrtf((ya.x * 1.0f + ya.y * 2.03211f) + ya.z * 0.0f);
    if (r != r || gg != gg || b != b) { r = 0.f; gg = 0.f; b = 0.f; }     // :351
    const float4 o = make_float4(to_srgb(r), to_srgb(gg), to_srgb(b), 1.0f);   // :353
    Store<ST>::st4(out, (size_t)(y - g.y0) * g.W + x, clamp01(o));        // :355 imageStore
}

// The same stage with the neighbourhood's YUV values computed ONCE per texel: a workgroup covers 64 x 8 pixels, encodes
// the 68 x 12 filtered texels its samples can touch into LDS (the nearest-texel coordinates floor(uv*(N-1)) land one to
// three texels up-left of the pixel, depending on fp32 rounding), and every pixel takes its nine neighbours from there
// instead of nine gathers + nine YUV encodings.  Same functions on the same inputs: bit-identical to taa_kernel.
constexpr int kTaaW = 68, kTaaH = 12, kTaaRows = 8;
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void taa_lds_kernel(Geo g, const void* filtered, const void* history, void* out) {
    __shared__ float4 yuv[kTaaH][kTaaW];                                  // 16-B records: one ds_read_b128 per neighbour
    const int x0 = blockIdx.x * kBX, yb = g.yb + blockIdx.y * kTaaRows;
    auto stage = [&](int lx, int ly) {
        const int gx = x0 - 3 + lx, gy = yb - 3 + ly;
        float3 e = make_float3(0.f, 0.f, 0.f);
        if (gx >= 0 && gx < g.W && gy >= 0 && gy < g.H && gy >= g.y0 && gy < g.y0 + g.rows) {
            const float4 c = clamp01(Store<ST>::ld4(filtered, (size_t)(gy - g.y0) * g.W + gx));
            e = enc_yuv(make_float3(c.x, c.y, c.z));
        }
        yuv[ly][lx] = make_float4(e.x, e.y, e.z, 0.f);
    };
#pragma unroll
    for (int k = 0; k < kTaaH / kBY; k++) stage(threadIdx.x, threadIdx.y + kBY * k);     // columns 0..63 of the 12 rows
    {
        const int id = threadIdx.y * kBX + threadIdx.x;                                   // columns 64..67
        if (id < (kTaaW - kBX) * kTaaH) stage(kBX + (id & 3), id >> 2);
    }
    __syncthreads();
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
        const float4 last = clamp01(Store<ST>::ld4(history, ci));         // :299
        const float mix = fminf(last.w, 0.5f);                            // :302
        const float4 c0 = clamp01(Store<ST>::ld4(filtered, ci));          // :305
        float3 aa = make_float3(sqrtf(mix_exact(last.x * last.x, c0.x * c0.x, mix)), sqrtf(mix_exact(last.y * last.y, c0.y * c0.y, mix)),
                                sqrtf(mix_exact(last.z * last.z, c0.z * c0.z, mix)));   // :307-308
        float3 ya = enc_yuv(aa);                                          // :319
        float3 mn, mx, mnd, mxd;
        mn = nb(1, 1); mx = mn;
        const int px[4] = {2, 0, 1, 1}, py[4] = {1, 1, 2, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float3 yk = nb(px[k], py[k]);
            mn = make_float3(fminf(mn.x, yk.x), fminf(mn.y, yk.y), fminf(mn.z, yk.z));
            mx = make_float3(fmaxf(mx.x, yk.x), fmaxf(mx.y, yk.y), fmaxf(mx.z, yk.z));
        }
        mnd = mn; mxd = mx;
        const int dx[4] = {2, 0, 2, 0}, dy[4] = {2, 2, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float3 yk = nb(dx[k], dy[k]);
            mnd = make_float3(fminf(mnd.x, yk.x), fminf(mnd.y, yk.y), fminf(mnd.z, yk.z));
            mxd = make_float3(fmaxf(mxd.x, yk.x), fmaxf(mxd.y, yk.y), fmaxf(mxd.z, yk.z));
        }
        mn = make_float3(mix_exact(mn.x, mnd.x, 0.5f), mix_exact(mn.y, mnd.y, 0.5f), mix_exact(mn.z, mnd.z, 0.5f));
        mx = make_float3(mix_exact(mx.x, mxd.x, 0.5f), mix_exact(mx.y, mxd.y, 0.5f), mix_exact(mx.z, mxd.z, 0.5f));
        ya = make_float3(fminf(fmaxf(ya.x, mn.x), mx.x), fminf(fmaxf(ya.y, mn.y), mx.y), fminf(fmaxf(ya.z, mn.z), mx.z));   // :338
        float rr = sqrtf((ya.x * 1.0f + ya.y * 0.0f) + ya.z * 1.13983f);
        float gg = sqrtf((ya.x * 1.0f + ya.y * -0.39465f) + ya.z * -0.58060f);
        float bb = sqrtf((ya.x * 1.0f + ya.y * 2.03211f) + ya.z * 0.0f);
        if (rr != rr || gg != gg || bb != bb) { rr = 0.f; gg = 0.f; bb = 0.f; }     // :351
        const float4 o = make_float4(to_srgb(rr), to_srgb(gg), to_srgb(bb), 1.0f);  // :353
        Store<ST>::st4(out, (size_t)(y - g.y0) * g.W + x, clamp01(o));    // :355 imageStore
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
    if (a.cold_only && a.young_list) {
        const int nsegs = (g.ye - g.yb) * ((g.W + kBX - 1) / kBX);
        int scan = (nsegs + 255) / 256;                            // >= one flag per lane and load ...
        if (scan > 4 * num_cus()) scan = 4 * num_cus();            // ... on at most one resident round
        const int walk = std::min(4 * num_cus(), std::max(1, nsegs / 16));   // the list holds at most 63 pixels per segment
        if (storage == 0) moments_young_kernel<0><<<walk + scan, 256, 0, s>>>(g, a, walk);
        else moments_young_kernel<1><<<walk + scan, 256, 0, s>>>(g, a, walk);
        return hipGetLastError();
    }
    const dim3 block(kBX, kBY), grid = grid_for(g);
    if (storage == 0) moments_kernel<0><<<grid, block, 0, s>>>(g, a);
    else moments_kernel<1><<<grid, block, 0, s>>>(g, a);
    return hipGetLastError();
}

#ifdef SVGF_STAMPS
// the raw per-wave log: kStampSlots x 16 words (slot = blockIdx * 8 + wave)
extern "C" int svgf_diag_stamp_log(unsigned long long* out, unsigned long long words) {
    const unsigned long long all = (unsigned long long)kStampSlots * 16;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_log), (words < all ? words : all) * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int svgf_diag_stamps(unsigned long long* out, int reset) {
    std::vector<unsigned long long> h((size_t)kStampSlots * 16);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamp_log), h.size() * sizeof(h[0])) != hipSuccess) return -1;
    for (int i = 0; i < 16; i++) out[i] = 0;
    for (size_t k = 0; k < h.size(); k++) out[k & 15] += h[k];
#ifdef SVGF_STAMPS_PLACEMENT
    unsigned hist[32];
    if (hipMemcpyFromSymbol(hist, HIP_SYMBOL(g_simd_hist), sizeof(hist)) == hipSuccess) {
        fprintf(stderr, "[svgf stamps] waves per (wave of the workgroup, SIMD):");
        for (int w = 0; w < 8; w++) fprintf(stderr, "  w%d: %u %u %u %u", w, hist[w * 4], hist[w * 4 + 1], hist[w * 4 + 2], hist[w * 4 + 3]);
        fprintf(stderr, "\n");
    }
    if (reset) { unsigned zh[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_simd_hist), zh, sizeof(zh)); }
#endif
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamp_log)) != hipSuccess || hipMemset(p, 0, h.size() * sizeof(h[0])) != hipSuccess) return -1;
    }
    return 0;
}
#endif

hipError_t launch_atrous(const Geo& g, int storage, int variant, const AtrousArgs& a, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    const bool lds_ok = a.step == 1 || a.step == 2 || a.step == 4 || a.step == 8 || a.step == 16;
    // phi_normal == 0 (pow(x,0) = 1 even at x = 0) is left to the direct kernel: the fused exponent would see 0 * -inf
    if (variant != 1 /* SVGF_VARIANT_DIRECT */ && lds_ok && a.phi_normal != 0.0f)
        return storage == 0 ? launch_atrous_lds_step<0>(g, a, s) : launch_atrous_lds_step<1>(g, a, s);
    const dim3 block(kBX, kBY), grid = grid_for(g);
    if (storage == 0) atrous_direct_kernel<0><<<grid, block, 0, s>>>(g, a);
    else atrous_direct_kernel<1><<<grid, block, 0, s>>>(g, a);
    return hipGetLastError();
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
    const float4 o = MODE == 0 ? make_float4(c.x / dr, c.y / dg, c.z / db, c.w) : make_float4(c.x * dr, c.y * dg, c.z * db, c.w);
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
