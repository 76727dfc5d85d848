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
// glm::dot order; exact (no contraction) — used by threshold tests
__device__ __forceinline__ float dot3_exact(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float dot3_fma(float3 a, float3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// CalculateLuminance, Filter.cuh:260-263
__device__ __forceinline__ float lum_exact(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
__device__ __forceinline__ float lum_fma(float r, float g, float b) { return fmaf(0.0722f, b, fmaf(0.7152f, g, 0.2126f * r)); }
__device__ __forceinline__ float mix_exact(float x, float y, float a) { return x * (1.0f - a) + y * a; }   // glm::mix

__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Edge-stopping weight, computeWeight Filter.cuh:407-427:
//   w = exp(-max(|dl|/phi_l,0) - max(|dz|/phi_z,0)) * pow(saturate(n.n'), phi_n)
// evaluated as exp2( phi_n*log2(sat(n.n')) - (max(|dl|*il,0) + |dz|*iz)*log2(e) ) with il = 1/phi_l,
// iz = 1/phi_z precomputed per pixel.  n_scale = phi_n, or 0 with n_mask = 0 when phi_n == 0
// (pow(x,0) = 1 even at x = 0).
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
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;

    const float4 mc = a.motion_c[idx];
    const int qx = x + (int)mc.x, qy = y + (int)mc.y;                 // :232, truncation toward zero
    bool ok = qx >= 0 && qx < g.W && qy >= 0 && qy < g.H;             // :235
    const int ql = qy - g.y0;
    ok = ok && ql >= 0 && ql < g.rows;                                // strip guard: never read outside the local planes
    const size_t q = ok ? (size_t)ql * g.W + qx : idx;

    const float4 c = clamp01(Store<ST>::ld4(a.radiance, idx));        // :370 imageLoad
    const uint2 nc_raw = a.normal_c[idx];
    const uint2 uc_raw = a.uv_c[idx];
    const float4 mp = a.motion_p[q];
    const uint2 np_raw = a.normal_p[q];
    const uint2 up_raw = a.uv_p[q];
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
    Store<ST>::st4(a.colour_out, idx, clamp01(o));                    // :401 imageStore
    Store<ST>::st2(a.mom_cur, idx, m);                                // :402
}

// ------------------------------------------------------------------ moments -------------------
// Filter.cuh:430-525.  Steady state (h >= 4) is a plane copy; the (2R+1)^2 bilateral estimate runs
// only for young pixels.
template <int ST>
__global__ __launch_bounds__(kBX* kBY) void moments_kernel(Geo g, MomentsArgs a) {
    const int x = blockIdx.x * kBX + threadIdx.x;
    const int y = g.yb + blockIdx.y * kBY + threadIdx.y;
    if (x >= g.W || y >= g.ye) return;
    const size_t idx = (size_t)(y - g.y0) * g.W + x;
    const float h = (float)a.hist[idx];                               // :442
    const float4 cc = Store<ST>::ld4(a.colour, idx);                  // :450 raw load
    if (!(h < 4.0f)) { Store<ST>::st4(a.out, idx, cc); return; }      // :521

    const float lc = lum_exact(cc.x, cc.y, cc.z);
    float zc, dzc;
    depth_of(a.motion[idx], zc, dzc);
    const float3 nc = normal_of(a.normal[idx]);
    const float il = hw_rcp(a.phi_colour);                            // :460
    const float phi_d = fmaxf(dzc, 1e-8f) * 3.0f;                     // :461
    float sw = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sm1 = 0.f, sm2 = 0.f;
    const int R = a.radius;
    for (int yy = -R; yy <= R; yy++) {
        const int py = y + yy;
        if (py < 0 || py >= g.H) continue;                            // :473
        for (int xx = -R; xx <= R; xx++) {
            const int px = x + xx;
            if (px < 0 || px >= g.W) continue;
            const size_t p = (size_t)(py - g.y0) * g.W + px;
            const float4 cp = Store<ST>::ld4(a.colour, p);            // :479 raw
            const float2 mp = Store<ST>::ld2(a.mom, p);               // :480
            float zp, dzp;
            depth_of(a.motion[p], zp, dzp);                           // :482
            const float3 np = normal_of(a.normal[p]);                 // :483
            const float len = sqrtf((float)(xx * xx + yy * yy));      // :488
            const float iz = (xx == 0 && yy == 0) ? 0.0f : hw_rcp(phi_d * len);   // phiDepth == 0 -> wZ = 0, :420
            const float w = edge_weight(fabsf(lc - lum_exact(cp.x, cp.y, cp.z)), il, fabsf(zc - zp), iz, dot3_fma(nc, np), a.phi_normal);
            sw += w;                                                  // :497-499
            sr = fmaf(cp.x, w, sr); sg = fmaf(cp.y, w, sg); sb = fmaf(cp.z, w, sb);
            sm1 = fmaf(mp.x, w, sm1); sm2 = fmaf(mp.y, w, sm2);
        }
    }
    sw = fmaxf(sw, 1e-6f);                                            // :505
    const float inv = 1.0f / sw;
    sm1 *= inv; sm2 *= inv;
    const float var = (sm2 - sm1 * sm1) * (4.0f / h);                 // :511-514
    Store<ST>::st4(a.out, idx, make_float4(sr * inv, sg * inv, sb * inv, var));   // :516 unclamped
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
// Filter.cuh:527-624 re-designed for CDNA4.  For step S a pixel only ever reads pixels of its own
// row residue (y mod S), so a workgroup owns ONE residue of a band of rows and a 256-pixel-wide
// column block, and streams down the band: a ring of kRing = kR+4 decimated rows (tile + 2 S-halo
// columns each side) lives in LDS as fp32 records, every step each thread produces kR vertically
// adjacent (decimated) outputs of its column from the whole ring, while the next kR rows are already
// in flight from HBM into registers.  Global loads are always full-width row segments (16 B per
// lane, coalesced) whatever the step; the y over-fetch is (band+4)/band and the x over-fetch
// (256+4S)/256 instead of the 25x gather of a per-pixel kernel.
//
// LDS record per pixel (36 B): A = {r,g,b,variance} clamped (imageLoad :78-83),
// B = {luminance, depth (sky -> 1e30), (nx,ny) as packed halfs, nz as float}, D = ddepth.
// Pixels outside the frame are staged as {0 | 0, +inf, 0, 0}: their weight is exactly 0, which is
// what skipping the tap (:579,584) does.
constexpr int kTX = 256;                 // columns per workgroup = threads per workgroup (4 waves)
constexpr int kR = 2;                    // outputs per thread and step
constexpr int kRing = kR + 4;
constexpr int kBand = 32;                // decimated rows per workgroup
constexpr float kInf = __builtin_inff();

template <int ST> struct RawColour;
template <> struct RawColour<0> { typedef float4 type; };
template <> struct RawColour<1> { typedef uint2 type; };

template <int ST> struct RawPx {
    typename RawColour<ST>::type c;
    float2 zd;
    uint2 n;
};

template <int ST> __device__ __forceinline__ void raw_invalid(RawPx<ST>& r) {
    if constexpr (ST == 0) r.c = make_float4(0.f, 0.f, 0.f, 0.f); else r.c = make_uint2(0u, 0u);
    r.zd = make_float2(kInf, 0.f);
    r.n = make_uint2(0u, 0u);
}

template <int ST> __device__ __forceinline__ void raw_load(RawPx<ST>& r, const AtrousArgs& a, size_t idx) {
    r.c = ((const typename RawColour<ST>::type*)a.in)[idx];
    r.zd = *(const float2*)((const float*)(a.motion + idx) + 2);       // {depth, ddepth} of the motion texel
    r.n = a.normal[idx];
}

template <int ST>
__device__ __forceinline__ void commit_px(const RawPx<ST>& r, float4* recA, float4* recB, float* recD, int at) {
    float4 c;
    if constexpr (ST == 0) c = r.c;
    else { float2 lo = unpack_h2(r.c.x), hi = unpack_h2(r.c.y); c = make_float4(lo.x, lo.y, hi.x, hi.y); }
    c = clamp01(c);                                                     // imageLoad, :586
    float z = r.zd.x, dz = r.zd.y;
    if (z == 0.0f) { z = kSkyZ; dz = 0.0f; }                            // GetDepth, :199-207
    recA[at] = c;
    recB[at] = make_float4(lum_exact(c.x, c.y, c.z), z, __uint_as_float(r.n.x), unpack_h2(r.n.y).x);
    recD[at] = dz;
}

// log2 of the kernel weight K[|xx|]*K[|yy|] (:540,582), folded into the exponent
__device__ __forceinline__ constexpr float klog2(int axx, int ayy) {
    // K = {1, 2/3, 1/6} as floats, product in float like the reference; log2 tabulated offline
    // (1*2/3, 1*1/6, 2/3*2/3, 2/3*1/6, 1/6*1/6)
    return (axx + ayy == 1) ? -0.5849624872207642f      // 2/3
         : (axx == 1 && ayy == 1) ? -1.1699249744415283f // 4/9
         : (axx + ayy == 2) ? -2.5849626064300537f       // 1/6
         : (axx + ayy == 3) ? -3.1699249744415283f       // 1/9
         : -5.169925212860107f;                          // 1/36
}

template <int ST, int S>
__global__ __launch_bounds__(kTX, 2) void atrous_lds_kernel(Geo g, AtrousArgs a) {
    constexpr int WL = kTX + 4 * S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* recA = (float4*)smem;
    float4* recB = recA + kRing * WL;
    float* recD = (float*)(recB + kRing * WL);

    const int t = threadIdx.x;
    const int x0 = blockIdx.x * kTX;
    const int rv = blockIdx.y % S;                 // row residue (relative to g.yb) this workgroup owns
    const int band = blockIdx.y / S;
    const int nrows = g.ye - g.yb;
    const int nj = (nrows - rv + S - 1) / S;       // decimated rows of this residue
    const int j0 = band * kBand;
    if (j0 >= nj) return;
    const int j1 = min(nj, j0 + kBand);
    const int ybase = g.yb + rv;                   // global row of decimated index j: ybase + S*j

    const int gx = x0 + t;                         // own column
    const bool has_halo = t < 4 * S;
    const int hx = (t < 2 * S) ? x0 - 2 * S + t : x0 + kTX + t - 2 * S;
    const int hli = (t < 2 * S) ? t : kTX + t;     // LDS column of the halo pixel
    const int oli = t + 2 * S;                     // LDS column of the own pixel

    auto fetch = [&](int j, RawPx<ST>& own, RawPx<ST>& halo) {
        const int y = ybase + S * j;
        const int yl = y - g.y0;
        const bool rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
        const size_t rowoff = (size_t)yl * g.W;
        if (rok && gx < g.W) raw_load<ST>(own, a, rowoff + gx); else raw_invalid<ST>(own);
        if (has_halo) {
            if (rok && hx >= 0 && hx < g.W) raw_load<ST>(halo, a, rowoff + hx); else raw_invalid<ST>(halo);
        }
    };
    auto commit = [&](int slot, const RawPx<ST>& own, const RawPx<ST>& halo) {
        commit_px<ST>(own, recA, recB, recD, slot * WL + oli);
        if (has_halo) commit_px<ST>(halo, recA, recB, recD, slot * WL + hli);
    };

    // prologue: ring rows 0..kRing-1 = decimated rows j0-2 .. j0+kR+1
#pragma unroll 1
    for (int r = 0; r < kRing; r += kR) {          // kR rows in flight at a time keeps the prologue's registers small
        RawPx<ST> o[kR], h[kR];
#pragma unroll
        for (int i = 0; i < kR; i++) fetch(j0 - 2 + r + i, o[i], h[i]);
#pragma unroll
        for (int i = 0; i < kR; i++) commit(r + i, o[i], h[i]);
    }
    __syncthreads();

    const float phi_n = a.phi_normal;
    int slot0 = 0;
    for (int j = j0; j < j1; j += kR) {
        const bool more = (j + kR) < j1;
        RawPx<ST> po[kR], ph[kR];
        if (more) {
#pragma unroll
            for (int i = 0; i < kR; i++) fetch(j + kR + 2 + i, po[i], ph[i]);
        }

        int rowbase[kRing];
#pragma unroll
        for (int r = 0; r < kRing; r++) { int sl = slot0 + r; sl = sl >= kRing ? sl - kRing : sl; rowbase[r] = sl * WL + oli; }

        // centres
        float4 cc[kR]; float lc[kR], zc[kR], ncz[kR], il[kR], iz[kR][5], sw[kR], sr[kR], sg[kR], sb[kR], sv[kR];
        half2_t nc01[kR];
#pragma unroll
        for (int i = 0; i < kR; i++) {
            const float4 A = recA[rowbase[2 + i]], B = recB[rowbase[2 + i]];
            const float dz = recD[rowbase[2 + i]];
            cc[i] = A; lc[i] = B.x; zc[i] = B.y; nc01[i] = __builtin_bit_cast(half2_t, __float_as_uint(B.z)); ncz[i] = B.w;
            const float phi_l = a.phi_colour * sqrtf(fmaxf(0.0f, 1e-10f + A.w));          // :562
            il[i] = fminf(hw_rcp(phi_l), 1e30f) * kLog2e;
            const float izb = hw_rcp(fmaxf(dz, 1e-6f) * (float)S) * kLog2e;               // :563
            iz[i][0] = izb;                              // |(xx,yy)| = 1
            iz[i][1] = izb * 0.70710678118654752f;       // sqrt 2
            iz[i][2] = izb * 0.5f;                       // 2
            iz[i][3] = izb * 0.44721359549995794f;       // sqrt 5
            iz[i][4] = izb * 0.35355339059327376f;       // 2 sqrt 2
            sw[i] = 1.0f; sr[i] = A.x; sg[i] = A.y; sb[i] = A.z; sv[i] = A.w;             // :567-568
        }

        // Software pipeline over the ring rows: the 5 taps (10 x ds_read_b128) of row r+1 are issued before
        // row r is consumed.  The empty asm statements pin that order: left alone, instruction selection
        // sinks all arithmetic below all 60 LDS reads of the unrolled step (256 VGPRs + scratch spills).
        float4 tapA[2][5], tapB[2][5];
        auto load_row = [&](int r, int buf) {
#pragma unroll
            for (int xx = -2; xx <= 2; xx++) { tapA[buf][xx + 2] = recA[rowbase[r] + xx * S]; tapB[buf][xx + 2] = recB[rowbase[r] + xx * S]; }
        };
        load_row(0, 0);
#pragma unroll
        for (int r = 0; r < kRing; r++) {
            const int buf = r & 1;
            if (r + 1 < kRing) load_row(r + 1, buf ^ 1);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int xx = -2; xx <= 2; xx++) {
                const float4 A = tapA[buf][xx + 2], B = tapB[buf][xx + 2];
                const half2_t n01 = __builtin_bit_cast(half2_t, __float_as_uint(B.z));
#pragma unroll
                for (int i = 0; i < kR; i++) {
                    const int yy = r - 2 - i;
                    if (yy < -2 || yy > 2 || (xx == 0 && yy == 0)) continue;              // compile-time
                    const int axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                    const int l2 = axx * axx + ayy * ayy;
                    const int cls = l2 == 1 ? 0 : l2 == 2 ? 1 : l2 == 4 ? 2 : l2 == 5 ? 3 : 4;
                    float e = klog2(axx, ayy);
                    const float d = clamp01(fmaf(B.w, ncz[i], __builtin_amdgcn_fdot2(n01, nc01[i], 0.0f, false)));
                    e = fmaf(hw_log2(d), phi_n, e);                                   // phi_n != 0 here (launcher)
                    e = fmaf(-fabsf(B.x - lc[i]), il[i], e);
                    e = fmaf(-fabsf(B.y - zc[i]), iz[i][cls], e);
                    const float w = hw_exp2(e);
                    sw[i] += w;                                                           // :607
                    sr[i] = fmaf(w, A.x, sr[i]); sg[i] = fmaf(w, A.y, sg[i]); sb[i] = fmaf(w, A.z, sb[i]);
                    sv[i] = fmaf(w * w, A.w, sv[i]);                                      // :608
                }
            }
#pragma unroll
            for (int i = 0; i < kR; i++)
                asm volatile("" : "+v"(sw[i]), "+v"(sr[i]), "+v"(sg[i]), "+v"(sb[i]), "+v"(sv[i]) :: "memory");
        }

#pragma unroll
        for (int i = 0; i < kR; i++) {
            const int jj = j + i;
            if (jj < j1 && gx < g.W) {
                const size_t idx = (size_t)(ybase + S * jj - g.y0) * g.W + gx;
                if (zc[i] == kSkyZ) {
                    Store<ST>::st4(a.out, idx, cc[i]);                                    // :554-558
                } else {
                    const float inv = 1.0f / sw[i];
                    const float4 o = make_float4(sr[i] * inv, sg[i] * inv, sb[i] * inv, sv[i] * (inv * inv));   // :615
                    Store<ST>::st4(a.out, idx, o);
                    if (a.feedback) Store<ST>::st4(a.feedback, idx, o);                   // :619-622
                }
            }
        }

        if (more) {
            __syncthreads();                       // every wave is done reading the two oldest ring rows
#pragma unroll
            for (int i = 0; i < kR; i++) { int sl = slot0 + i; sl = sl >= kRing ? sl - kRing : sl; commit(sl, po[i], ph[i]); }
            slot0 += kR; if (slot0 >= kRing) slot0 -= kRing;
            __syncthreads();
        }
    }
}

template <int ST, int S>
hipError_t launch_atrous_lds(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    constexpr int WL = kTX + 4 * S;
    constexpr size_t lds = (size_t)kRing * WL * 36;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)atrous_lds_kernel<ST, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nrows = g.ye - g.yb;
    const int njmax = (nrows + S - 1) / S;
    const dim3 grid((g.W + kTX - 1) / kTX, S * ((njmax + kBand - 1) / kBand));
    atrous_lds_kernel<ST, S><<<grid, dim3(kTX), lds, s>>>(g, a);
    return hipGetLastError();
}

template <int ST>
hipError_t launch_atrous_lds_step(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    switch (a.step) {
        case 1: return launch_atrous_lds<ST, 1>(g, a, s);
        case 2: return launch_atrous_lds<ST, 2>(g, a, s);
        case 4: return launch_atrous_lds<ST, 4>(g, a, s);
        case 8: return launch_atrous_lds<ST, 8>(g, a, s);
        case 16: return launch_atrous_lds<ST, 16>(g, a, s);
        default: return hipErrorInvalidValue;
    }
}

inline dim3 grid_for(const Geo& g) { return dim3((g.W + kBX - 1) / kBX, (g.ye - g.yb + kBY - 1) / kBY); }

}  // namespace

hipError_t launch_temporal(const Geo& g, int storage, const TemporalArgs& a, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    const dim3 block(kBX, kBY), grid = grid_for(g);
    if (storage == 0) temporal_kernel<0><<<grid, block, 0, s>>>(g, a);
    else temporal_kernel<1><<<grid, block, 0, s>>>(g, a);
    return hipGetLastError();
}

hipError_t launch_moments(const Geo& g, int storage, const MomentsArgs& a, hipStream_t s) {
    if (g.ye <= g.yb) return hipSuccess;
    const dim3 block(kBX, kBY), grid = grid_for(g);
    if (storage == 0) moments_kernel<0><<<grid, block, 0, s>>>(g, a);
    else moments_kernel<1><<<grid, block, 0, s>>>(g, a);
    return hipGetLastError();
}

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

}  // namespace svgf
