// svgf_device.h — device-side helpers shared by the gfx950 kernels of svgf_kernels.hip: storage traits (fp32 / fp16 planes), the
// reference's load / store conventions (Filter.cuh:55-83,199-207), the fused-exponent edge-stopping weight, and the staging of
// G-buffer / colour texels into LDS records for the streaming kernels.
#pragma once
#include "svgf_kernels.h"

#include <atomic>

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

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }          // NaN -> 0: saturate() of a dot product
__device__ __forceinline__ float4 clamp01(float4 v) { return make_float4(clamp01(v.x), clamp01(v.y), clamp01(v.z), clamp01(v.w)); }
// The reference's value clamp of imageLoad / imageStore (Filter.cuh:63-69,78-83) is glm::clamp = min(max(x, 0), 1) built from
// (x < y) ? y : x comparisons: +-inf clamp to 1 / 0 and a NaN texel STAYS NaN (it then poisons the history through mix, :398, and the
// a-trous sums channel by channel).  fminf / fmaxf, v_med3 and the hardware's result clamp in its default mode all turn a NaN into 0.
// The kernels that clamp texels therefore clear MODE.DX10_CLAMP for the life of their waves (keep_nan_in_clamps, first statement):
// with it cleared the `clamp` result modifier still maps +-inf and every finite value into [0, 1] and passes a NaN through — the
// reference's clamp in ONE instruction (two values per instruction in its packed form), with nothing to detect and nothing to put back.
// (MODE is per-wave state; nothing the compiler generates for these kernels depends on the bit except `saturate(n.n')` of the taps,
// where a NaN can only come from a non-finite normal.)
__device__ __forceinline__ void keep_nan_in_clamps() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 8, 1), 0" ::: "memory"); }
__device__ __forceinline__ float clamp01_hw(float v) { float r; asm("v_max_f32 %0, %1, %1 clamp" : "=v"(r) : "v"(v)); return r; }   // needs keep_nan_in_clamps()
// ... except for the sign of a zero: `(x < 0) ? 0 : x` and `(1 < x) ? 1 : x` hand -0.0 through, the result clamp returns +0.0.  The
// per-pixel kernels (temporal, direct a-trous, TAA: none of them bound by vector issue) put the input back where it compares equal to
// zero (one v_cmp + one v_cndmask per channel; a NaN compares unequal and keeps the clamp's result, itself).  The streaming kernels
// keep clamp01_pk for the texels they stage and repair a -0.0 where it can reach memory (commit_px below, atrous_band).
__device__ __forceinline__ float clamp01_ref(float v) { const float r = clamp01_hw(v); return v == 0.0f ? v : r; }
__device__ __forceinline__ float4 clamp01_ref(float4 v) { return make_float4(clamp01_ref(v.x), clamp01_ref(v.y), clamp01_ref(v.z), clamp01_ref(v.w)); }

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
// int(x) / ivec2(vec2) of CUDA (Filter.cuh:232: `Coord + ivec2(MotionVector)`; the instance ID of :245-246) is cvt.rzi.s32.f32: truncation toward
// zero, out-of-range values SATURATE to INT_MIN / INT_MAX and a NaN converts to 0.  v_cvt_i32_f32 does exactly that — but a C++ cast of an
// out-of-range float is undefined behaviour, which the compiler may exploit; the instruction is therefore written out.  The sum with the pixel
// coordinate wraps (two's complement): every motion beyond +-2^31 pixels lands outside the frame and is rejected (:235); a NaN motion is 0.
__device__ __forceinline__ int cvt_rzi_sat(float f) { int r; asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f)); return r; }
__device__ __forceinline__ int add_wrap(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
// glm::dot order; exact (no contraction) — used by threshold tests
__device__ __forceinline__ float dot3_exact(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float dot3_fma(float3 a, float3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// CalculateLuminance, Filter.cuh:260-263, in the reference's operation order: where the temporal variance is 0 the
// a-trous weights amplify a one-ulp luminance difference ~1e4 times (Tolerance), so no FMA here
__device__ __forceinline__ float lum_exact(float r, float g, float b) {
    float pr = 0.2126f * r, pg = 0.7152f * g;
    asm("" : "+v"(pr));      // (keeps hipcc from pairing two of the three products into one v_pk_mul_f32 behind two v_mov: 6 issue slots instead of 2)
    asm("" : "+v"(pg));
    return pr + pg + 0.0722f * b;
}
__device__ __forceinline__ float mix_exact(float x, float y, float a) { return x * (1.0f - a) + y * a; }   // glm::mix

__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// log2(e) / phi_l of the a-trous filter, phi_l = PhiColour * sqrt(max(0, 1e-10 + variance)) (Filter.cuh:562), as ONE v_rsq_f32 times the launch's
// log2(e) / PhiColour (`k`); capped like the reciprocal of a zero phi_l (PhiColour = 0: the luminance term is -inf, or NaN -> fmax(.., 0) in :424).
// The variance is a clamped texel (imageLoad: in [0,1]), so the max(0, .) of :562 is the identity.  (sqrtf() compiles to a 17-instruction
// correctly rounded sequence whose result only ever fed an approximate reciprocal: -2.5 % per launch, profiles/r03_small_experiments.txt.)
__device__ __forceinline__ float inv_phi_l_log2e(float variance01, float k) { return fminf(hw_rsq(1e-10f + variance01) * k, 1.4426950e30f); }

// Edge-stopping weight, computeWeight Filter.cuh:407-427:
//   w = exp(-max(|dl|/phi_l,0) - max(|dz|/phi_z,0)) * pow(saturate(n.n'), phi_n)
// evaluated as exp2( phi_n*log2(sat(n.n')) - (max(|dl|*il,0) + |dz|*iz)*log2(e) ) with il = 1/phi_l,
// iz = 1/phi_z precomputed per pixel; phi_n == 0 drops the normal term (pow(x,0) = 1 even at x = 0).
__device__ __forceinline__ float edge_weight(float dl_abs, float il, float dz_abs, float iz, float ndot, float phi_n) {
    const float d = clamp01(ndot != ndot ? 0.0f : ndot);              // saturate(): NaN -> 0 (explicitly: some callers run with keep_nan_in_clamps())
    const float ln = (phi_n == 0.0f) ? 0.0f : phi_n * hw_log2(d);
    const float wl = fmaxf(dl_abs * il, 0.0f);                        // NaN (0*inf at phi_l = 0) -> 0 like fmax() in :424
    const float wz = fmaxf(dz_abs * iz, 0.0f);                        // NaN (a NaN depth in the G-buffer, inf - inf) -> 0 likewise
    const float e = fmaf(-kLog2e, wl + wz, ln);
    return hw_exp2(e);
}

constexpr int kBX = 64, kBY = 4;                                      // one wave = 64 consecutive pixels of one row

// ------------------------------------------------------------------ LDS streaming: records and staging ----------
// A streaming kernel (atrous_lds_kernel, atrous_fused12_kernel) keeps a ring of rows in LDS as fp32 records, 32 B per pixel in
// three planes: A = {r,g,b,variance} clamped (imageLoad :78-83), L = {luminance, depth (sky -> 1e30)}, N = {(nx,ny) as packed
// halfs, nz as float}.  Planes are addressed as buffer resources: a per-lane constant byte offset (voffset, kOob for a column
// outside the frame) + a per-row scalar offset (soffset); a row outside the frame is read through a zero-length resource.
// Out-of-range texels come back all-zero: depth 0 = sky sentinel and a zero normal give weight exactly 0 — what skipping the
// tap (:579,584) does.
constexpr int kRS = 2;                   // rows produced per step
constexpr int kRing = kRS + 4;           // ring rows of a 5-row window
constexpr int kXcds = 8;                 // MI355X: 8 accelerator dies, workgroup id i is dispatched to XCD i % 8
constexpr int kRecBytes = 32;            // LDS bytes per staged pixel
constexpr unsigned kOob = 0xFFFFFF00u;   // byte offset no plane reaches (planes are < 4 GiB)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;   // explicitly in LDS (a volatile access through a generic pointer would be a flat load)

// One staged pixel as it comes off the planes: colour (16 B fp32 / 8 B fp16), {depth, ddepth} (ddepth only for pixels of the
// thread's own column: DZ), normal.
template <int ST, bool DZ> struct RawPx;
template <> struct RawPx<0, true> { u32x4 c; u32x2 zd; u32x2 n; };
template <> struct RawPx<1, true> { u32x2 c; u32x2 zd; u32x2 n; };
template <> struct RawPx<0, false> { u32x4 c; unsigned zd; u32x2 n; };
template <> struct RawPx<1, false> { u32x2 c; unsigned zd; u32x2 n; };

struct PlaneRsrc { __amdgpu_buffer_rsrc_t colour, motion, normal; };

// Depth / normal source of a launch: the G-buffer's motion plane (16-B texels, {depth, ddepth} at +8) and normal plane (8-B
// texels), or the frame's guide plane (16-B texels: {depth, ddepth} at +0, normal at +8): the same two loads either way.
struct GuideSel {
    unsigned m_off, n_off, n_shift;
    __device__ __forceinline__ explicit GuideSel(bool guided) : m_off(guided ? 0u : 8u), n_off(guided ? 8u : 0u), n_shift(guided ? 4u : 3u) {}
};
__device__ __forceinline__ PlaneRsrc plane_rsrc(const AtrousArgs& a, unsigned npx, int colour_bytes, unsigned n_shift, bool row_ok) {
    const bool guided = a.guide != nullptr;
    PlaneRsrc r;
    r.colour = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, row_ok ? (int)(npx * colour_bytes) : 0, 0x00020000);
    r.motion = __builtin_amdgcn_make_buffer_rsrc(guided ? (void*)a.guide : (void*)a.motion, 0, row_ok ? (int)(npx * 16u) : 0, 0x00020000);
    r.normal = __builtin_amdgcn_make_buffer_rsrc(guided ? (void*)a.guide : (void*)a.normal, 0, row_ok ? (int)(npx << n_shift) : 0, 0x00020000);
    return r;
}

// voff_*: the lane's constant byte offsets into the colour, depth and normal planes (voff_m WITHOUT the position of {depth, ddepth}
// inside the 16-byte texel, m_off: it travels in the scalar offset, so that with fp32 storage voff_m IS voff_c — one register less in
// a loop that has none to spare); srow: the row's scalar element offset yl*W.
template <int ST, bool DZ>
__device__ __forceinline__ void raw_load(RawPx<ST, DZ>& r, const PlaneRsrc& rs, unsigned voff_c, unsigned voff_m, unsigned voff_n, int srow, unsigned n_shift, unsigned m_off) {
    constexpr int cb = ST == 0 ? 16 : 8;
    if constexpr (ST == 0) r.c = __builtin_amdgcn_raw_buffer_load_b128(rs.colour, voff_c, srow * cb, 0);
    else r.c = __builtin_amdgcn_raw_buffer_load_b64(rs.colour, voff_c, srow * cb, 0);
    if constexpr (DZ) r.zd = __builtin_amdgcn_raw_buffer_load_b64(rs.motion, voff_m, srow * 16 + (int)m_off, 0);      // {depth, ddepth}
    else r.zd = __builtin_amdgcn_raw_buffer_load_b32(rs.motion, voff_m, srow * 16 + (int)m_off, 0);                   // depth
    r.n = __builtin_amdgcn_raw_buffer_load_b64(rs.normal, voff_n, srow << n_shift, 0);
}

// "does any lane of the wave hold `pred`": HIP's __ballot() compares a materialised 0/1 (v_cndmask + v_cmp per call); the builtin folds
// into the compare that produced the predicate.
__device__ __forceinline__ bool wave_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }

// Raw barrier: __syncthreads() would also wait for vmcnt(0), i.e. for the rows a step has just requested.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float med01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }   // = min(max(v,0),1) for non-NaN v
// the reference's clamp (NaN kept: the kernel has called keep_nan_in_clamps()) for two values with ONE instruction: x * 1.0 with the result
// clamp of a packed multiply
__device__ __forceinline__ f32x2 clamp01_pk(f32x2 v) {
    f32x2 r;
    const f32x2 one = {1.0f, 1.0f};
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(v), "s"(one));
    return r;
}

// What a wave reports of a ring row it staged (the flag words of the streaming kernels): a surface texel's normal differs from the
// workgroup's reference normal — no uniform-normal path while the row is in the ring.
constexpr uint32_t kFlagNormal = 1u;

// Convert a staged pixel into its LDS records (the kernel runs with keep_nan_in_clamps(): a NaN channel stays NaN in the colour record
// and makes the luminance NaN).  -> the LANE MASK of "a surface texel whose normal differs from the reference normal (ref01, refz)".
// A mask, not a per-lane bool: the ballot builtin folds into ONE compare, but a predicate that is an and / or of compares gets
// materialised as 0 / 1 in a vector register and compared again — so every compare is balloted by itself and the masks are combined on
// the scalar unit.  `store` masks the LDS writes only: everything else runs on every lane (the halo pass's idle lanes hold all-zero
// texels: no depth, so they never differ).
//
// Sign of zero.  The reference's clamp keeps a -0.0 channel (clamp01_ref above); clamp01_pk — two channels per instruction — returns +0.0.
// Where can the difference reach memory?  A SKY centre is copied (:554-558), bit for bit.  A filtered pixel's sums start from the centre's
// channel and add weight x tap (:567-568,604-608): the result is -0.0 iff the centre AND every tap inside the frame hold -0.0 in that channel,
// and +0.0 or positive otherwise.  Either way only a pixel whose OWN texel holds a -0.0 channel can come out -0.0.  So the product pass
// (EXACT = false) only LOOKS: *negzero receives the lanes whose own texel holds one (v_min3_i32 + v_min_i32 + one compare: INT_MIN is
// the bit pattern of -0.0 and the smallest integer), the band is run again like one that produced a NaN, and the second pass (EXACT = true)
// stages with the sign-keeping clamp, gives the texels OUTSIDE the frame the colour -0.0 — the identity of the sums: their weight is exactly
// +0 (no depth), and fma(+0, -0.0, s) = s whatever s is, which is what skipping the tap does (:579,584) — and stores those pixels again
// (atrous_band).  No frame without a -0.0 texel gets there.
__device__ __forceinline__ unsigned long long lanes_where(bool single_compare) { return __builtin_amdgcn_ballot_w64(single_compare); }
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
__device__ __forceinline__ bool has_negzero(float4 c) {
    const int m = min(min(__float_as_int(c.x), __float_as_int(c.y)), min(__float_as_int(c.z), __float_as_int(c.w)));
    return m == (int)0x80000000;
}
// addr_a / addr_l: LDS byte addresses of the pixel's colour and {luminance, depth} records; its normal record sits noff bytes behind the latter.
template <int ST, bool DZ, bool EXACT = false>
__device__ __forceinline__ unsigned long long commit_px(const RawPx<ST, DZ>& r, uint32_t addr_a, uint32_t addr_l, int noff, uint32_t ref01, uint32_t refz, bool store = true,
                                                        unsigned long long* negzero = nullptr, bool outside = false) {
    float4 c;
    if constexpr (ST == 0) c = make_float4(__uint_as_float(r.c.x), __uint_as_float(r.c.y), __uint_as_float(r.c.z), __uint_as_float(r.c.w));
    else { float2 lo = unpack_h2(r.c.x), hi = unpack_h2(r.c.y); c = make_float4(lo.x, lo.y, hi.x, hi.y); }
    const float4 raw = c;
    const f32x2 c01 = clamp01_pk((f32x2){c.x, c.y}), c23 = clamp01_pk((f32x2){c.z, c.w});    // imageLoad, :78-83,586
    c = make_float4(c01.x, c01.y, c23.x, c23.y);
    if constexpr (EXACT) {
        c = make_float4(raw.x == 0.0f ? raw.x : c.x, raw.y == 0.0f ? raw.y : c.y, raw.z == 0.0f ? raw.z : c.z, raw.w == 0.0f ? raw.w : c.w);
        if (outside) c = make_float4(-0.0f, -0.0f, -0.0f, -0.0f);
    }
    float z;
    if constexpr (DZ) z = __uint_as_float(r.zd.x); else z = __uint_as_float(r.zd);
    if (z == 0.0f) z = kSkyZ;                                           // GetDepth, :199-207
    const float lum = lum_exact(c.x, c.y, c.z), nz = unpack_h2(r.n.y).x;
    if (store) {
        *(lds_f32x4*)(uintptr_t)addr_a = (f32x4){c.x, c.y, c.z, c.w};
        *(lds_f32x2*)(uintptr_t)addr_l = (f32x2){lum, z};
        *(lds_f32x2*)(uintptr_t)(addr_l + noff) = (f32x2){__uint_as_float(r.n.x), nz};
    }
    // A texel without depth (sky, or outside the frame) whose normal is all-zero — a cleared texel — does not count: its weight is 0 through the
    // depth term in the uniform form and through n.n' = 0 in the general one (and where the centre's depth is a NaN both forms turn NaN and the
    // exact form decides).  A texel without depth that DOES hold a normal counts like a surface texel: the general taps read that normal — a NaN
    // in it makes them NaN and sends the pixel to the exact form — while the uniform taps never look at it, so leaving it out would make the
    // rounding of the pixels around it depend on what else the workgroup's tile holds, i.e. on how strips and row ranges cut the frame.
    const uint32_t nzb = r.n.y & 0xffffu;
    const unsigned long long surface = lanes_where(z != kSkyZ);
    if constexpr (!EXACT) { if (negzero) *negzero = lanes_where(has_negzero(raw)); }
    return (surface | lanes_where((r.n.x | nzb) != 0u)) & (lanes_where(r.n.x != ref01) | lanes_where(nzb != refz));
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

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
// (nx,ny).(nx',ny') of two packed-half pairs: exact products, one rounding of their sum (v_dot2_f32_f16 with a zero addend; the
// builtin would pick the accumulating v_dot2c form and spend a v_mov on the zero).  hipcc does not look inside asm statements,
// so the three wait states a non-dot VALU needs before it may read (or overwrite) a dot result on gfx940+ are part of the
// statement; with 4 waves per SIMD they cost no VALU issue.
__device__ __forceinline__ float dot2_h2(uint32_t a, uint32_t b) {
    float d;
    asm("v_dot2_f32_f16 %0, %1, %2, 0\n\ts_nop 2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// XCD-aware tile order of the streaming launches.  The dispatcher deals consecutive workgroup ids to the 8 XCDs in turn, and
// each XCD has its own L2: with a plain (x, y) grid two tiles that share a halo always sit on different XCDs and every halo
// texel comes from memory twice.  Tiles are numbered with x fastest, cut into groups of `xgroup` consecutive tiles, and group k
// goes to XCD k % 8 (rotated by `xrot` per round, so that an XCD's groups come from different parts of the frame): neighbours
// inside a group run on one XCD at about the same time and share their halos in its L2.  -> tile number, or >= ntiles (padding).
__device__ __forceinline__ int xcd_tile_of(int bid, int xgroup, int xrot) {      // bid: the workgroup's index (a multiple of 8 may have been taken off)
    const int wid = bid >> 3;                      // index among the workgroups of this XCD
    const int round = wid / xgroup;                // the XCD's round-th group
    return (round * kXcds + ((bid + xrot * round) & (kXcds - 1))) * xgroup + wid % xgroup;
}
__device__ __forceinline__ int xcd_tile(int xgroup, int xrot) { return xcd_tile_of((int)blockIdx.x, xgroup, xrot); }
inline dim3 xcd_grid(int ntiles, int xm, int& xgroup) {      // xm groups per XCD
    xgroup = (ntiles + kXcds * xm - 1) / (kXcds * xm);
    if (xgroup < 1) xgroup = 1;
    const int ngroups = (ntiles + xgroup - 1) / xgroup;
    return dim3((unsigned)((ngroups + kXcds - 1) / kXcds) * kXcds * xgroup);
}

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

}  // namespace
}  // namespace svgf
