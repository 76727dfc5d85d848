// Micro-benchmark 2: the kernel's row structure — per ring row NRD ds_read_b128 (immediate offsets) then 5 taps of
// 13 VALU — at W waves per SIMD.  Isolates what LDS reads cost the VALU stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int NRD, bool KR2, bool DB>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, unsigned long long* clk) {
    extern __shared__ f32x4 lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 256) lds[i] = (f32x4){in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]};
    __syncthreads();
    float lc = in[t & 1023], zc = in[(t + 7) & 1023], ncz = in[(t + 3) & 1023], il = 3.0f, iz = 2.0f, phi = 128.0f;
    half2_t nc = __builtin_bit_cast(half2_t, in[(t + 11) & 1023]);
    float sw = 1.0f; f32x2 srg = {0.1f, 0.2f}, sbv = {0.3f, 0.4f};
    float lc2 = lc * 1.1f, zc2 = zc * 0.9f, ncz2 = ncz * 1.05f, il2 = 2.5f, iz2 = 1.5f; half2_t nc2 = __builtin_bit_cast(half2_t, in[(t + 13) & 1023]);
    float sw2 = 1.0f; f32x2 srg2 = {0.1f, 0.2f}, sbv2 = {0.3f, 0.4f};
    int base = t;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    f32x4 tA[2][5], tB[2][5];
    auto load_row = [&](int buf, int it) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 5; c++) {
            tA[buf][c] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f} + (float)it; tB[buf][c] = (f32x4){0.5f, 0.6f, 0.7f, 0.8f} * (float)it;
            if (2 * c < NRD) tB[buf][c] = lds[(base + c * 16 + 512) & 1023];
            if (2 * c + 1 < NRD) tA[buf][c] = lds[(base + c * 16) & 1023];
        }
        base = (base + 320) & 511;
    };
    auto compute_row = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const f32x4 A = tA[buf][c], B = tB[buf][c];
            float d = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, B.z), nc, 0.0f, false);
            d = fminf(fmaxf(fmaf(B.w, ncz, d), 0.f), 1.f);
            float e = fmaf(__builtin_amdgcn_logf(d), phi, -0.58f);
            e = fmaf(-fabsf(B.x - lc), il, e);
            e = fmaf(-fabsf(B.y - zc), iz, e);
            float w = __builtin_amdgcn_exp2f(e);
            f32x2 ww = {w, w * w};
            sw += w;
            srg = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, srg);
            sbv = __builtin_elementwise_fma(ww, (f32x2){A.z, A.w}, sbv);
            if (KR2) {
                float d2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, B.z), nc2, 0.0f, false);
                d2 = fminf(fmaxf(fmaf(B.w, ncz2, d2), 0.f), 1.f);
                float e2 = fmaf(__builtin_amdgcn_logf(d2), phi, -0.58f);
                e2 = fmaf(-fabsf(B.x - lc2), il2, e2);
                e2 = fmaf(-fabsf(B.y - zc2), iz2, e2);
                float w2 = __builtin_amdgcn_exp2f(e2);
                f32x2 ww2 = {w2, w2 * w2};
                sw2 += w2;
                srg2 = __builtin_elementwise_fma((f32x2){w2, w2}, (f32x2){A.x, A.y}, srg2);
                sbv2 = __builtin_elementwise_fma(ww2, (f32x2){A.z, A.w}, sbv2);
            }
        }
        asm volatile("" : "+v"(sw), "+v"(srg), "+v"(sbv), "+v"(sw2), "+v"(srg2), "+v"(sbv2) :: "memory");
    };
    if (DB) {
        load_row(0, 0);
        for (int it = 0; it < iters; it += 2) {
            load_row(1, it + 1); asm volatile("" ::: "memory"); compute_row(0);
            load_row(0, it + 2); asm volatile("" ::: "memory"); compute_row(1);
        }
    } else {
        for (int it = 0; it < iters; it++) { load_row(0, it); asm volatile("" ::: "memory"); compute_row(0); }
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 7 && t == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    out[blockIdx.x * 256 + t] = sw + srg.x + srg.y + sbv.x + sbv.y + sw2 + srg2.x + srg2.y + sbv2.x + sbv2.y;
}

template <int NRD, bool KR2, bool DB> void run(float* d, float* in, int wps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, grid = 256 * wps;
    hipFuncSetAttribute((const void*)k<NRD, KR2, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
    static unsigned long long* clk = nullptr; if (!clk) hipMalloc(&clk, 16);
    k<NRD, KR2, DB><<<grid, 256, 16384>>>(d, in, 8, clk);
    hipEventRecord(e0);
    k<NRD, KR2, DB><<<grid, 256, 16384>>>(d, in, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("   shader clock: %.0f MHz (s_memtime ticks %llu over %llu x 10 ns)\n", (double)h[0] / ((double)h[1] * 0.01), h[0], h[1]);
    double taps_per_simd = (double)wps * iters * 5 * (KR2 ? 2 : 1);
    printf("%s %s waves/SIMD %d, ds_read_b128 per row %2d: %.3f ms, %.1f ns per tap per SIMD; 4K a-trous iteration at this rate: %.1f us\n", KR2 ? "2 outputs/thread" : "1 output/thread ", DB ? "double-buffered" : "single-buffered", wps, NRD, ms,
           ms * 1e6 / taps_per_simd, ms * 1e3 / taps_per_simd * (8294400.0 / 64 * 24 / 1024));
}

int main() {
    float *d, *in; hipMalloc(&d, 256 * 2048 * 4); hipMalloc(&in, 4096);
    hipMemset(in, 0x3c, 4096);
    for (int wps : {4, 2}) {   // 64 KB of LDS per block: at most 2 blocks per CU = 2 waves per SIMD
        run<10, false, false>(d, in, wps); run<10, false, true>(d, in, wps); run<10, true, false>(d, in, wps); run<10, true, true>(d, in, wps);
    }
    return 0;
}
