// Micro-benchmark 2: the kernel's row structure — per ring row NRD ds_read_b128 (immediate offsets) then 5 taps of
// 13 VALU — at W waves per SIMD.  Isolates what LDS reads cost the VALU stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int NRD>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    extern __shared__ f32x4 lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 256) lds[i] = (f32x4){in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]};
    __syncthreads();
    float lc = in[t & 1023], zc = in[(t + 7) & 1023], ncz = in[(t + 3) & 1023], il = 3.0f, iz = 2.0f, phi = 128.0f;
    half2_t nc = __builtin_bit_cast(half2_t, in[(t + 11) & 1023]);
    float sw = 1.0f; f32x2 srg = {0.1f, 0.2f}, sbv = {0.3f, 0.4f};
    int base = t;
    for (int it = 0; it < iters; it++) {
        f32x4 tA[5], tB[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            tA[c] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f} + (float)it; tB[c] = (f32x4){0.5f, 0.6f, 0.7f, 0.8f} * (float)it;
            if (2 * c < NRD) tB[c] = lds[(base + c * 16 + 512) & 1023];
            if (2 * c + 1 < NRD) tA[c] = lds[(base + c * 16) & 1023];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const f32x4 A = tA[c], B = tB[c];
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
        }
        asm volatile("" : "+v"(sw), "+v"(srg), "+v"(sbv) :: "memory");
        base = (base + 320) & 511;
    }
    out[blockIdx.x * 256 + t] = sw + srg.x + srg.y + sbv.x + sbv.y;
}

template <int NRD> void run(float* d, float* in, int wps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, grid = 256 * wps;
    hipFuncSetAttribute((const void*)k<NRD>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
    k<NRD><<<grid, 256, 16384>>>(d, in, 8);
    hipEventRecord(e0);
    k<NRD><<<grid, 256, 16384>>>(d, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double taps_per_simd = (double)wps * iters * 5;
    printf("waves/SIMD %d, ds_read_b128 per 5 taps %2d: %.3f ms, %.1f ns per tap per SIMD; 4K a-trous iteration at this rate: %.1f us\n", wps, NRD, ms,
           ms * 1e6 / taps_per_simd, ms * 1e3 / taps_per_simd * (8294400.0 / 64 * 24 / 1024));
}

int main() {
    float *d, *in; hipMalloc(&d, 256 * 2048 * 4); hipMalloc(&in, 4096);
    hipMemset(in, 0x3c, 4096);
    for (int wps : {8, 6, 4, 2}) {   // 64 KB of LDS per block: at most 2 blocks per CU = 2 waves per SIMD
        run<0>(d, in, wps); run<4>(d, in, wps); run<6>(d, in, wps); run<10>(d, in, wps);
    }
    return 0;
}
