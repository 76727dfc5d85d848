// Micro-benchmark: an LDS-free imitation of the a-trous tap arithmetic (13 VALU incl. log2 + exp2 per tap) as a
// function of waves per SIMD.  Tells how much of the kernel's time is the VALU floor at a given occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(float* out, const float* in, int iters) {
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    float lc = in[t & 1023], zc = in[(t + 7) & 1023], ncz = in[(t + 3) & 1023], il = 3.0f, iz = 2.0f, phi = 128.0f;
    half2_t nc = __builtin_bit_cast(half2_t, in[(t + 11) & 1023]);
    float sw = 1.0f; f32x2 srg = {0.1f, 0.2f}, sbv = {0.3f, 0.4f};
    float bx = in[(t + 1) & 1023], by = in[(t + 2) & 1023], bz = in[(t + 5) & 1023], bw = in[(t + 9) & 1023];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            bx += 1e-3f;   // keep taps distinct (1 op of overhead)
            float d = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, bz), nc, 0.0f, false);
            d = fminf(fmaxf(fmaf(bw, ncz, d), 0.f), 1.f);
            float e = fmaf(__builtin_amdgcn_logf(d), phi, -0.58f);
            e = fmaf(-fabsf(bx - lc), il, e);
            e = fmaf(-fabsf(by - zc), iz, e);
            float w = __builtin_amdgcn_exp2f(e);
            f32x2 ww = {w, w * w};
            sw += w;
            srg = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){bx, by}, srg);
            sbv = __builtin_elementwise_fma(ww, (f32x2){bz, bw}, sbv);
        }
    }
    out[t] = sw + srg.x + srg.y + sbv.x + sbv.y;
}

int main() {
    float *d, *in; hipMalloc(&d, 256 * 2048 * 4 * 8); hipMalloc(&in, 4096);
    hipMemset(in, 0x3c, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1024;
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
        int grid = 256 * wps;   // blocks of 256 threads = 4 waves = 1 wave per SIMD each
        k<256><<<grid, 256>>>(d, in, 8);
        hipEventRecord(e0);
        k<256><<<grid, 256>>>(d, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double taps_per_simd = (double)wps * iters * 8;
        printf("waves/SIMD %d: %.3f ms, %.1f ns per tap per SIMD (= %.1f cycles at 2.4 GHz); 4K a-trous iteration at this rate: %.1f us\n", wps, ms,
               ms * 1e6 / taps_per_simd, ms * 1e6 / taps_per_simd * 2.4, ms * 1e3 / taps_per_simd * (8294400.0 / 64 * 24 / 1024));
    }
    return 0;
}
