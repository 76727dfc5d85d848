// Micro-benchmark: the tap arithmetic (13 VALU) with NL ds_read_b128 per tap feeding it, 4 waves per SIMD.
// Question: do LDS reads (well below LDS bandwidth) slow the VALU stream down?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int NL>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    __shared__ f32x4 lds[2048];
    const int t = threadIdx.x;
    for (int i = t; i < 2048; i += 256) lds[i] = (f32x4){in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]};
    __syncthreads();
    float lc = in[t & 1023], zc = in[(t + 7) & 1023], ncz = in[(t + 3) & 1023], il = 3.0f, iz = 2.0f, phi = 128.0f;
    half2_t nc = __builtin_bit_cast(half2_t, in[(t + 11) & 1023]);
    float sw = 1.0f; f32x2 srg = {0.1f, 0.2f}, sbv = {0.3f, 0.4f};
    int base = t;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            f32x4 A = {0.1f, 0.2f, 0.3f, 0.4f}, B = {0.5f, 0.6f, 0.7f, 0.8f};
            if (NL >= 1) B = lds[(base + u * 256) & 2047];
            if (NL >= 2) A = lds[(base + u * 256 + 64) & 2047];
            if (NL >= 3) { f32x4 C = lds[(base + u * 256 + 128) & 2047]; A += C; }
            if (NL >= 4) { f32x4 C = lds[(base + u * 256 + 192) & 2047]; B += C; }
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
        base += 1;
    }
    out[blockIdx.x * 256 + t] = sw + srg.x + srg.y + sbv.x + sbv.y;
}

template <int NL> void run(float* d, float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1024, wps = 4, grid = 256 * wps;
    k<NL><<<grid, 256>>>(d, in, 8);
    hipEventRecord(e0);
    k<NL><<<grid, 256>>>(d, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double taps_per_simd = (double)wps * iters * 8;
    printf("ds_read_b128 per tap %d: %.3f ms, %.1f ns per tap per SIMD; 4K a-trous iteration at this rate: %.1f us\n", NL, ms,
           ms * 1e6 / taps_per_simd, ms * 1e3 / taps_per_simd * (8294400.0 / 64 * 24 / 1024));
}

int main() {
    float *d, *in; hipMalloc(&d, 256 * 2048 * 4); hipMalloc(&in, 4096);
    hipMemset(in, 0x3c, 4096);
    run<0>(d, in); run<1>(d, in); run<2>(d, in); run<3>(d, in); run<4>(d, in);
    return 0;
}
