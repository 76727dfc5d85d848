// Micro-benchmark: issue cost of the VALU instructions the a-trous tap loop is made of, at 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3f + i;
    f32x2 p[4];
    for (int i = 0; i < 4; i++) p[i] = (f32x2){a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) a[u] = fmaf(a[u], 1.0001f, 0.5f);
            if (OP == 1) a[u] = __builtin_amdgcn_exp2f(a[u]) ;
            if (OP == 2) a[u] = __builtin_amdgcn_logf(a[u]);
            if (OP == 3) p[u & 3] = __builtin_elementwise_fma(p[u & 3], (f32x2){1.0001f, 0.9999f}, (f32x2){0.5f, 0.25f});
            if (OP == 4) a[u] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a[u]), __builtin_bit_cast(half2_t, a[(u + 1) & 7]), a[u], false);
            if (OP == 5) a[u] = __builtin_amdgcn_rcpf(a[u]);
            if (OP == 6) a[u] = a[u] * a[u];
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> double run(const char* name, int blocks_per_cu) {
    int iters = 4096;
    float* d; hipMalloc(&d, 256 * 256 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    k<OP><<<grid, 256>>>(d, 16, 1.0f);
    hipEventRecord(e0);
    k<OP><<<grid, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves = blocks_per_cu (4 waves per block over 4 SIMDs), instr per wave = iters*8
    double instr_per_simd = (double)blocks_per_cu * iters * 8;
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", name, blocks_per_cu, ms, ns_per_instr, ns_per_instr * 2.4);
    hipFree(d);
    return ns_per_instr;
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<3>("v_pk_fma_f32", w); run<6>("v_mul_f32", w); run<1>("v_exp_f32", w); run<2>("v_log_f32", w);
        run<5>("v_rcp_f32", w); run<4>("v_dot2c_f32_f16", w);
    }
    return 0;
}
