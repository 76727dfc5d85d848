// Micro-benchmark: SIMD time per wave-instruction (cycles, from s_memtime) of the instructions the a-trous tap loop is made
// of or could be made of, at 1 / 2 / 4 waves per SIMD, every CU busy.  Instructions are pinned with inline asm so that the
// compiler cannot fold, pack or reorder them.  Build: hipcc --offload-arch=gfx950 -O3 issue_costs.hip -o issue_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Op { FMA, FMA_MOD, PK_FMA, PK_ADD, PK_MUL, MUL, ADD, EXP, LOG, RCP, SQRT, MED3, CVT_F16, FMA_MIX, PK_FMA_F16, EXP_F16, DOT2, MOV_DPP, PERM,
          LDS_B128, LDS_B64, LDS_B32, LDS_R2B32, LDS_R2B64, AND_B32, MAX_F32, LDEXP, NOPS };
static const char* kNames[] = {"v_fma_f32", "v_fma_f32 -|a|", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_mul_f32", "v_add_f32", "v_exp_f32", "v_log_f32",
                               "v_rcp_f32", "v_sqrt_f32", "v_med3_f32", "v_cvt_f32_f16", "v_fma_mix_f32", "v_pk_fma_f16", "v_exp_f16", "v_dot2_f32_f16",
                               "v_mov_b32 dpp row_shr:1", "ds_bpermute_b32", "ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_read2_b32", "ds_read2_b64", "v_and_b32",
                               "v_max_f32", "v_ldexp_f32"};

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
    __shared__ f32x4 lds[1024];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 256) lds[i] = (f32x4){seed + i, seed, 1.0f, 2.0f};
    __syncthreads();
    float a[8];
    f32x2 p[8];
    f32x4 q[4];
    for (int i = 0; i < 8; i++) { a[i] = seed + t * 1e-3f + i; p[i] = (f32x2){a[i], a[i] + 0.5f}; }
    for (int i = 0; i < 4; i++) q[i] = (f32x4){a[i], a[i], a[i], a[i]};
    const float c1 = 1.0001f, c2 = 0.5f;
    const f32x2 pc1 = {1.0001f, 0.9999f}, pc2 = {0.5f, 0.25f};
    const unsigned addr = (unsigned)(t * 16) & 0x3fff;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int uu = 0; uu < 64; uu++) {
            const int u = uu & 7;
            if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(c1), "v"(c2));
            if constexpr (OP == FMA_MOD) asm volatile("v_fma_f32 %0, -|%0|, %1, %2" : "+v"(a[u]) : "v"(c1), "v"(c2));
            if constexpr (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(pc1), "v"(pc2));
            if constexpr (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[u]) : "v"(pc1));
            if constexpr (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u]) : "v"(pc1));
            if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c1));
            if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c1));
            if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[u]));
            if constexpr (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[u]));
            if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[u]));
            if constexpr (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[u]));
            if constexpr (OP == MED3) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(a[u]));
            if constexpr (OP == CVT_F16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[u]));
            if constexpr (OP == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %0, %0 op_sel_hi:[1,0,0]" : "+v"(a[u]) : "v"(c1));
            if constexpr (OP == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[u]) : "v"(c1), "v"(c2));
            if constexpr (OP == EXP_F16) asm volatile("v_exp_f16 %0, %0" : "+v"(a[u]));
            if constexpr (OP == DOT2) asm volatile("v_dot2_f32_f16 %0, %0, %1, 0" : "+v"(a[u]) : "v"(c1));
            if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
            if constexpr (OP == PERM) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(4)" : "+v"(a[u]) : "v"(addr >> 2));
            if constexpr (OP == LDS_B128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[u & 3]) : "v"(addr), "i"(u * 64));
            if constexpr (OP == LDS_B64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[u]) : "v"(addr >> 1), "i"(u * 64));
            if constexpr (OP == LDS_B32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[u]) : "v"(addr >> 2), "i"(u * 64));
            if constexpr (OP >= LDS_B128 && OP <= LDS_R2B64) { if (u == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            if constexpr (OP == LDS_R2B32) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p[u]) : "v"(addr >> 2), "i"(u), "i"(u + 32));
            if constexpr (OP == LDS_R2B64) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(q[u & 3]) : "v"(addr >> 1), "i"(u), "i"(u + 32));
            if constexpr (OP == AND_B32) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(a[u]));
            if constexpr (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c1));
            if constexpr (OP == LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[u]) : "v"(1));
        }
    }
    if constexpr (OP == PERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; i++) s += q[i].x + q[i].y + q[i].z + q[i].w;
    out[blockIdx.x * 256 + t] = s;
    if ((t & 63) == 0) cyc[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
}

static float* d_out; static unsigned long long* d_cyc;

template <int OP> void run(int wps) {
    const int iters = 1024, grid = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 256>>>(d_out, d_cyc, 16, 1.0f);
    hipEventRecord(e0);
    k<OP><<<grid, 256>>>(d_out, d_cyc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // a SIMD hosts `wps` waves; each issues iters*8 instructions during `med` cycles
    printf("%-26s waves/SIMD %d: %7.3f ms  %6.2f cycles of SIMD time per wave-instruction (wave sees %6.2f)  clock %.2f GHz\n", kNames[OP], wps, ms,
           med / ((double)iters * 64 * wps), med / ((double)iters * 64), med / (ms * 1e6));
}

template <int OP> void sweep() { for (int w : {1, 2, 3, 4, 6, 8}) run<OP>(w); }

int main() {
    hipMalloc(&d_out, 256 * 8 * 256 * 4); hipMalloc(&d_cyc, 256 * 8 * 4 * 8);
    sweep<FMA>(); sweep<FMA_MOD>(); sweep<PK_FMA>(); sweep<PK_ADD>(); sweep<PK_MUL>(); sweep<MUL>(); sweep<ADD>(); sweep<EXP>(); sweep<LOG>(); sweep<RCP>(); sweep<SQRT>();
    sweep<MED3>(); sweep<CVT_F16>(); sweep<FMA_MIX>(); sweep<PK_FMA_F16>(); sweep<EXP_F16>(); sweep<DOT2>(); sweep<MOV_DPP>(); sweep<PERM>();
    sweep<LDS_B128>(); sweep<LDS_B64>(); sweep<LDS_B32>(); sweep<LDS_R2B32>(); sweep<LDS_R2B64>(); sweep<AND_B32>(); sweep<MAX_F32>(); sweep<LDEXP>();
    return 0;
}
