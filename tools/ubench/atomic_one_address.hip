// How fast do waves all over the chip append to ONE device counter?  (temporal_kernel: one atomicAdd per wave that holds young pixels.)
// mode 0: 32-bit atomicAdd; 1: 64-bit atomicAdd; 2: relaxed agent-scope load of the counter, then the 32-bit add; 3: load, then the 64-bit add;
// 4: the load alone.   hipcc --offload-arch=gfx950 -O3 -o build/atomic_one tools/ubench/atomic_one_address.hip && build/atomic_one
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void append(unsigned long long* counter, unsigned* list, int every, const float* in, float* out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float v = in[blockIdx.x * blockDim.x + threadIdx.x];          // a little streaming work per thread, as the temporal launch has
    unsigned base = 0;
    if (every && wave % every == 0) {
        if (lane == 0) {
            unsigned long long seen = 0;
            if (MODE >= 2) seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 4) base = (unsigned)seen;
            else if ((unsigned)(seen >> 32) < 0x7fffffffu) {
                if (MODE == 0 || MODE == 2) base = atomicAdd((unsigned*)counter, 1u);
                else base = (unsigned)atomicAdd(counter, (1ull << 32) | 1ull);
            }
        }
        base = __shfl(base, 0);
        if (lane == 0) list[base & 0xffff] = wave;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v * 1.0001f + (float)base;
}
template <int MODE> void run(unsigned long long* c, unsigned* l, const float* in, float* out, int waves) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int every : {0, 1, 16, 52}) {
        float best = 1e9f;
        for (int rep = 0; rep < 12; rep++) {
            hipMemset(c, 0, 8);
            hipEventRecord(a);
            append<MODE><<<waves * 64 / 256, 256>>>(c, l, every, in, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (rep > 2 && ms < best) best = ms;
        }
        printf("mode %d, one append per %3d waves (%6d appends): %.4f ms\n", MODE, every, every ? waves / every : 0, best);
    }
}
int main() {
    const int waves = 129600, threads = waves * 64;
    unsigned long long* c; unsigned* l; float *in, *out;
    (void)hipMalloc(&c, 8); (void)hipMalloc(&l, 65536 * 4); (void)hipMalloc(&in, threads * 4); (void)hipMalloc(&out, threads * 4);
    (void)hipMemset(in, 0, threads * 4);
    run<0>(c, l, in, out, waves); run<1>(c, l, in, out, waves); run<2>(c, l, in, out, waves); run<3>(c, l, in, out, waves); run<4>(c, l, in, out, waves);
    return 0;
}
