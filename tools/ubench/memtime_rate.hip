// What one s_memtime tick is worth: s_memtime against s_memrealtime (100 MHz) over the same busy loop, once alone and once with every CU busy.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/memtime_rate.hip -o build/memtime_rate && build/memtime_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int iters) {
    unsigned long long t0, t1, w0, w1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(w0) :: "memory");
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) x = fmaf(x, 1.0000001f, 1e-7f);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(w1) : "v"(x) : "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (unsigned long long)x; }
}
int main() {
    unsigned long long* d; unsigned long long h[3];
    hipMalloc(&d, 24);
    for (int blocks : {1, 2048}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 4000000);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("blocks %d: s_memtime ticks %llu, s_memrealtime ticks %llu (100 MHz) -> %.1f s_memtime ticks per microsecond; 4e6 dependent v_fma -> %.2f ticks per fma\n",
               blocks, h[0], h[1], 100.0 * h[0] / h[1], h[0] / 4e6);
    }
    return 0;
}
