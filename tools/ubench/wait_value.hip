// wait_value.hip — can a stream be made to wait for a value that a RUNNING kernel of another stream writes?  (VERDICT r04 #2a: one a-trous launch
// per iteration with the edge tiles first; the communication stream starts the halo exchange when the last edge tile has signalled, while
// the interior tiles of the same launch still run.)  Three ways to wait, each timed against a kernel whose first 64 workgroups signal
// after ~100 us while its other 1 984 run ~200 us:
//   1. hipStreamWaitValue64 on signal memory (hipExtMallocWithFlags(hipMallocSignalMemory)), written by the kernel
//   2. hipStreamWaitValue64 on plain device memory
//   3. a one-wave kernel on the waiting stream that polls the word (what RCCL's own kernels do)
// Prints, per way: ms from the kernel's start to "the waiting stream got past its wait", next to the kernel's end; a wait that only
// returns when the kernel has ENDED is useless here.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/wait_value tools/ubench/wait_value.hip ; run under `timeout 60`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void signaller(unsigned long long* flag, unsigned long long* counter, unsigned long long value, long long half_ticks, int nfirst) {
    const long long t0 = wall_clock64();
    const bool edge = (int)blockIdx.x < nfirst;
    while (wall_clock64() - t0 < (edge ? half_ticks : 2 * half_ticks)) __builtin_amdgcn_s_sleep(8);
    if (edge && threadIdx.x == 0) {
        __threadfence();
        const unsigned long long n = atomicAdd(counter, 1ull);
        if (n + 1 == (unsigned long long)nfirst) { *counter = 0ull; __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}
__global__ void poller(const unsigned long long* flag, unsigned long long value) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) __builtin_amdgcn_s_sleep(16);
}

int main() {
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e0, eb, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&eb)); CK(hipEventCreate(&e1));
    unsigned long long *sig = nullptr, *plain = nullptr, *counter = nullptr;
    hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory, 8 B) -> %s\n", hipGetErrorString(es));
    if (es != hipSuccess) { sig = nullptr; (void)hipGetLastError(); }
    CK(hipMalloc((void**)&plain, 256)); CK(hipMemset(plain, 0, 256));
    CK(hipMalloc((void**)&counter, 256)); CK(hipMemset(counter, 0, 256));
    if (sig) { hipError_t e = hipMemset(sig, 0, 8); printf("hipMemset(signal memory) -> %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
    const long long half = 100 * 100;            // wall_clock64 ticks at 100 MHz: 100 us
    for (int way = 1; way <= 3; way++) {
        unsigned long long* flag = way == 1 ? sig : plain;
        if (!flag) { printf("way %d: no memory\n", way); continue; }
        for (int rep = 0; rep < 3; rep++) {
            const unsigned long long value = 10 * way + rep + 1;
            CK(hipEventRecord(e0, a));
            hipLaunchKernelGGL(signaller, dim3(2048), dim3(64), 0, a, flag, counter, value, half, 64);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, a));
            hipError_t ew = hipSuccess;
            if (way <= 2) ew = hipStreamWaitValue64(b, flag, value, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
            else hipLaunchKernelGGL(poller, dim3(1), dim3(64), 0, b, flag, value);
            if (ew != hipSuccess) { printf("way %d: hipStreamWaitValue64 -> %s\n", way, hipGetErrorString(ew)); (void)hipGetLastError(); CK(hipDeviceSynchronize()); break; }
            CK(hipEventRecord(eb, b));
            CK(hipDeviceSynchronize());
            float tb = 0, t1 = 0;
            CK(hipEventElapsedTime(&tb, e0, eb)); CK(hipEventElapsedTime(&t1, e0, e1));
            printf("way %d rep %d: waiting stream released at %.3f ms, signalling kernel ended at %.3f ms (signal at ~0.1, end at ~0.2+)\n", way, rep, tb, t1);
        }
    }
    printf("done\n");
    return 0;
}
