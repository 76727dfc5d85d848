// cu_mask.hip — can a stream be kept OFF a few compute units, so that the halo exchange's kernels always find room beside a filter launch that
// oversubscribes the chip?  (tools/archive/rccl_selfcopy.py: a loop-back RCCL group takes 15-23 us on an idle device and 75-150 us beside the filter
// launches, at any stream priority: its workgroups wait for whole CUs' worth of resources that the dispatcher keeps handing to the next filter workgroup.)
//   1. hipExtStreamCreateWithCUMask with n of the low mask bits set: how long does a fixed grid take?  (which bits are which CUs: per-XCD interleaved or blocked)
//   2. a saturating "filter" kernel on a stream masked to all but k CUs per XCD, and an 8-workgroup x 512-thread "exchange" kernel (64 KB of LDS each) on an unmasked
//      high-priority stream enqueued while the first runs: the exchange kernel's start-to-end time and its completion time relative to the filter kernel's.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/cu_mask tools/ubench/cu_mask.hip ; run under `timeout 120`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void busy(long long ticks, unsigned* sink) {            // a filter-like workgroup: 256 threads, 30 KB of LDS, ~90 registers are not modelled
    __shared__ unsigned lds[7680];
    const long long t0 = wall_clock64();
    unsigned acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) { acc = acc * 1664525u + 1013904223u; lds[threadIdx.x] = acc; }
    if (acc == 0xdeadbeefu) sink[0] = lds[(threadIdx.x + 1) & 255];
}
__global__ void exchange(long long ticks, unsigned* sink) {        // an exchange-like workgroup: 512 threads, 64 KB of LDS
    __shared__ unsigned lds[16384];
    const long long t0 = wall_clock64();
    unsigned acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) { acc = acc * 1664525u + 1013904223u; lds[threadIdx.x] = acc; }
    if (acc == 0xdeadbeefu) sink[0] = lds[(threadIdx.x + 1) & 511];
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s: %d CUs\n", p.name, ncu);
    unsigned* sink = nullptr;
    CK(hipMalloc((void**)&sink, 64));
    hipEvent_t e0, e1, x0, x1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&x0)); CK(hipEventCreate(&x1));
    const int words = (ncu + 31) / 32;
    auto masked = [&](std::vector<uint32_t> m, hipStream_t* s) { return hipExtStreamCreateWithCUMask(s, (uint32_t)m.size(), m.data()); };
    // 1. which bits are which CUs
    for (int pattern = 0; pattern < 5; pattern++) {
        std::vector<uint32_t> m(words, 0u);
        int set = 0;
        for (int i = 0; i < ncu; i++) {
            bool on = pattern == 0 ? true : pattern == 1 ? i < ncu / 2 : pattern == 2 ? (i % 2 == 0) : pattern == 3 ? (i % 8 != 0) : (i / 8 != 0) /* all but CUs 0..7 */;
            if (on) { m[i / 32] |= 1u << (i % 32); set++; }
        }
        hipStream_t s;
        hipError_t e = masked(m, &s);
        if (e != hipSuccess) { printf("pattern %d: hipExtStreamCreateWithCUMask -> %s\n", pattern, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(busy, dim3(ncu * 5), dim3(256), 0, s, 100, sink); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(busy, dim3(ncu * 5 * 4), dim3(256), 0, s, 2000, sink);      // 4 rounds of 20 us on the whole chip
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const char* names[] = {"all bits", "low half of the bits", "even bits", "all but every 8th bit", "all but bits 0..7"};
        printf("mask %-22s (%3d bits set): %d workgroups x 20 us take %.3f ms\n", names[pattern], set, ncu * 20, ms);
        CK(hipStreamDestroy(s));
    }
    // 2. does an exchange kernel find room?
    for (int reserve = 0; reserve <= 2; reserve++) {
        for (int how = 0; how < 2; how++) {                    // how the reserved CUs are chosen: bit i % 8 == 0 (one per "row of 8") or the first 8 * reserve bits
            if (reserve == 0 && how == 1) continue;
            std::vector<uint32_t> m(words, 0u);
            int set = 0;
            for (int i = 0; i < ncu; i++) {
                const bool off = how == 0 ? (i % 32) < reserve * 1 && false : false;
                (void)off;
                bool on = true;
                if (reserve) on = how == 0 ? !((i / 8) % 4 == 0 && (i % 8) < 8 && (i / 32) * 0 == 0 && (i % 32) < reserve) : !(i < 8 * reserve);
                if (on) { m[i / 32] |= 1u << (i % 32); set++; }
            }
            hipStream_t s, c;
            if (masked(m, &s) != hipSuccess) { printf("reserve %d: mask refused\n", reserve); (void)hipGetLastError(); continue; }
            int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            CK(hipStreamCreateWithPriority(&c, hipStreamNonBlocking, hi));
            hipLaunchKernelGGL(exchange, dim3(8), dim3(512), 0, c, 100, sink); CK(hipStreamSynchronize(c));
            float worst = 0, sum = 0, rel = 0;
            const int reps = 10;
            for (int r = 0; r < reps; r++) {
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(busy, dim3(ncu * 5 * 12), dim3(256), 0, s, 1000, sink);       // ~12 rounds of 10 us: a 120 us filter launch, 12x oversubscribed
                CK(hipEventRecord(e1, s));
                CK(hipEventRecord(x0, c));
                hipLaunchKernelGGL(exchange, dim3(8), dim3(512), 0, c, 1000, sink);             // 10 us of its own
                CK(hipEventRecord(x1, c));
                CK(hipDeviceSynchronize());
                float tx = 0, tf = 0, t01 = 0;
                CK(hipEventElapsedTime(&tx, x0, x1)); CK(hipEventElapsedTime(&tf, e0, e1)); CK(hipEventElapsedTime(&t01, e0, x1));
                sum += tx; worst = tx > worst ? tx : worst; rel += t01 / tf;
            }
            printf("filter stream keeps off %d CUs (%s, %3d bits set): exchange kernel (8 x 512 threads, 64 KB LDS, 10 us of work) takes %.1f us on average, %.1f worst; done at %.2f of the filter launch\n",
                   ncu - set, how == 0 ? "bits 0..k-1 of every 32" : "the first bits", set, sum / reps * 1e3, worst * 1e3, rel / reps);
            CK(hipStreamDestroy(s)); CK(hipStreamDestroy(c));
        }
    }
    printf("done\n");
    return 0;
}
