// Probe: how long do large device allocations take on this box, by API?  (tools/lab: measurement helpers, not product code)
//   hipcc --offload-arch=gfx950 -O2 tools/lab/alloc_probe.cpp -o /tmp/alloc_probe && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0);
    hipStream_t st;
    hipStreamCreate(&st);
    const size_t GB = 1ull << 30;
    for (int round = 0; round < 2; ++round) {
        for (size_t gb : {1ull, 8ull, 16ull}) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, gb * GB);
            double t1 = now();
            hipMemsetAsync(p, 0, gb * GB, st);
            hipStreamSynchronize(st);
            double t2 = now();
            hipFree(p);
            double t3 = now();
            printf("round %d hipMalloc      %2zu GB: alloc %8.2f ms  first memset %8.2f ms  free %8.2f ms (%s)\n", round, gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, hipGetErrorString(e));
        }
        for (size_t gb : {1ull, 8ull, 16ull}) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMallocAsync(&p, gb * GB, st);
            hipStreamSynchronize(st);
            double t1 = now();
            hipMemsetAsync(p, 0, gb * GB, st);
            hipStreamSynchronize(st);
            double t2 = now();
            hipFreeAsync(p, st);
            hipStreamSynchronize(st);
            double t3 = now();
            printf("round %d hipMallocAsync %2zu GB: alloc %8.2f ms  first memset %8.2f ms  free %8.2f ms (%s)\n", round, gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, hipGetErrorString(e));
        }
    }
    return 0;
}
