// Probe 2: does the virtual-memory API get device memory faster than hipMalloc on a box whose hipMalloc is slow (27 ms per GB:
// VRAM a process has just released is still being scrubbed)?  (tools/lab: measurement helper, not product code.)
//   hipcc --offload-arch=gfx950 -O2 tools/lab/alloc_probe2.cpp -o /tmp/alloc_probe2
//   /tmp/alloc_probe2 hog 250     # a first process takes 250 GB, touches them, exits
//   /tmp/alloc_probe2 probe       # right behind it: hipMalloc vs hipMemCreate + hipMemMap, 16 GB pieces up to 128 GB
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    CK(hipSetDevice(0));
    const size_t GB = 1ull << 30;
    if (argc > 1 && !strcmp(argv[1], "hog")) {
        const size_t want = argc > 2 ? (size_t)atoi(argv[2]) : 250;
        std::vector<void *> ps;
        double t0 = now();
        for (size_t g = 0; g < want; g += 10) {
            void *p = nullptr;
            if (hipMalloc(&p, 10 * GB) != hipSuccess) break;
            hipMemset(p, 1, 10 * GB);
            ps.push_back(p);
        }
        hipDeviceSynchronize();
        printf("hog: %zu GB taken and touched in %.2f s\n", ps.size() * 10, now() - t0);
        return 0;   // the driver gets everything back at exit
    }
    const size_t piece = 16 * GB;
    {   // hipMalloc
        std::vector<void *> ps;
        double t0 = now();
        for (int i = 0; i < 8; ++i) { void *p = nullptr; CK(hipMalloc(&p, piece)); ps.push_back(p); }
        double t1 = now();
        for (void *p : ps) hipMemsetAsync(p, 0, piece, 0);
        hipDeviceSynchronize();
        double t2 = now();
        for (void *p : ps) hipFree(p);
        double t3 = now();
        printf("hipMalloc            8 x 16 GB: alloc %8.1f ms (%.1f ms per GB)  first touch %8.1f ms  free %8.1f ms\n", (t1 - t0) * 1e3, (t1 - t0) * 1e3 / 128, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    {   // virtual range + physical handles
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        void *base = nullptr;
        double t0 = now();
        CK(hipMemAddressReserve(&base, 8 * piece, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        for (int i = 0; i < 8; ++i) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, piece, &prop, 0));
            CK(hipMemMap((char *)base + i * piece, piece, 0, h, 0));
            CK(hipMemSetAccess((char *)base + i * piece, piece, &acc, 1));
            hs.push_back(h);
        }
        double t1 = now();
        hipMemsetAsync(base, 0, 8 * piece, 0);
        hipDeviceSynchronize();
        double t2 = now();
        for (int i = 0; i < 8; ++i) { hipMemUnmap((char *)base + i * piece, piece); hipMemRelease(hs[i]); }
        hipMemAddressFree(base, 8 * piece);
        double t3 = now();
        printf("hipMemCreate+Map     8 x 16 GB: alloc %8.1f ms (%.1f ms per GB)  first touch %8.1f ms  free %8.1f ms  (granularity %zu KB)\n", (t1 - t0) * 1e3, (t1 - t0) * 1e3 / 128, (t2 - t1) * 1e3, (t3 - t2) * 1e3, gran >> 10);
    }
    {   // hipMalloc again (what the scrubber has caught up with meanwhile)
        std::vector<void *> ps;
        double t0 = now();
        for (int i = 0; i < 8; ++i) { void *p = nullptr; CK(hipMalloc(&p, piece)); ps.push_back(p); }
        double t1 = now();
        for (void *p : ps) hipFree(p);
        printf("hipMalloc again      8 x 16 GB: alloc %8.1f ms (%.1f ms per GB)\n", (t1 - t0) * 1e3, (t1 - t0) * 1e3 / 128);
    }
    return 0;
}
