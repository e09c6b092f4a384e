// How fast can ONE compute unit read?  One workgroup of 1024 threads streams a buffer with 16-byte loads, `depth` loads in
// flight per lane; the buffer is either far larger than the L2 (HBM / MALL) or small enough to stay in the XCD's L2 (re-read).
// Also: the same with a second workgroup on the same XCD touching the data ahead ("warmer").  Lab tool for the single-
// workgroup Gauss-Seidel sweep.   hipcc --offload-arch=gfx950 -O3 -o one_cu_stream one_cu_stream.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(1024) void k_stream(const uint4 *__restrict__ src, size_t n16, int passes, unsigned *out) {
    unsigned acc = 0;
    for (int p = 0; p < passes; ++p) {
        for (size_t i = threadIdx.x; i + (size_t)(DEPTH - 1) * 1024 < n16; i += (size_t)DEPTH * 1024) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = src[i + (size_t)d * 1024];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// workgroup 0 consumes; workgroups whose XCC id equals workgroup 0's touch the data `ahead` bytes in front of it
__global__ __launch_bounds__(1024) void k_stream_warm(const uint4 *__restrict__ src, size_t n16, unsigned *out, unsigned long long *progress,
                                                      int *xcc_of_consumer, size_t ahead16) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) __hip_atomic_store(xcc_of_consumer, (int)xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned acc = 0;
        constexpr int DEPTH = 8;
        for (size_t i = threadIdx.x; i + (size_t)(DEPTH - 1) * 1024 < n16; i += (size_t)DEPTH * 1024) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = src[i + (size_t)d * 1024];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
            if (threadIdx.x == 0) __hip_atomic_store(progress, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) __hip_atomic_store(progress, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (acc == 0x12345678u) out[0] = acc;
        return;
    }
    int cx;
    while ((cx = __hip_atomic_load(xcc_of_consumer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) __builtin_amdgcn_s_sleep(8);
    if ((unsigned)cx != xcc) return;
    if (threadIdx.x == 0) atomicAdd(out + 1, 1u);  // how many warmers share the consumer's XCD
    unsigned acc = 0;
    // one 4-byte load per 128-byte line, 1024 lines (128 KB) per round, 8 rounds in flight
    unsigned v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t base = 0; base < n16; base += 8 * 8 * 1024) {
        for (;;) {
            const unsigned long long p = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p == ~0ull) return;
            if (base < p + ahead16) break;
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const size_t i = base + (size_t)r * 8 * 1024 + (size_t)threadIdx.x * 8;
            acc += v[r];
            if (i < n16) v[r] = src[i].x;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    unsigned *out;
    CK(hipMalloc(&out, 64));
    CK(hipMemset(out, 0, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](auto launch) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
    for (size_t mb : {1, 2, 64, 512}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        uint4 *buf;
        CK(hipMalloc(&buf, bytes));
        CK(hipMemset(buf, 1, bytes));
        const int passes = mb <= 2 ? 256 : (mb == 64 ? 4 : 1);
        auto run = [&](auto kernel, const char *name) {
            hipLaunchKernelGGL(kernel, dim3(1), dim3(1024), 0, 0, buf, n16, 1, out);  // warm
            const float ms = time([&] { hipLaunchKernelGGL(kernel, dim3(1), dim3(1024), 0, 0, buf, n16, passes, out); });
            printf("buffer %4zu MB  %-10s  %7.3f ms  %7.1f GB/s\n", mb, name, ms, (double)bytes * passes / ms / 1e6);
        };
        run(k_stream<2>, "depth 2");
        run(k_stream<4>, "depth 4");
        run(k_stream<8>, "depth 8");
        CK(hipFree(buf));
    }
    // consumer + warmers
    {
        const size_t bytes = (size_t)512 << 20, n16 = bytes / 16;
        uint4 *buf;
        CK(hipMalloc(&buf, bytes));
        CK(hipMemset(buf, 1, bytes));
        unsigned long long *progress;
        int *cx;
        CK(hipMalloc(&progress, 8));
        CK(hipMalloc(&cx, 4));
        for (size_t ahead_kb : {1024, 2048, 3072}) {
            for (int wgs : {1, 9, 17, 33}) {
                CK(hipMemset(progress, 0, 8));
                CK(hipMemset(cx, 0xff, 4));
                CK(hipMemset(out, 0, 64));
                const float ms = time([&] { hipLaunchKernelGGL(k_stream_warm, dim3(wgs), dim3(1024), 0, 0, buf, n16, out, progress, cx, ahead_kb * 1024 / 16); });
                unsigned h[2];
                CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
                printf("512 MB, %2d workgroups (%u warmers on the consumer's XCD), ahead %4zu KB: %7.3f ms  %7.1f GB/s\n", wgs, h[1], ahead_kb, ms,
                       (double)bytes / ms / 1e6);
            }
        }
    }
    return 0;
}
