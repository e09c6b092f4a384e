// What does the SHAPE of the loads cost a streaming kernel?  256 x k workgroups of 1024 threads, one contiguous stream each (as the
// tall-cell kernel's row blocks, slp_tall.hip), the same bytes read with
//   b32 : 64 lanes x 4 bytes per instruction, 10 instructions per "packet" at 10 different offsets of the stream (the kernel's slots),
//   b128: 64 lanes x 16 bytes per instruction (a quarter of the instructions),
// each with `DEPTH` packets in flight per wave, through raw buffer loads (as the kernel issues them) with the stream-once policy.
// Lab tool for DESIGN.md section 3 ("is the 4.45 TB/s floor of the packet stream the price of 4-byte loads?").
//   hipcc --offload-arch=gfx950 -O3 -o stream_shapes stream_shapes.cpp && ./stream_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kT = 1024;
constexpr int kSlots = 10;  // 4-byte words per lane and packet

template <int DEPTH, int AUX>
__global__ __launch_bounds__(kT) void k_b32(const unsigned *__restrict__ src, size_t words_per_wg, unsigned *out) {
    const unsigned *base = src + (size_t)blockIdx.x * words_per_wg;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(base), 0, 0x7fffffff, 0x00020000);
    const unsigned mine = threadIdx.x * 4u;
    const unsigned pkt_bytes = kSlots * kT * 4u;
    const int npk = (int)(words_per_wg / (kSlots * kT));
    unsigned v[DEPTH][kSlots];
    unsigned acc = 0;
    auto issue = [&](unsigned (&r)[kSlots], int j) {
        unsigned so = (unsigned)j * pkt_bytes;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            r[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, mine, so, AUX);
            so += kT * 4u;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(v[d], d);
    for (int j = 0; j + DEPTH <= npk - DEPTH; j += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int k = 0; k < kSlots; ++k) acc += v[d][k];
            issue(v[d], j + d + DEPTH);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH, int AUX>
__global__ __launch_bounds__(kT) void k_b128(const unsigned *__restrict__ src, size_t words_per_wg, unsigned *out) {
    const unsigned *base = src + (size_t)blockIdx.x * words_per_wg;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(base), 0, 0x7fffffff, 0x00020000);
    const unsigned mine = threadIdx.x * 16u;
    constexpr int kQ = (kSlots + 3) / 4;             // 16-byte loads per lane and packet (12 words: a little more than the b32 form)
    const unsigned pkt_bytes = kQ * kT * 16u;
    const int npk = (int)(words_per_wg / (kQ * kT * 4));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v[DEPTH][kQ];
    unsigned acc = 0;
    auto issue = [&](u4 (&r)[kQ], int j) {
        unsigned so = (unsigned)j * pkt_bytes;
#pragma unroll
        for (int k = 0; k < kQ; ++k) {
            r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, mine, so, AUX);
            so += kT * 16u;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(v[d], d);
    for (int j = 0; j + DEPTH <= npk - DEPTH; j += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int k = 0; k < kQ; ++k) acc += v[d][k].x ^ v[d][k].y ^ v[d][k].z ^ v[d][k].w;
            issue(v[d], j + d + DEPTH);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class K>
static double time_ms(K launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const size_t total = (size_t)13 << 30;  // ~ the slice's copy
    unsigned *src, *out;
    CK(hipMalloc(&src, total));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, total));
    for (int wgs : {256, 512, 1024}) {
        const size_t words = (total / 4 / wgs) / (12 * kT) * (12 * kT);
        const double gb = (double)words * 4 * wgs / 1e9;
#define RUN(NAME, KERNEL)                                                                                          \
    {                                                                                                              \
        const double ms = time_ms([&] { hipLaunchKernelGGL(KERNEL, dim3(wgs), dim3(kT), 0, 0, src, words, out); }, 5); \
        printf("%4d workgroups  %-28s %7.3f ms  %6.2f TB/s\n", wgs, NAME, ms, gb / ms);                           \
    }
        RUN("b32  depth 4 stream-once", (k_b32<4, 2>));
        RUN("b32  depth 4 default", (k_b32<4, 0>));
        RUN("b32  depth 2 stream-once", (k_b32<2, 2>));
        RUN("b128 depth 4 stream-once", (k_b128<4, 2>));
        RUN("b128 depth 4 default", (k_b128<4, 0>));
        RUN("b128 depth 2 stream-once", (k_b128<2, 2>));
        RUN("b128 depth 1 stream-once", (k_b128<1, 2>));
    }
    return 0;
}
