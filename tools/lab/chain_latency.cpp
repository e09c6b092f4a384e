// What one dependent step costs on ONE wave (the per-level chain of the Gauss-Seidel sweeps is made of these): a chain of
// dependent fp64 adds, of LDS write -> barrier -> read round trips (1, 4 and 16 waves in the workgroup), and of DPP lane shifts;
// nanoseconds per step from HIP events, so the clock the part really runs at under a one-CU load is in the number.
//   hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_add_chain(int n, double a, double *out) {
    double v = threadIdx.x;
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v = __dadd_rn(v, a);
    }
    out[threadIdx.x] = v;
}

__global__ void k_lds_barrier_chain(int n, double *out) {
    __shared__ double cell[1024];
    double v = threadIdx.x;
    cell[threadIdx.x] = v;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        v = cell[(threadIdx.x + 1) & (blockDim.x - 1)] + 1.0;  // read a neighbour's value of the previous step
        __syncthreads();
        cell[threadIdx.x] = v;
        __syncthreads();
    }
    out[threadIdx.x] = v;
}

__global__ void k_dpp_chain(int n, double *out) {
    double v = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
        v = __hiloint2double(hi, lo) + 1.0;
    }
    out[threadIdx.x] = v;
}

template <class F>
static double timed(F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    double *out;
    CK(hipMalloc(&out, 1024 * sizeof(double)));
    const int n = 1 << 20;
    double ms = timed([&] { hipLaunchKernelGGL(k_add_chain, dim3(1), dim3(64), 0, 0, n, 1.0, out); });
    printf("{\"dependent_fp64_add_ns\": %.2f", ms * 1e6 / n);
    ms = timed([&] { hipLaunchKernelGGL(k_dpp_chain, dim3(1), dim3(64), 0, 0, n, out); });
    printf(", \"dpp_shift_plus_add_ns\": %.2f", ms * 1e6 / n);
    for (int threads : {64, 256, 1024}) {
        const int m = 1 << 17;
        ms = timed([&] { hipLaunchKernelGGL(k_lds_barrier_chain, dim3(1), dim3(threads), 0, 0, m, out); });
        printf(", \"lds_write_barrier_read_barrier_ns_%d_threads\": %.2f", threads, ms * 1e6 / m);
    }
    printf("}\n");
    return 0;
}
