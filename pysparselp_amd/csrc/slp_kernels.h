// slp_kernels.h -- device code shared by the translation units of libslp_hip.so.
// CDNA4 / gfx950 only: 64-lane wavefronts are assumed throughout.
//
// Numerics contract (DESIGN.md "parity"): every multiply and every add rounds
// once (the library is built with -ffp-contract=off), dot products in
// SEQUENTIAL order walk a row in storage order with a single accumulator, so
// they reproduce scipy's csr_matvec / csc_matvec bit for bit; TREE order
// spreads a row over L lanes and combines with cross-lane shuffles.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace slp {

typedef int64_t i64;
typedef int32_t i32;

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 wavefronts per workgroup

// ---- cross-lane helpers ----------------------------------------------------
template <int L>
__device__ __forceinline__ double group_sum(double v) {
    // butterfly over the L lanes that share a row (L is a power of two <= 64)
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_down(v, off, kWave);
        v = (o > v) ? o : v;
    }
    return v;
}

// Workgroup reduction (sum or max) of one double per thread; valid in thread 0.
template <bool IS_MAX>
__device__ __forceinline__ double block_reduce(double v, double *lds /* >= kBlock/kWave */) {
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double r = IS_MAX ? -__builtin_inf() : 0.0;
    if (threadIdx.x == 0) {
        r = lds[0];
        for (int i = 1; i < (int)(blockDim.x / kWave); ++i) {
            if (IS_MAX) r = (lds[i] > r) ? lds[i] : r;
            else r += lds[i];
        }
    }
    __syncthreads();
    return r;
}

// numpy evaluates |a| ** p; p == 1, 2, 0 and 0.5 (numpy's scalar-exponent fast paths: identity, square, ones, sqrt) are exact,
// other p go through pow -- the host's libm / SIMD pow there, ocml's here: equal to rounding, not to the bit
// (ChambollePockPPD.py:134,144,161,172)
__device__ __forceinline__ double abs_pow(double a, double p) {
    a = fabs(a);
    if (p == 1.0) return a;
    if (p == 2.0) return a * a;
    if (p == 0.0) return 1.0;
    if (p == 0.5) return sqrt(a);
    return pow(a, p);
}

// ---- one row . dense vector -------------------------------------------------
// L == 1: storage order, single accumulator (bit-exact csr_matvec semantics).
// L  > 1: lane `sub` of the row's L-lane group takes entries sub, sub+L, ...;
//         consecutive lanes read consecutive (value, index) pairs -> coalesced.
template <int L>
__device__ __forceinline__ double row_dot(const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                          const double *__restrict__ val, const double *__restrict__ x,
                                          i64 row, int sub) {
    const i64 s = ptr[row], e = ptr[row + 1];
    if (L == 1) {
        // single accumulator, storage order; loads are issued four entries at a time so that the
        // index -> gather latency chain is paid once per batch, the adds stay in order
        double acc = 0.0;
        for (i64 k = s; k < e; k += 4) {
            i32 j[4];
            double a[4], xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i64 kk = (k + q < e) ? k + q : e - 1;
                j[q] = idx[kk];
                a[q] = val[kk];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = x[j[q]];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (k + q < e) acc += a[q] * xv[q];
        }
        return acc;
    } else {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        i64 k = s + sub;
        // four independent loads in flight per lane before the first use
        for (; k + 3 * L < e; k += 4 * L) {
            const double v0 = val[k], v1 = val[k + L], v2 = val[k + 2 * L], v3 = val[k + 3 * L];
            const i32 j0 = idx[k], j1 = idx[k + L], j2 = idx[k + 2 * L], j3 = idx[k + 3 * L];
            a0 += v0 * x[j0];
            a1 += v1 * x[j1];
            a2 += v2 * x[j2];
            a3 += v3 * x[j3];
        }
        for (; k < e; k += L) a0 += val[k] * x[idx[k]];
        return group_sum<L>((a0 + a1) + (a2 + a3));
    }
}

// Same walk, but terms whose index is below `split` go to *lo, the rest to
// *hi (column of K = [A_eq; A_ineq]: the reference accumulates y_eq*A_eq and
// y_ineq*A_ineq separately, ChambollePockPPD.py:206,216).
template <int L>
__device__ __forceinline__ void row_dot_split(const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                              const double *__restrict__ val, const double *__restrict__ x,
                                              i64 row, int sub, i32 split, double *lo, double *hi) {
    const i64 s = ptr[row], e = ptr[row + 1];
    double a = 0.0, b = 0.0;
    if (L == 1) {
        for (i64 k = s; k < e; ++k) {
            const i32 j = idx[k];
            const double p = val[k] * x[j];
            if (j < split) a += p;
            else b += p;
        }
    } else {
        double a1 = 0.0, b1 = 0.0;
        i64 k = s + sub;
        for (; k + L < e; k += 2 * L) {
            const double v0 = val[k], v1 = val[k + L];
            const i32 j0 = idx[k], j1 = idx[k + L];
            const double p0 = v0 * x[j0], p1 = v1 * x[j1];
            if (j0 < split) a += p0; else b += p0;
            if (j1 < split) a1 += p1; else b1 += p1;
        }
        if (k < e) {
            const i32 j = idx[k];
            const double p = val[k] * x[j];
            if (j < split) a += p; else b += p;
        }
        a = group_sum<L>(a + a1);
        b = group_sum<L>(b + b1);
    }
    *lo = a;
    *hi = b;
}

// Dispatch a kernel template on the lanes-per-row count chosen on the host.
#define SLP_DISPATCH_LANES(lanes, CALL)                   \
    switch (lanes) {                                      \
        case 1:  { constexpr int L = 1;  CALL; } break;   \
        case 2:  { constexpr int L = 2;  CALL; } break;   \
        case 4:  { constexpr int L = 4;  CALL; } break;   \
        case 8:  { constexpr int L = 8;  CALL; } break;   \
        case 16: { constexpr int L = 16; CALL; } break;   \
        case 32: { constexpr int L = 32; CALL; } break;   \
        default: { constexpr int L = 64; CALL; } break;   \
    }

}  // namespace slp
