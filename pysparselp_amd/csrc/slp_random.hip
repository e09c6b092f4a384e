// slp_random.hip -- the benchmark distribution of randomLP.py:14-75, restated as
// a sparse, scalable, counter-based generator that runs on the device (the
// reference materialises dense (n_ineq x nbvar) arrays, randomLP.py:17-19,37,
// which cannot exist at 1e6 x 2e6).
//
//   rand_sparse(shape, p): value round(N(0,1)*100)/100 where rand() < p, else 0;
//   csr_matrix() then drops the exact zeros.
// Row r: the positions with rand() < p are a Bernoulli process, sampled here by
// geometric gaps (col += 1 + floor(ln u / ln(1-p))): same distribution, O(nnz).
// Every random number is a hash of (seed, stream, row/index, counter), so any
// row block can be regenerated independently on any GPU.
#include <cstring>
#include <cstdlib>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
// uniform in (0,1), 53 bits
__device__ __forceinline__ double u01(uint64_t seed, uint64_t stream, uint64_t index, uint64_t counter) {
    uint64_t h = mix64(seed ^ mix64(stream * 0x632be59bd9b4e019ull + 0x1234567ull));
    h = mix64(h ^ mix64(index));
    h = mix64(h ^ (counter * 0xd1342543de82ef95ull));
    return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double gauss(double u1, double u2) {
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
__device__ __forceinline__ double round2(double z) { return rint(z * 100.0) / 100.0; }  // np.round(x*100)/100

enum { STREAM_A = 1, STREAM_XF = 2, STREAM_C = 3, STREAM_T = 4, STREAM_B = 5 };

// One thread per row.  FILL == false: count the kept entries; FILL == true: write them.
template <bool FILL>
__global__ void k_random_rows(i64 nrow, i64 ncol, double inv_log1mp, uint64_t seed, i64 row_offset, i64 *__restrict__ len,
                              const i64 *__restrict__ ptr, i32 *__restrict__ idx, double *__restrict__ val) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        const uint64_t grow = (uint64_t)(r + row_offset);
        i64 col = -1, kept = 0;
        i64 out = FILL ? ptr[r] : 0;
        for (uint64_t e = 0;; ++e) {
            const double ug = u01(seed, STREAM_A, grow, 3 * e);
            const double gap = floor(log(ug) * inv_log1mp);
            if (gap >= (double)(ncol - col)) break;
            col += 1 + (i64)gap;
            if (col >= ncol) break;
            const double v = round2(gauss(u01(seed, STREAM_A, grow, 3 * e + 1), u01(seed, STREAM_A, grow, 3 * e + 2)));
            if (v != 0.0) {
                if (FILL) {
                    idx[out] = (i32)col;
                    val[out] = v;
                    ++out;
                }
                ++kept;
            }
        }
        if (!FILL) len[r] = kept;
    }
}

// The same rows, one WAVE per row (rows of many entries): every random number is a function of (row, event), so the 64 events
// of a chunk are drawn by the 64 lanes at once, the columns are a prefix sum of the gaps over the wave, and the kept entries
// are written side by side -- with a thread per row every lane wrote 4 + 8 bytes into a line of its own per event, and each
// line went to memory several times before it was full (PMC: 164 GB written for the 24 GB of config 3).
// The sequential loop stops at the first event whose column is past the end; columns grow with the event number, so that
// is the first lane (of the first chunk) with column >= ncol.  Gaps are clamped to ncol before they are added: no overflow.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_random_rows_wave(i64 nrow, i64 ncol, double inv_log1mp, uint64_t seed, i64 row_offset,
                                                             i64 *__restrict__ len, const i64 *__restrict__ ptr, i32 *__restrict__ idx,
                                                             double *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 r = wave; r < nrow; r += waves) {
        const uint64_t grow = (uint64_t)(r + row_offset);
        i64 col = -1, kept = 0;                      // (uniform over the wave)
        i64 out = FILL ? ptr[r] : 0;
        for (uint64_t e0 = 0;; e0 += 64) {
            const uint64_t e = e0 + (uint64_t)lane;
            const double ug = u01(seed, STREAM_A, grow, 3 * e);
            double gap = floor(log(ug) * inv_log1mp);
            gap = gap < (double)ncol ? gap : (double)ncol;
            i64 step = 1 + (i64)gap;                 // inclusive prefix sum over the lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const i64 up = __shfl_up(step, d);
                if (lane >= d) step += up;
            }
            const i64 c = col + step;
            const bool inside = c < ncol;
            double v = 0.0;
            if (inside) v = round2(gauss(u01(seed, STREAM_A, grow, 3 * e + 1), u01(seed, STREAM_A, grow, 3 * e + 2)));
            const bool keep = inside && v != 0.0;
            const unsigned long long mask = __ballot(keep);
            if (FILL && keep) {
                const i64 o = out + __popcll(mask & ((1ull << lane) - 1ull));
                idx[o] = (i32)c;
                val[o] = v;
            }
            const int n = __popcll(mask);
            out += n;
            kept += n;
            if (__ballot(!inside) != 0ull) break;     // the row ends inside this chunk
            col = __shfl(c, 63);
        }
        if (!FILL && lane == 0) len[r] = kept;
    }
}

__global__ void k_random_cols(i64 n, uint64_t seed, double *__restrict__ xf, double *__restrict__ c, double *__restrict__ lb,
                              double *__restrict__ ub) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const uint64_t g = (uint64_t)j;
        const double x = round2(gauss(u01(seed, STREAM_XF, g, 0), u01(seed, STREAM_XF, g, 1)));  // :33
        const double cc = round2(gauss(u01(seed, STREAM_C, g, 0), u01(seed, STREAM_C, g, 1)));  // :51
        const double t = round2(gauss(u01(seed, STREAM_T, g, 0), u01(seed, STREAM_T, g, 1)));   // :53
        xf[j] = x;
        c[j] = cc;
        lb[j] = x + (t < 0.0 ? t : 0.0);  // :54
        ub[j] = x + (t > 0.0 ? t : 0.0);  // :55
    }
}

// b_upper = ceil((A x_f + |rand_sparse(m, p)|) * 1000) / 1000   (:43-46); the first m_eq rows are equalities: b_eq = A_e x_f (:63)
__global__ void k_random_bupper(i64 m, i64 m_eq, uint64_t seed, i64 row_offset, double density, const double *__restrict__ axf,
                                double *__restrict__ b) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        if (i < m_eq) { b[i] = axf[i]; continue; }
        const uint64_t g = (uint64_t)(i + row_offset);
        double extra = 0.0;
        if (u01(seed, STREAM_B, g, 0) < density)
            extra = fabs(round2(gauss(u01(seed, STREAM_B, g, 1), u01(seed, STREAM_B, g, 2))));
        b[i] = ceil((axf[i] + extra) * 1000.0) / 1000.0;
    }
}

}  // namespace slp

using namespace slp;

extern "C" {

slp_matrix *slp_matrix_random(int64_t nrow, int64_t ncol, double density, uint64_t seed, int64_t row_offset) {
    SLP_API_PTR({
        SLP_REQUIRE(nrow >= 0 && ncol > 0 && ncol < ((i64)1 << 31) && density > 0.0 && density < 1.0,
                    "slp_matrix_random: bad arguments");
        Phase ph("slp_matrix_random");
        hipStream_t st = ctx().stream;
        auto *m = new slp_matrix();
        try {
            CsrDev &a = m->a;
            a.nrow = nrow;
            a.ncol = ncol;
            a.ptr.alloc((size_t)nrow + 1);
            const double inv = 1.0 / log1p(-density);
            DevBuf<i64> len((size_t)nrow + 1);
            len.zero();
            const int grid = grid_for(nrow, kBlock, 16);
            // rows of 48 entries and more (expected): a wave per row, coalesced writes; SLP_RANDOM_WAVE=0/1 overrides (tests)
            const char *ew = getenv("SLP_RANDOM_WAVE");
            const bool by_wave = ew ? ew[0] == '1' : density * (double)ncol >= 48.0;
            const int wgrid = grid_for(nrow * 64, kBlock, 16);
            if (nrow && by_wave) {
                hipLaunchKernelGGL((k_random_rows_wave<false>), dim3(wgrid), dim3(kBlock), 0, st, nrow, ncol, inv, seed, row_offset, len.p,
                                   nullptr, nullptr, nullptr);
                SLP_HIP(hipGetLastError());
            } else if (nrow) {
                hipLaunchKernelGGL((k_random_rows<false>), dim3(grid), dim3(kBlock), 0, st, nrow, ncol, inv, seed, row_offset, len.p,
                                   nullptr, nullptr, nullptr);
                SLP_HIP(hipGetLastError());
            }
            size_t bytes = 0;
            SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, len.p, a.ptr.p, (i64)0, (size_t)nrow + 1, rocprim::plus<i64>(), st));
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, len.p, a.ptr.p, (i64)0, (size_t)nrow + 1, rocprim::plus<i64>(), st));
            i64 nnz = 0;
            SLP_HIP(hipMemcpyAsync(&nnz, a.ptr.p + nrow, sizeof(i64), hipMemcpyDeviceToHost, st));
            SLP_HIP(hipStreamSynchronize(st));
            a.nnz = nnz;
            a.idx.alloc((size_t)nnz);
            a.val.alloc((size_t)nnz);
            if (nrow && by_wave) {
                hipLaunchKernelGGL((k_random_rows_wave<true>), dim3(wgrid), dim3(kBlock), 0, st, nrow, ncol, inv, seed, row_offset, nullptr,
                                   a.ptr.p, a.idx.p, a.val.p);
                SLP_HIP(hipGetLastError());
            } else if (nrow) {
                hipLaunchKernelGGL((k_random_rows<true>), dim3(grid), dim3(kBlock), 0, st, nrow, ncol, inv, seed, row_offset, nullptr,
                                   a.ptr.p, a.idx.p, a.val.p);
                SLP_HIP(hipGetLastError());
            }
            SLP_HIP(hipStreamSynchronize(st));
            finish_stats(a);
        } catch (...) {
            delete m;
            throw;
        }
        return m;
    })
}

int slp_random_lp_vectors(slp_matrix *m, double density, uint64_t seed, int64_t row_offset, double *feasible_x, double *c,
                          double *lb, double *ub, double *b_upper) {
    return slp_random_lp_vectors_eq(m, density, seed, row_offset, 0, feasible_x, c, lb, ub, b_upper);
}

int slp_random_lp_vectors_eq(slp_matrix *m, double density, uint64_t seed, int64_t row_offset, int64_t m_eq, double *feasible_x,
                             double *c, double *lb, double *ub, double *b_upper) {
    SLP_API_INT({
        SLP_REQUIRE(m, "slp_random_lp_vectors: NULL matrix");
        SLP_REQUIRE(m_eq >= 0 && m_eq <= m->a.nrow, "slp_random_lp_vectors_eq: m_eq out of range");
        Phase ph("slp_random_lp_vectors");
        hipStream_t st = ctx().stream;
        const i64 n = m->a.ncol, rows = m->a.nrow;
        DevBuf<double> xf((size_t)n), dc((size_t)n), dl((size_t)n), du((size_t)n), ax((size_t)rows), db((size_t)rows);
        hipLaunchKernelGGL(k_random_cols, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, n, seed, xf.p, dc.p, dl.p, du.p);
        SLP_HIP(hipGetLastError());
        if (rows && b_upper) {
            // A x_f as the matrix's own product forms it: through the product copies where the matrix has (or qualifies for) them --
            // every row one chain of additions in storage order, scipy's csr_matvec of randomLP.py:43 -- else the CSR kernel in the
            // same order.  A chunked matrix is asked once, after the last append (all rows in one launch of the fused product).  The
            // order matters: x_f and the coefficients are multiples of 0.01, so A x_f * 1000 lies within rounding of a whole number in
            // a tenth of the rows and the ceiling below turns on the last bit there -- one order everywhere, and every chunking
            // of an LP draws the same b_upper.  (Round 1-4 took it chunk by chunk from the CSR with the rows spread over 64 lanes:
            // every x gathered from L2, 11 x the CSR's bytes in HBM traffic at 1e7 columns, 45 ms per 2.5e9 entries against 3.)
            matrix_spmv(m, false, xf.p, ax.p, SLP_ORDER_SEQUENTIAL);
            hipLaunchKernelGGL(k_random_bupper, dim3(grid_for(rows, kBlock)), dim3(kBlock), 0, st, rows, (i64)m_eq, seed, row_offset,
                               density, ax.p, db.p);
            SLP_HIP(hipGetLastError());
        }
        if (feasible_x) xf.download(feasible_x, (size_t)n);
        if (c) dc.download(c, (size_t)n);
        if (lb) dl.download(lb, (size_t)n);
        if (ub) du.download(ub, (size_t)n);
        if (b_upper) db.download(b_upper, (size_t)rows);
        SLP_HIP(hipStreamSynchronize(st));
    })
}

}  // extern "C"
