// slp_strip.hip -- the bandwidth-bound SpMV for long rows: column strips with the
// x-tile staged in LDS, rows stored as jagged diagonals inside every
// (row block, strip) cell ("strip-JDS").
//
// Why: with ~1000 stored entries per row and uniformly random columns, the plain
// CSR kernel gathers x[j] from an 8-16 MB vector 2e9 times per product; every
// gather is a separate cache-line request to L2 / Infinity Cache and the kernel
// runs at the fabric's line rate (measured: 1.1 TB/s algorithmic, 14 % of HBM
// peak) instead of streaming the matrix.  Here the matrix is cut into vertical
// strips of C columns; a workgroup owns R rows, keeps their R running sums in
// LDS, and walks the strips: it stages x[strip] (C doubles) in LDS once and then
// streams the cell's entries with perfectly coalesced loads -- lane p reads
// entry `slot_offset[s] + p` for s = 0, 1, ... -- and gathers x from LDS.
//
// Layout of one cell (row block b, strip t), all cells back to back, b-major:
//   rows of the cell are sorted by their entry count in the strip (descending),
//   `perm[p]` = local row at sorted position p, `len[p]` = its count;
//   jagged diagonal s holds the s-th entry of every row with len > s, in sorted
//   order, so diagonal s has cnt[s] entries at offset soff[s] = sum_{s'<s} cnt[s'].
//   Entry = (fp64 value, uint16 column inside the strip): 10 B instead of 12 B.
// A row's entries keep their column order across and inside strips, and each
// row is accumulated by one thread at a time starting from its running sum, so
// the result equals the SEQUENTIAL single-accumulator sum of the CSR row -- bit
// for bit (rows must be sorted by column, which holds for generated matrices
// and for every device-built transpose).
#include <cstring>
#include <cstdlib>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

constexpr int kStripC = 8192;    // columns per strip: 64 KB of x in LDS
constexpr int kStripR = 1024;    // rows per block = threads per workgroup
constexpr int kStripSL = 256;    // slots (max entries of one row inside one strip)

// ---- conversion -------------------------------------------------------------
// pass 1: per (block, strip): every row's entry count (uint8) and the cell total
__global__ __launch_bounds__(kStripR) void k_strip_count(i64 nrow, i64 T, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                         unsigned char *__restrict__ len, unsigned long long *__restrict__ total,
                                                         int *__restrict__ bad) {
    __shared__ unsigned long long lds[kStripR / kWave];
    const i64 b = blockIdx.x;
    const i64 row = b * kStripR + threadIdx.x;
    i64 k = 0, e = 0;
    if (row < nrow) { k = ptr[row]; e = ptr[row + 1]; }
    i32 prev = -1;
    for (i64 t = 0; t < T; ++t) {
        const i64 hi = (t + 1) * (i64)kStripC;
        i64 c = 0;
        while (k < e) {
            const i32 j = idx[k];
            if (j >= hi) break;
            if (j <= prev) atomicOr(bad, 1);  // rows must be strictly increasing in column
            prev = j;
            ++k; ++c;
        }
        if (c >= kStripSL) { atomicOr(bad, 2); c = kStripSL - 1; }
        len[(b * T + t) * kStripR + threadIdx.x] = (unsigned char)c;
        // workgroup sum of c
        unsigned long long v = (unsigned long long)c;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long s = 0;
            for (int i = 0; i < kStripR / kWave; ++i) s += lds[i];
            total[b * T + t] = s;
        }
        __syncthreads();
    }
}

// pass 2: sort the rows of every cell by count, write perm / sorted len / slot offsets and the entries
__global__ __launch_bounds__(kStripR) void k_strip_fill(i64 nrow, i64 T, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                        const double *__restrict__ val, const unsigned char *__restrict__ len,
                                                        const i64 *__restrict__ base, unsigned short *__restrict__ perm,
                                                        unsigned char *__restrict__ slen, unsigned int *__restrict__ soff,
                                                        double *__restrict__ oval, unsigned short *__restrict__ ocol) {
    __shared__ unsigned int hist[kStripSL];   // rows with exactly this count
    __shared__ unsigned int start[kStripSL];  // rows with a larger count  (= first sorted position of this count)
    __shared__ unsigned int offs[kStripSL];   // slot offsets
    const i64 b = blockIdx.x;
    const i64 row = b * kStripR + threadIdx.x;
    i64 k = (row < nrow) ? ptr[row] : 0;
    for (i64 t = 0; t < T; ++t) {
        const i64 cell = b * T + t;
        if (threadIdx.x < kStripSL) hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned int c = len[cell * kStripR + threadIdx.x];
        atomicAdd(&hist[c], 1u);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int above = 0;
            for (int l = kStripSL - 1; l >= 0; --l) { start[l] = above; above += hist[l]; }
            // cnt[s] = rows with count > s = start[s];  soff[s] = sum_{s' < s} cnt[s']
            unsigned int o = 0;
            for (int s = 0; s < kStripSL; ++s) { offs[s] = o; o += start[s]; }
        }
        __syncthreads();
        if (threadIdx.x < kStripSL) {
            soff[cell * kStripSL + threadIdx.x] = offs[threadIdx.x];
            hist[threadIdx.x] = 0;  // reused as the per-count cursor
        }
        __syncthreads();
        const unsigned int pos = start[c] + atomicAdd(&hist[c], 1u);
        perm[cell * kStripR + pos] = (unsigned short)threadIdx.x;
        slen[cell * kStripR + pos] = (unsigned char)c;
        const i64 bs = base[cell];
        const i32 col0 = (i32)(t * (i64)kStripC);
        for (unsigned int s = 0; s < c; ++s) {
            const i64 o = bs + offs[s] + pos;
            oval[o] = val[k + s];
            ocol[o] = (unsigned short)(idx[k + s] - col0);
        }
        k += c;
        __syncthreads();
    }
}

// ---- the product ---------------------------------------------------------------
// One workgroup per row block; 64 KB x-tile + 8 KB running sums + 1 KB slot offsets of LDS
// (two workgroups per CU).  out[row] = sum over the row (single accumulator, storage order).
// ABLATE (timing experiments only, wrong results): 1 = no x-tile staging, 2 = no entry streaming
template <int ABLATE>
__global__ __launch_bounds__(kStripR, 8) void k_strip_spmv(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
                                                        const unsigned short *__restrict__ perm,
                                                        const unsigned char *__restrict__ slen,
                                                        const unsigned int *__restrict__ soff, const double *__restrict__ val,
                                                        const unsigned short *__restrict__ col, const double *__restrict__ x,
                                                        double *__restrict__ out) {
    __shared__ double xt[kStripC];
    __shared__ double acc[kStripR];
    __shared__ unsigned int offs[kStripSL];
    const i64 b = blockIdx.x;
    const int p = threadIdx.x;
    acc[p] = 0.0;
    for (i64 t = 0; t < T; ++t) {
        const i64 cell = b * T + t;
        // stage the strip of x: 8 doubles per thread, 16-byte loads
        const i64 c0 = t * (i64)kStripC;
#pragma unroll
        for (int q = 0; q < (ABLATE == 1 ? 0 : kStripC / kStripR / 2); ++q) {
            const int j = (q * kStripR + p) * 2;
            double2 v = make_double2(0.0, 0.0);
            if (c0 + j + 1 < ncol) v = *reinterpret_cast<const double2 *>(x + c0 + j);
            else if (c0 + j < ncol) v.x = x[c0 + j];
            *reinterpret_cast<double2 *>(&xt[j]) = v;
        }
        if (p < kStripSL) offs[p] = soff[cell * kStripSL + p];
        const unsigned int r = perm[cell * kStripR + p];
        const unsigned int n = (ABLATE == 2) ? 0u : slen[cell * kStripR + p];
        const double *__restrict__ v = val + base[cell];
        const unsigned short *__restrict__ c = col + base[cell];
        __syncthreads();
        double a = acc[r];
        unsigned int s = 0;
        for (; s + 4 <= n; s += 4) {  // four independent loads in flight per lane
            const unsigned int o0 = offs[s] + p, o1 = offs[s + 1] + p, o2 = offs[s + 2] + p, o3 = offs[s + 3] + p;
            const double v0 = v[o0], v1 = v[o1], v2 = v[o2], v3 = v[o3];
            const unsigned int j0 = c[o0], j1 = c[o1], j2 = c[o2], j3 = c[o3];
            a += v0 * xt[j0];
            a += v1 * xt[j1];
            a += v2 * xt[j2];
            a += v3 * xt[j3];
        }
        for (; s < n; ++s) {
            const unsigned int o = offs[s] + p;
            a += v[o] * xt[c[o]];
        }
        acc[r] = a;
        __syncthreads();
    }
    const i64 row = b * kStripR + p;
    if (row < nrow) out[row] = acc[p];
}

// ---- host side ------------------------------------------------------------------
void strip_release(StripJds &f) { f = StripJds(); }

// Builds the strip format of `a` (rows sorted by column).  Returns false (and leaves f.ok == false)
// when the matrix does not qualify: unsorted rows, or a row with >= 256 entries inside one strip.
bool strip_build(const CsrDev &a, StripJds &f) {
    hipStream_t st = ctx().stream;
    f = StripJds();
    if (a.nrow == 0 || a.nnz == 0) return false;
    const i64 T = (a.ncol + kStripC - 1) / kStripC, B = (a.nrow + kStripR - 1) / kStripR;
    const size_t cells = (size_t)(B * T);
    DevBuf<unsigned char> len(cells * kStripR);
    DevBuf<unsigned long long> total(cells + 1);
    DevBuf<int> bad(1);
    total.zero();
    bad.zero();
    hipLaunchKernelGGL(k_strip_count, dim3((unsigned)B), dim3(kStripR), 0, st, a.nrow, T, a.ptr.p, a.idx.p, len.p, total.p, bad.p);
    SLP_HIP(hipGetLastError());
    int hbad = 0;
    bad.download(&hbad, 1);
    if (hbad) return false;
    f.base.alloc(cells + 1);
    {
        size_t bytes = 0;
        SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, total.p, (unsigned long long *)f.base.p, 0ull, cells + 1,
                                        rocprim::plus<unsigned long long>(), st));
        DevBuf<char> tmp(bytes);
        SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, total.p, (unsigned long long *)f.base.p, 0ull, cells + 1,
                                        rocprim::plus<unsigned long long>(), st));
        SLP_HIP(hipStreamSynchronize(st));
    }
    f.perm.alloc(cells * kStripR);
    f.slen.alloc(cells * kStripR);
    f.soff.alloc(cells * kStripSL);
    f.val.alloc((size_t)a.nnz);
    f.col.alloc((size_t)a.nnz);
    hipLaunchKernelGGL(k_strip_fill, dim3((unsigned)B), dim3(kStripR), 0, st, a.nrow, T, a.ptr.p, a.idx.p, a.val.p, len.p, f.base.p,
                       f.perm.p, f.slen.p, f.soff.p, f.val.p, f.col.p);
    SLP_HIP(hipGetLastError());
    SLP_HIP(hipStreamSynchronize(st));
    f.nrow = a.nrow; f.ncol = a.ncol; f.nnz = a.nnz; f.T = T; f.B = B;
    f.ok = true;
    return true;
}

void strip_spmv(const StripJds &f, const double *x, double *out) {
    const char *e = getenv("SLP_STRIP_ABLATE");
    const int ab = e ? atoi(e) : 0;
#define SLP_STRIP_LAUNCH(A)                                                                                                        \
    hipLaunchKernelGGL((k_strip_spmv<A>), dim3((unsigned)f.B), dim3(kStripR), 0, ctx().stream, f.nrow, f.ncol, f.T, f.base.p, f.perm.p, \
                       f.slen.p, f.soff.p, f.val.p, f.col.p, x, out)
    if (ab == 1) SLP_STRIP_LAUNCH(1);
    else if (ab == 2) SLP_STRIP_LAUNCH(2);
    else SLP_STRIP_LAUNCH(0);
#undef SLP_STRIP_LAUNCH
    SLP_HIP(hipGetLastError());
}

// Does the format pay?  Long rows (the gather-bound regime) and enough entries per (row, strip) to amortise
// the 3 bytes of per-(row, strip) metadata and the x-tile staging.
bool strip_wanted(const CsrDev &a) {
    const char *e = getenv("SLP_STRIP_MIN_NNZ");  // below this size launch latency, not the gathers, dominates
    const i64 min_nnz = e ? atoll(e) : 20000000ll;
    if (a.nnz < min_nnz) return false;
    const double per_cell = a.mean_row_len() / (double)((a.ncol + kStripC - 1) / kStripC);
    return per_cell >= 3.0 && per_cell <= 64.0;
}

}  // namespace slp
