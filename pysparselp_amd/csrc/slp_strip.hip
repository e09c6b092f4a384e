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
// streams the cell's entries with perfectly coalesced loads and gathers x from LDS.
//
// Layout of one cell (row block b, strip t), all cells back to back, b-major:
//   rows of the cell are sorted by their entry count in the strip (descending),
//   `perm[p]` = local row at sorted position p, `len[p]` = its count;
//   jagged diagonal s holds the s-th entry of every row with len > s, in sorted
//   order, padded to an even number of entries, at offset soff[s] inside the cell.
//   Entry = (fp64 value, uint16 column inside the strip): 10 B instead of 12 B.
// Thread p of the 1024 owns sorted positions 2p and 2p+1: one 16-byte load brings
// the s-th value of both rows, one 4-byte load both columns.
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

constexpr int kStripC = 7680;    // columns per strip: 60 KB of x in LDS
constexpr int kStripT = 1024;    // threads per workgroup
constexpr int kStripR = 2048;    // rows per block (two per thread)
constexpr int kStripSL = 256;    // slots (max entries of one row inside one strip)
static_assert(kStripC % 2 == 0 && kStripR == 2 * kStripT, "strip geometry");

// ---- conversion -------------------------------------------------------------
// pass 1: per (block, strip): every row's entry count (uint8) and the padded cell size
//         sum_s even(cnt[s]),  cnt[s] = rows of the cell with more than s entries
__global__ __launch_bounds__(kStripT) void k_strip_count(i64 nrow, i64 T, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                         unsigned char *__restrict__ len, unsigned long long *__restrict__ total,
                                                         int *__restrict__ bad) {
    __shared__ unsigned int hist[kStripSL];
    const i64 b = blockIdx.x;
    i64 k[2], e[2];
    i32 prev[2] = {-1, -1};
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + threadIdx.x;
        k[h] = e[h] = 0;
        if (row < nrow) { k[h] = ptr[row]; e[h] = ptr[row + 1]; }
    }
    for (i64 t = 0; t < T; ++t) {
        if (threadIdx.x < kStripSL) hist[threadIdx.x] = 0;
        __syncthreads();
        const i64 hi = (t + 1) * (i64)kStripC;
        for (int h = 0; h < 2; ++h) {
            i64 c = 0;
            while (k[h] < e[h]) {
                const i32 j = idx[k[h]];
                if (j >= hi) break;
                if (j <= prev[h]) atomicOr(bad, 1);  // rows must be strictly increasing in column
                prev[h] = j;
                ++k[h]; ++c;
            }
            if (c >= kStripSL) { atomicOr(bad, 2); c = kStripSL - 1; }
            len[(b * T + t) * kStripR + h * kStripT + threadIdx.x] = (unsigned char)c;
            atomicAdd(&hist[c], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int above = 0;  // rows with a count > l
            unsigned long long sum = 0;
            for (int l = kStripSL - 1; l >= 0; --l) {
                sum += (above + 1u) & ~1u;  // slot l holds `above` entries, padded to even (0 stays 0)
                above += hist[l];
            }
            total[b * T + t] = sum;
        }
        __syncthreads();
    }
}

// pass 2: sort the rows of every cell by count, write perm / sorted len / slot offsets and the entries
// (the output arrays are zero-filled beforehand: the pad entries stay (0.0, column 0))
__global__ __launch_bounds__(kStripT) void k_strip_fill(i64 nrow, i64 T, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                        const double *__restrict__ val, const unsigned char *__restrict__ len,
                                                        const i64 *__restrict__ base, unsigned short *__restrict__ perm,
                                                        unsigned char *__restrict__ slen, unsigned int *__restrict__ soff,
                                                        double *__restrict__ oval, unsigned short *__restrict__ ocol) {
    __shared__ unsigned int hist[kStripSL];   // rows with exactly this count, then the per-count cursor
    __shared__ unsigned int start[kStripSL];  // rows with a larger count  (= first sorted position of this count)
    __shared__ unsigned int offs[kStripSL];   // slot offsets
    const i64 b = blockIdx.x;
    i64 k[2];
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + threadIdx.x;
        k[h] = (row < nrow) ? ptr[row] : 0;
    }
    for (i64 t = 0; t < T; ++t) {
        const i64 cell = b * T + t;
        if (threadIdx.x < kStripSL) hist[threadIdx.x] = 0;
        __syncthreads();
        unsigned int c[2];
        for (int h = 0; h < 2; ++h) {
            c[h] = len[cell * kStripR + h * kStripT + threadIdx.x];
            atomicAdd(&hist[c[h]], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int above = 0;
            for (int l = kStripSL - 1; l >= 0; --l) { start[l] = above; above += hist[l]; }
            unsigned int o = 0;  // cnt[s] = start[s]; slot s occupies even(cnt[s]) entries
            for (int s = 0; s < kStripSL; ++s) { offs[s] = o; o += (start[s] + 1u) & ~1u; }
        }
        __syncthreads();
        if (threadIdx.x < kStripSL) {
            soff[cell * kStripSL + threadIdx.x] = offs[threadIdx.x];
            hist[threadIdx.x] = 0;
        }
        __syncthreads();
        const i64 bs = base[cell];
        const i32 col0 = (i32)(t * (i64)kStripC);
        for (int h = 0; h < 2; ++h) {
            const unsigned int pos = start[c[h]] + atomicAdd(&hist[c[h]], 1u);
            perm[cell * kStripR + pos] = (unsigned short)(h * kStripT + threadIdx.x);
            slen[cell * kStripR + pos] = (unsigned char)c[h];
            for (unsigned int s = 0; s < c[h]; ++s) {
                const i64 o = bs + offs[s] + pos;
                oval[o] = val[k[h] + s];
                ocol[o] = (unsigned short)(idx[k[h] + s] - col0);
            }
            k[h] += c[h];
        }
        __syncthreads();
    }
}

// ---- the product ---------------------------------------------------------------
// One workgroup per row block; 60 KB x-tile + 16 KB running sums + 1 KB slot offsets of LDS
// (two workgroups per CU).  out[row] = sum over the row (single accumulator, storage order).
// ABLATE (timing experiments only, wrong results): 1 = no x-tile staging, 2 = no entry streaming
template <int ABLATE>
__global__ __launch_bounds__(kStripT, 8) void k_strip_spmv(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
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
    acc[p + kStripT] = 0.0;
    // gridDim.y > 1: this workgroup covers only its share of the strips and writes partial sums
    const i64 t_begin = (T * (i64)blockIdx.y) / gridDim.y, t_end = (T * (i64)(blockIdx.y + 1)) / gridDim.y;
    for (i64 t = t_begin; t < t_end; ++t) {
        const i64 cell = b * T + t;
        // stage the strip of x with 16-byte loads
        const i64 c0 = t * (i64)kStripC;
#pragma unroll
        for (int q = 0; q < (ABLATE == 1 ? 0 : (kStripC / 2 + kStripT - 1) / kStripT); ++q) {
            const int j = (q * kStripT + p) * 2;
            if (j < kStripC) {
                double2 v = make_double2(0.0, 0.0);
                if (c0 + j + 1 < ncol) v = *reinterpret_cast<const double2 *>(x + c0 + j);
                else if (c0 + j < ncol) v.x = x[c0 + j];
                *reinterpret_cast<double2 *>(&xt[j]) = v;
            }
        }
        if (p < kStripSL) offs[p] = soff[cell * kStripSL + p];
        const ushort2 r = reinterpret_cast<const ushort2 *>(perm + cell * kStripR)[p];
        const uchar2 nn = reinterpret_cast<const uchar2 *>(slen + cell * kStripR)[p];
        const unsigned int n0 = (ABLATE == 2) ? 0u : nn.x, n1 = (ABLATE == 2) ? 0u : nn.y;  // n0 >= n1 (sorted)
        const double2 *__restrict__ v2 = reinterpret_cast<const double2 *>(val + base[cell]);
        const ushort2 *__restrict__ c2 = reinterpret_cast<const ushort2 *>(col + base[cell]);
        __syncthreads();
        double a0 = acc[r.x], a1 = acc[r.y];
        unsigned int s = 0;
        for (; s + 4 <= n0; s += 4) {  // four independent 16-byte + 4-byte loads in flight per lane
            const unsigned int o0 = (offs[s] >> 1) + p, o1 = (offs[s + 1] >> 1) + p, o2 = (offs[s + 2] >> 1) + p,
                               o3 = (offs[s + 3] >> 1) + p;
            const double2 w0 = v2[o0], w1 = v2[o1], w2 = v2[o2], w3 = v2[o3];
            const ushort2 j0 = c2[o0], j1 = c2[o1], j2 = c2[o2], j3 = c2[o3];
            a0 += w0.x * xt[j0.x];
            a0 += w1.x * xt[j1.x];
            a0 += w2.x * xt[j2.x];
            a0 += w3.x * xt[j3.x];
            if (s < n1) a1 += w0.y * xt[j0.y];
            if (s + 1 < n1) a1 += w1.y * xt[j1.y];
            if (s + 2 < n1) a1 += w2.y * xt[j2.y];
            if (s + 3 < n1) a1 += w3.y * xt[j3.y];
        }
        for (; s < n0; ++s) {
            const unsigned int o = (offs[s] >> 1) + p;
            const double2 w = v2[o];
            const ushort2 j = c2[o];
            a0 += w.x * xt[j.x];
            if (s < n1) a1 += w.y * xt[j.y];
        }
        acc[r.x] = a0;
        acc[r.y] = a1;
        __syncthreads();
    }
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        if (row < nrow) out[(i64)blockIdx.y * nrow + row] = acc[h * kStripT + p];
    }
}


// Two right-hand sides in one pass over the matrix: out0 = A x0, out1 = A x1.  The matrix (the only
// HBM-sized operand) is streamed once; both x-tiles and both sets of running sums live in LDS (153 KB,
// one workgroup per CU).  Per row and per vector the additions are the same chain as in k_strip_spmv,
// so each output is bit-identical to the single-vector product.
__global__ __launch_bounds__(kStripT, 4) void k_strip_spmv2(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
                                                            const unsigned short *__restrict__ perm,
                                                            const unsigned char *__restrict__ slen,
                                                            const unsigned int *__restrict__ soff, const double *__restrict__ val,
                                                            const unsigned short *__restrict__ col, const double *__restrict__ x0,
                                                            const double *__restrict__ x1, double *__restrict__ out0,
                                                            double *__restrict__ out1) {
    __shared__ double xt0[kStripC];
    __shared__ double xt1[kStripC];
    __shared__ double acc0[kStripR];
    __shared__ double acc1[kStripR];
    __shared__ unsigned int offs[kStripSL];
    const i64 b = blockIdx.x;
    const int p = threadIdx.x;
    acc0[p] = 0.0; acc0[p + kStripT] = 0.0;
    acc1[p] = 0.0; acc1[p + kStripT] = 0.0;
    const i64 t_begin = (T * (i64)blockIdx.y) / gridDim.y, t_end = (T * (i64)(blockIdx.y + 1)) / gridDim.y;
    for (i64 t = t_begin; t < t_end; ++t) {
        const i64 cell = b * T + t;
        const i64 c0 = t * (i64)kStripC;
#pragma unroll
        for (int q = 0; q < (kStripC / 2 + kStripT - 1) / kStripT; ++q) {
            const int j = (q * kStripT + p) * 2;
            if (j < kStripC) {
                double2 v = make_double2(0.0, 0.0), u = make_double2(0.0, 0.0);
                if (c0 + j + 1 < ncol) {
                    v = *reinterpret_cast<const double2 *>(x0 + c0 + j);
                    u = *reinterpret_cast<const double2 *>(x1 + c0 + j);
                } else if (c0 + j < ncol) {
                    v.x = x0[c0 + j];
                    u.x = x1[c0 + j];
                }
                *reinterpret_cast<double2 *>(&xt0[j]) = v;
                *reinterpret_cast<double2 *>(&xt1[j]) = u;
            }
        }
        if (p < kStripSL) offs[p] = soff[cell * kStripSL + p];
        const ushort2 r = reinterpret_cast<const ushort2 *>(perm + cell * kStripR)[p];
        const uchar2 nn = reinterpret_cast<const uchar2 *>(slen + cell * kStripR)[p];
        const unsigned int n0 = nn.x, n1 = nn.y;
        const double2 *__restrict__ v2 = reinterpret_cast<const double2 *>(val + base[cell]);
        const ushort2 *__restrict__ c2 = reinterpret_cast<const ushort2 *>(col + base[cell]);
        __syncthreads();
        double a0 = acc0[r.x], a1 = acc0[r.y], b0 = acc1[r.x], b1 = acc1[r.y];
        unsigned int s = 0;
        for (; s + 4 <= n0; s += 4) {
            const unsigned int o0 = (offs[s] >> 1) + p, o1 = (offs[s + 1] >> 1) + p, o2 = (offs[s + 2] >> 1) + p,
                               o3 = (offs[s + 3] >> 1) + p;
            const double2 w0 = v2[o0], w1 = v2[o1], w2 = v2[o2], w3 = v2[o3];
            const ushort2 j0 = c2[o0], j1 = c2[o1], j2 = c2[o2], j3 = c2[o3];
            a0 += w0.x * xt0[j0.x]; b0 += w0.x * xt1[j0.x];
            a0 += w1.x * xt0[j1.x]; b0 += w1.x * xt1[j1.x];
            a0 += w2.x * xt0[j2.x]; b0 += w2.x * xt1[j2.x];
            a0 += w3.x * xt0[j3.x]; b0 += w3.x * xt1[j3.x];
            if (s < n1) { a1 += w0.y * xt0[j0.y]; b1 += w0.y * xt1[j0.y]; }
            if (s + 1 < n1) { a1 += w1.y * xt0[j1.y]; b1 += w1.y * xt1[j1.y]; }
            if (s + 2 < n1) { a1 += w2.y * xt0[j2.y]; b1 += w2.y * xt1[j2.y]; }
            if (s + 3 < n1) { a1 += w3.y * xt0[j3.y]; b1 += w3.y * xt1[j3.y]; }
        }
        for (; s < n0; ++s) {
            const unsigned int o = (offs[s] >> 1) + p;
            const double2 w = v2[o];
            const ushort2 j = c2[o];
            a0 += w.x * xt0[j.x]; b0 += w.x * xt1[j.x];
            if (s < n1) { a1 += w.y * xt0[j.y]; b1 += w.y * xt1[j.y]; }
        }
        acc0[r.x] = a0; acc0[r.y] = a1;
        acc1[r.x] = b0; acc1[r.y] = b1;
        __syncthreads();
    }
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        if (row < nrow) {
            out0[(i64)blockIdx.y * nrow + row] = acc0[h * kStripT + p];
            out1[(i64)blockIdx.y * nrow + row] = acc1[h * kStripT + p];
        }
    }
}

// out[row] = ((part[0][row] + part[1][row]) + ...) in strip order (deterministic)
__global__ void k_strip_combine(i64 nrow, int S, const double *__restrict__ part, double *__restrict__ out) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        double a = part[r];
        for (int s = 1; s < S; ++s) a += part[(i64)s * nrow + r];
        out[r] = a;
    }
}

// ---- host side ------------------------------------------------------------------
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
    hipLaunchKernelGGL(k_strip_count, dim3((unsigned)B), dim3(kStripT), 0, st, a.nrow, T, a.ptr.p, a.idx.p, len.p, total.p, bad.p);
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
    i64 padded = 0;
    SLP_HIP(hipMemcpyAsync(&padded, f.base.p + cells, sizeof(i64), hipMemcpyDeviceToHost, st));
    SLP_HIP(hipStreamSynchronize(st));
    f.perm.alloc(cells * kStripR);
    f.slen.alloc(cells * kStripR);
    f.soff.alloc(cells * kStripSL);
    f.val.alloc((size_t)padded);
    f.col.alloc((size_t)padded);
    f.val.zero();
    f.col.zero();
    hipLaunchKernelGGL(k_strip_fill, dim3((unsigned)B), dim3(kStripT), 0, st, a.nrow, T, a.ptr.p, a.idx.p, a.val.p, len.p, f.base.p,
                       f.perm.p, f.slen.p, f.soff.p, f.val.p, f.col.p);
    SLP_HIP(hipGetLastError());
    SLP_HIP(hipStreamSynchronize(st));
    f.nrow = a.nrow; f.ncol = a.ncol; f.nnz = a.nnz; f.T = T; f.B = B;
    // Few row blocks (a 1/4 or 1/8 row partition of the constraints): split every block's strips over S
    // workgroups so that the launch still fills the 256 CUs x 2 resident workgroups.
    const char *es = getenv("SLP_STRIP_SPLIT");
    int S = es ? atoi(es) : 1;
    if (!es && a.nnz >= 30000000) while (S < 8 && B * S < 384 && 2 * S <= T) S *= 2;
    if (S < 1) S = 1;
    if (S > T) S = (int)T;
    f.S = S;
    if (S > 1) f.part.alloc((size_t)S * (size_t)a.nrow);
    f.ok = true;
    return true;
}

void strip_spmv(const StripJds &f, const double *x, double *out) {
    const char *e = getenv("SLP_STRIP_ABLATE");
    const int ab = e ? atoi(e) : 0;
#define SLP_STRIP_LAUNCH(A)                                                                                                        \
    hipLaunchKernelGGL((k_strip_spmv<A>), dim3((unsigned)f.B, (unsigned)f.S), dim3(kStripT), 0, ctx().stream, f.nrow, f.ncol, f.T, \
                       f.base.p, f.perm.p, f.slen.p, f.soff.p, f.val.p, f.col.p, x, f.S > 1 ? f.part.p : out)
    if (ab == 1) SLP_STRIP_LAUNCH(1);
    else if (ab == 2) SLP_STRIP_LAUNCH(2);
    else SLP_STRIP_LAUNCH(0);
#undef SLP_STRIP_LAUNCH
    if (f.S > 1)
        hipLaunchKernelGGL(k_strip_combine, dim3(grid_for(f.nrow, kBlock)), dim3(kBlock), 0, ctx().stream, f.nrow, f.S, f.part.p, out);
    SLP_HIP(hipGetLastError());
}

void strip_spmv2(const StripJds &f, const double *x0, const double *x1, double *out0, double *out1) {
    hipStream_t st = ctx().stream;
    double *o0 = out0, *o1 = out1;
    if (f.S > 1) {
        if (f.part2.n < 2 * (size_t)f.S * (size_t)f.nrow) f.part2.alloc(2 * (size_t)f.S * (size_t)f.nrow);
        o0 = f.part2.p;
        o1 = f.part2.p + (size_t)f.S * (size_t)f.nrow;
    }
    hipLaunchKernelGGL(k_strip_spmv2, dim3((unsigned)f.B, (unsigned)f.S), dim3(kStripT), 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p,
                       f.slen.p, f.soff.p, f.val.p, f.col.p, x0, x1, o0, o1);
    if (f.S > 1) {
        hipLaunchKernelGGL(k_strip_combine, dim3(grid_for(f.nrow, kBlock)), dim3(kBlock), 0, st, f.nrow, f.S, o0, out0);
        hipLaunchKernelGGL(k_strip_combine, dim3(grid_for(f.nrow, kBlock)), dim3(kBlock), 0, st, f.nrow, f.S, o1, out1);
    }
    SLP_HIP(hipGetLastError());
}

// Does the format pay?  Long rows (the gather-bound regime) and enough entries per (row, strip) to amortise
// the 3 bytes of per-(row, strip) metadata and the x-tile staging.
bool strip_wanted(const CsrDev &a) {
    const char *e = getenv("SLP_STRIP_MIN_NNZ");  // below this size launch latency, not the gathers, dominates
    const i64 min_nnz = e ? atoll(e) : 30000000ll;  // measured cross-over vs the CSR kernel (tools/strip_threshold.py)
    if (a.nnz < min_nnz) return false;
    const double per_cell = a.mean_row_len() / (double)((a.ncol + kStripC - 1) / kStripC);
    return per_cell >= 3.0 && per_cell <= 64.0;
}

}  // namespace slp
