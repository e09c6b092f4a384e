// slp_spgemm.hip -- device-side problem transforms of the ADMM / Chambolle-Pock setup:
//
//  * slp_matrix_normal: M = gamma_eq A^T A + gamma_ineq I (ADMM.py:93-101), the matrix of the projected Gauss-Seidel
//    x-step, as a sparse-sparse product on the device.  The reference evaluates `a.T * a` with scipy's SMMP csr_matmat:
//    entry (i, j) accumulates A[k, j] * A[k, i] over the shared rows k in increasing k from 0.0, one rounding per product
//    and per add; entries whose sum is exactly 0 are dropped; `gamma_eq * a_t_a` scales every stored value; the sparse `+`
//    adds gamma_ineq on the diagonal (dropping exact zeros); `.tocsr()` leaves sorted rows.  (The reference's own M for
//    one LP is kept as a fixture, tests/golden/kernel_kats.npz; tests/test_gpu_spgemm.py compares bit for bit.)
//    Here: expand - sort - compress.  Every product is written with the 64-bit key (i << 32 | j) in the order the SMMP
//    loops produce it (columns i; inside a column the rows k of A increasingly; inside row k its entries in storage
//    order), a STABLE radix sort by key brings the terms of one entry together without changing their order, and one
//    thread sums each run sequentially -- the same chain of roundings as the reference, bit for bit.  One extra +0.0 term
//    per column at (i, i) makes the diagonal exist for empty columns (x + 0.0 == x for every x that survives the != 0 test).
//
//  * slp_matrix_remove_columns: the column compaction of SparseLP.remove_fixed_variables (SparseLP.py:632-674):
//    a[:, free] with the entries kept in storage order, plus A * shift for the right-hand sides.
#include <cstring>
#include <cstdlib>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

typedef unsigned long long u64;

// products (and the +0.0 marker) each column i of M generates
__global__ void k_nm_count(i64 ncol, const i64 *__restrict__ cptr, const i32 *__restrict__ crow, const i64 *__restrict__ ptr,
                           u64 *__restrict__ cnt) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < ncol; i += (i64)gridDim.x * blockDim.x) {
        u64 c = 1;
        for (i64 p = cptr[i]; p < cptr[i + 1]; ++p) {
            const i64 k = crow[p];
            c += (u64)(ptr[k + 1] - ptr[k]);
        }
        cnt[i] = c;
    }
}

__global__ void k_nm_expand(i64 ncol, const i64 *__restrict__ cptr, const i32 *__restrict__ crow, const double *__restrict__ cval,
                            const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const double *__restrict__ val,
                            const u64 *__restrict__ off, u64 *__restrict__ key, double *__restrict__ prod) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < ncol; i += (i64)gridDim.x * blockDim.x) {
        u64 o = off[i];
        const u64 hi = (u64)i << 32;
        for (i64 p = cptr[i]; p < cptr[i + 1]; ++p) {
            const i64 k = crow[p];
            const double aki = cval[p];
            for (i64 q = ptr[k]; q < ptr[k + 1]; ++q, ++o) {
                key[o] = hi | (u64)(unsigned int)idx[q];
                prod[o] = val[q] * aki;  // csr_matmat: sums[j] += A[k, j] * A^T[i, k]
            }
        }
        key[o] = hi | (u64)i;
        prod[o] = 0.0;
    }
}

// One thread per run of equal keys (sorted, stable): sequential sum from 0.0, then the gamma scaling / diagonal rule.
// keep[p] = 1 on the head of a run whose final value is non-zero.
__global__ void k_nm_reduce(u64 total, const u64 *__restrict__ key, const double *__restrict__ prod, double gamma_eq, double gamma_ineq,
                            double *__restrict__ value, u64 *__restrict__ keep) {
    for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (u64)gridDim.x * blockDim.x) {
        const u64 k = key[p];
        u64 flag = 0;
        if (p == 0 || key[p - 1] != k) {
            double ata = 0.0;
            for (u64 q = p; q < total && key[q] == k; ++q) ata += prod[q];
            const bool diag = (k >> 32) == (k & 0xffffffffull);
            double v;
            if (diag) v = (ata != 0.0) ? (gamma_eq * ata + gamma_ineq) : gamma_ineq;
            else v = (ata != 0.0) ? gamma_eq * ata : 0.0;
            value[p] = v;
            flag = (v != 0.0) ? 1 : 0;
        }
        keep[p] = flag;
    }
}

__global__ void k_nm_compact(u64 total, const u64 *__restrict__ key, const double *__restrict__ value, const u64 *__restrict__ keep,
                             const u64 *__restrict__ pos, unsigned int *__restrict__ orow, i32 *__restrict__ ocol, double *__restrict__ oval) {
    for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (u64)gridDim.x * blockDim.x) {
        if (!keep[p]) continue;
        const u64 o = pos[p];
        orow[o] = (unsigned int)(key[p] >> 32);
        ocol[o] = (i32)(key[p] & 0xffffffffull);
        oval[o] = value[p];
    }
}

__global__ void k_ptr_from_sorted_rows(i64 nnz, i64 nrow, const unsigned int *__restrict__ row, i64 *__restrict__ ptr) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (i64)gridDim.x * blockDim.x) {
        const i64 c = row[p], prev = p > 0 ? (i64)row[p - 1] : -1;
        for (i64 j = prev + 1; j <= c; ++j) ptr[j] = p;
        if (p == nnz - 1)
            for (i64 j = c + 1; j <= nrow; ++j) ptr[j] = nnz;
    }
}

template <class T>
static void exclusive_scan(const T *in, T *out, size_t count) {
    hipStream_t st = ctx().stream;
    size_t bytes = 0;
    SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (T)0, count, rocprim::plus<T>(), st));
    DevBuf<char> tmp(bytes);
    SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, (T)0, count, rocprim::plus<T>(), st));
    SLP_HIP(hipStreamSynchronize(st));
}

static slp_matrix *matrix_normal(slp_matrix *a, double gamma_eq, double gamma_ineq) {
    SLP_REQUIRE(a, "slp_matrix_normal: NULL matrix");
    require_csr(a, "slp_matrix_normal");
    Phase ph("slp_matrix_normal");
    hipStream_t st = ctx().stream;
    build_transpose(a);  // CSC of A, rows increasing inside every column (csr_tocsc order)
    const CsrDev &r = a->a, &c = a->at;
    const i64 N = r.ncol;
    auto *m = new slp_matrix();
    try {
        m->a.nrow = m->a.ncol = N;
        m->a.ptr.alloc((size_t)N + 1);
        DevBuf<u64> cnt((size_t)N + 1), off((size_t)N + 1);
        cnt.zero();
        if (N) hipLaunchKernelGGL(k_nm_count, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, st, N, c.ptr.p, c.idx.p, r.ptr.p, cnt.p);
        SLP_HIP(hipGetLastError());
        exclusive_scan(cnt.p, off.p, (size_t)N + 1);
        u64 total = 0;
        SLP_HIP(hipMemcpyAsync(&total, off.p + N, sizeof(u64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        if (total == 0) {  // N == 0 (every column counts its diagonal marker): an empty 0 x 0 matrix, nothing to sort or compact
            m->a.ptr.zero();
            m->a.nnz = 0;
            m->a.idx.alloc(0);
            m->a.val.alloc(0);
            finish_stats(m->a);
            SLP_HIP(hipStreamSynchronize(st));
            return m;
        }
        {  // expand-sort-compress holds every product twice (key + value, double-buffered by the sort): 32 bytes each
            size_t free_b = 0, total_b = 0;
            SLP_HIP(hipMemGetInfo(&free_b, &total_b));
            SLP_REQUIRE(total < ((u64)1 << 40) && (double)total * 40.0 < (double)free_b + (double)slp_cached_bytes(),
                        "slp_matrix_normal: " + std::to_string(total) + " products A[k,i] * A[k,j] do not fit the device -- M is not sparse "
                        "at this size (use the matrix-free conjugate-gradient x-step, lp_admm(xstep=\"cg\"))");
        }
        DevBuf<u64> key((size_t)total), key2((size_t)total);
        DevBuf<double> prod((size_t)total), prod2((size_t)total);
        if (N) hipLaunchKernelGGL(k_nm_expand, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, st, N, c.ptr.p, c.idx.p, c.val.p, r.ptr.p, r.idx.p,
                                  r.val.p, off.p, key.p, prod.p);
        SLP_HIP(hipGetLastError());
        unsigned int bits = 1;
        while (bits < 32 && ((i64)1 << bits) < N) ++bits;
        {
            size_t bytes = 0;
            SLP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key2.p, prod.p, prod2.p, (size_t)total, 0u, 32u + bits, st));
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key2.p, prod.p, prod2.p, (size_t)total, 0u, 32u + bits, st));
            SLP_HIP(hipStreamSynchronize(st));
        }
        // key / prod are free again: reuse them for the per-run values and the keep flags
        double *value = prod.p;
        u64 *keep = key.p;
        hipLaunchKernelGGL(k_nm_reduce, dim3(grid_for((i64)total, kBlock)), dim3(kBlock), 0, st, total, key2.p, prod2.p, gamma_eq, gamma_ineq,
                           value, keep);
        SLP_HIP(hipGetLastError());
        DevBuf<u64> pos((size_t)total + 1);
        exclusive_scan(keep, pos.p, (size_t)total);
        u64 last_pos = 0, last_keep = 0;
        SLP_HIP(hipMemcpyAsync(&last_pos, pos.p + (total - 1), sizeof(u64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipMemcpyAsync(&last_keep, keep + (total - 1), sizeof(u64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        const i64 nnz = (i64)(last_pos + last_keep);
        m->a.nnz = nnz;
        m->a.idx.alloc((size_t)nnz);
        m->a.val.alloc((size_t)nnz);
        DevBuf<unsigned int> orow((size_t)nnz);
        hipLaunchKernelGGL(k_nm_compact, dim3(grid_for((i64)total, kBlock)), dim3(kBlock), 0, st, total, key2.p, value, keep, pos.p, orow.p,
                           m->a.idx.p, m->a.val.p);
        hipLaunchKernelGGL(k_ptr_from_sorted_rows, dim3(grid_for(nnz, kBlock)), dim3(kBlock), 0, st, nnz, N, orow.p, m->a.ptr.p);
        SLP_HIP(hipGetLastError());
        SLP_HIP(hipStreamSynchronize(st));
        finish_stats(m->a);
    } catch (...) {
        delete m;
        throw;
    }
    return m;
}

// ---- column removal ------------------------------------------------------------------------------------------
__global__ void k_rc_count(i64 nrow, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const i32 *__restrict__ newcol,
                           u64 *__restrict__ len) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        u64 c = 0;
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) c += newcol[idx[k]] >= 0 ? 1 : 0;
        len[r] = c;
    }
}

__global__ void k_rc_fill(i64 nrow, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const double *__restrict__ val,
                          const i32 *__restrict__ newcol, const i64 *__restrict__ optr, i32 *__restrict__ oidx, double *__restrict__ oval) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        i64 o = optr[r];
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) {
            const i32 j = newcol[idx[k]];
            if (j >= 0) {
                oidx[o] = j;
                oval[o] = val[k];
                ++o;
            }
        }
    }
}

static slp_matrix *matrix_remove_columns(slp_matrix *a, const unsigned char *keep, const double *shift, double *a_shift) {
    SLP_REQUIRE(a && keep, "slp_matrix_remove_columns: NULL argument");
    SLP_REQUIRE((shift == nullptr) == (a_shift == nullptr), "slp_matrix_remove_columns: shift and a_shift go together");
    require_csr(a, "slp_matrix_remove_columns");
    Phase ph("slp_matrix_remove_columns");
    hipStream_t st = ctx().stream;
    const CsrDev &r = a->a;
    const i64 n = r.ncol, rows = r.nrow;
    std::vector<i32> newcol((size_t)n);
    i32 kept = 0;
    for (i64 j = 0; j < n; ++j) newcol[(size_t)j] = keep[j] ? kept++ : -1;
    auto *m = new slp_matrix();
    try {
        if (shift) {  // A * shift, csr_matvec order (SparseLP.py:646-650), on the unreduced matrix
            DevBuf<double> dx((size_t)n), dy((size_t)rows);
            dx.upload(shift, (size_t)n);
            matrix_spmv(a, false, dx.p, dy.p, SLP_ORDER_SEQUENTIAL);
            dy.download(a_shift, (size_t)rows);
        }
        DevBuf<i32> dcol((size_t)n);
        dcol.upload(newcol.data(), (size_t)n);
        DevBuf<u64> len((size_t)rows + 1);
        len.zero();
        if (rows) hipLaunchKernelGGL(k_rc_count, dim3(grid_for(rows, kBlock)), dim3(kBlock), 0, st, rows, r.ptr.p, r.idx.p, dcol.p, len.p);
        SLP_HIP(hipGetLastError());
        m->a.nrow = rows;
        m->a.ncol = kept;
        m->a.ptr.alloc((size_t)rows + 1);
        exclusive_scan(len.p, reinterpret_cast<u64 *>(m->a.ptr.p), (size_t)rows + 1);
        i64 nnz = 0;
        SLP_HIP(hipMemcpyAsync(&nnz, m->a.ptr.p + rows, sizeof(i64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        m->a.nnz = nnz;
        m->a.idx.alloc((size_t)nnz);
        m->a.val.alloc((size_t)nnz);
        if (rows) hipLaunchKernelGGL(k_rc_fill, dim3(grid_for(rows, kBlock)), dim3(kBlock), 0, st, rows, r.ptr.p, r.idx.p, r.val.p, dcol.p,
                                     m->a.ptr.p, m->a.idx.p, m->a.val.p);
        SLP_HIP(hipGetLastError());
        SLP_HIP(hipStreamSynchronize(st));
        finish_stats(m->a);
    } catch (...) {
        delete m;
        throw;
    }
    return m;
}

// ---- row normalisation: precondition_constraints (tools.py:272-290), alpha = 2 ----------------------------------
// s_i = sqrt(sum_k |a_ik|^2) accumulated in storage order (scipy csr_matvec of |A|^2 against ones), 0 -> 1; the scaled
// matrix is scipy's `diags(1/s) * A`: entry (1/s_i) a_ik, each ROW STORED IN REVERSED ENTRY ORDER (csr_matmat walks a
// linked list of the touched columns, last touched first) and entries that underflow to exactly 0 dropped.
__global__ void k_pre_norm(i64 nrow, const i64 *__restrict__ ptr, const double *__restrict__ val, double *__restrict__ inv,
                           u64 *__restrict__ len) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        double sum = 0.0;
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) {
            const double a = fabs(val[k]);
            sum += a * a;
        }
        double nrm = sqrt(sum);
        if (nrm == 0.0) nrm = 1.0;
        const double iv = 1.0 / nrm;
        inv[r] = iv;
        u64 c = 0;
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) c += (iv * val[k] != 0.0) ? 1 : 0;
        len[r] = c;
    }
}

__global__ void k_pre_fill(i64 nrow, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const double *__restrict__ val,
                           const double *__restrict__ inv, const i64 *__restrict__ optr, i32 *__restrict__ oidx, double *__restrict__ oval) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        const double iv = inv[r];
        i64 o = optr[r];
        for (i64 k = ptr[r + 1] - 1; k >= ptr[r]; --k) {
            const double v = iv * val[k];
            if (v != 0.0) {
                oidx[o] = idx[k];
                oval[o] = v;
                ++o;
            }
        }
    }
}

__global__ void k_scale_vec(i64 n, const double *__restrict__ inv, double *__restrict__ b) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) b[i] = inv[i] * b[i];  // inf stays inf
}

slp_matrix *matrix_precondition_rows(slp_matrix *a, double *b, double *b2) {
    SLP_REQUIRE(a, "precondition_rows: NULL matrix");
    require_csr(a, "precondition_rows");
    hipStream_t st = ctx().stream;
    const CsrDev &r = a->a;
    const i64 rows = r.nrow;
    auto *m = new slp_matrix();
    try {
        DevBuf<double> inv((size_t)rows);
        DevBuf<u64> len((size_t)rows + 1);
        len.zero();
        const int g = grid_for(rows, kBlock);
        if (rows) hipLaunchKernelGGL(k_pre_norm, dim3(g), dim3(kBlock), 0, st, rows, r.ptr.p, r.val.p, inv.p, len.p);
        SLP_HIP(hipGetLastError());
        m->a.nrow = rows;
        m->a.ncol = r.ncol;
        m->a.ptr.alloc((size_t)rows + 1);
        exclusive_scan(len.p, reinterpret_cast<u64 *>(m->a.ptr.p), (size_t)rows + 1);
        i64 nnz = 0;
        SLP_HIP(hipMemcpyAsync(&nnz, m->a.ptr.p + rows, sizeof(i64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        m->a.nnz = nnz;
        m->a.idx.alloc((size_t)nnz);
        m->a.val.alloc((size_t)nnz);
        if (rows) {
            hipLaunchKernelGGL(k_pre_fill, dim3(g), dim3(kBlock), 0, st, rows, r.ptr.p, r.idx.p, r.val.p, inv.p, m->a.ptr.p, m->a.idx.p,
                               m->a.val.p);
            if (b) hipLaunchKernelGGL(k_scale_vec, dim3(g), dim3(kBlock), 0, st, rows, inv.p, b);
            if (b2) hipLaunchKernelGGL(k_scale_vec, dim3(g), dim3(kBlock), 0, st, rows, inv.p, b2);
        }
        SLP_HIP(hipGetLastError());
        SLP_HIP(hipStreamSynchronize(st));
        finish_stats(m->a);
    } catch (...) {
        delete m;
        throw;
    }
    return m;
}

// ---- slack standard form: convert_to_standard_form_with_bounds (tools.py:88-127) ---------------------------------
// A = [[A_eq, 0], [A_ineq, -I]] as scipy's vstack / hstack / tocsr leave it: rows sorted by column, duplicate (row, column)
// pairs summed in their original order.
__global__ void k_sf_count(i64 me, i64 mi, const i64 *__restrict__ eptr, const i64 *__restrict__ iptr, u64 *__restrict__ len) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < me + mi; r += (i64)gridDim.x * blockDim.x)
        len[r] = r < me ? (u64)(eptr[r + 1] - eptr[r]) : (u64)(iptr[r - me + 1] - iptr[r - me]) + 1;
}

__global__ void k_sf_fill(i64 me, i64 mi, i64 n, const i64 *__restrict__ eptr, const i32 *__restrict__ eidx, const double *__restrict__ eval,
                          const i64 *__restrict__ iptr, const i32 *__restrict__ iidx, const double *__restrict__ ival,
                          const i64 *__restrict__ optr, u64 *__restrict__ okey, double *__restrict__ oval) {
    // key = (column << 32 | position in the row): a total order, so duplicates of a (row, column) pair keep their storage
    // order whatever algorithm the segmented sort picks for short segments (its small-segment path is not a stable sort)
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < me + mi; r += (i64)gridDim.x * blockDim.x) {
        const i64 o0 = optr[r];
        i64 o = o0;
        if (r < me) {
            for (i64 k = eptr[r]; k < eptr[r + 1]; ++k, ++o) { okey[o] = ((u64)(unsigned int)eidx[k] << 32) | (u64)(o - o0); oval[o] = eval[k]; }
        } else {
            const i64 i = r - me;
            for (i64 k = iptr[i]; k < iptr[i + 1]; ++k, ++o) { okey[o] = ((u64)(unsigned int)iidx[k] << 32) | (u64)(o - o0); oval[o] = ival[k]; }
            okey[o] = ((u64)(n + i) << 32) | (u64)(o - o0);
            oval[o] = -1.0;
        }
    }
}

// unique columns per (sorted) row
__global__ void k_sf_unique(i64 rows, const i64 *__restrict__ ptr, const u64 *__restrict__ key, u64 *__restrict__ len) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (i64)gridDim.x * blockDim.x) {
        u64 c = 0;
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) c += (k == ptr[r] || (key[k] >> 32) != (key[k - 1] >> 32)) ? 1 : 0;
        len[r] = c;
    }
}

__global__ void k_sf_merge(i64 rows, const i64 *__restrict__ ptr, const u64 *__restrict__ key, const double *__restrict__ val,
                           const i64 *__restrict__ optr, i32 *__restrict__ oidx, double *__restrict__ oval) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (i64)gridDim.x * blockDim.x) {
        i64 o = optr[r] - 1;
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) {
            if (k == ptr[r] || (key[k] >> 32) != (key[k - 1] >> 32)) {
                ++o;
                oidx[o] = (i32)(key[k] >> 32);
                oval[o] = val[k];
            } else {
                oval[o] = oval[o] + val[k];  // duplicates summed left to right, like COO -> CSR
            }
        }
    }
}

slp_matrix *matrix_standard_form(slp_matrix *ae, slp_matrix *ai) {
    SLP_REQUIRE(ai, "standard_form: the inequality block is required (tools.py:92)");
    require_csr(ai, "standard_form");
    if (ae) require_csr(ae, "standard_form");
    hipStream_t st = ctx().stream;
    const i64 me = ae ? ae->a.nrow : 0, mi = ai->a.nrow, n = ai->a.ncol, rows = me + mi;
    SLP_REQUIRE(!ae || ae->a.ncol == n, "standard_form: A_eq and A_ineq differ in their column count");
    SLP_REQUIRE(n + mi < ((i64)1 << 31), "standard_form: column count must fit int32");
    auto *m = new slp_matrix();
    try {
        const int g = grid_for(rows, kBlock);
        DevBuf<u64> len((size_t)rows + 1);
        DevBuf<i64> tptr((size_t)rows + 1);
        len.zero();
        if (rows) hipLaunchKernelGGL(k_sf_count, dim3(g), dim3(kBlock), 0, st, me, mi, ae ? ae->a.ptr.p : nullptr, ai->a.ptr.p, len.p);
        SLP_HIP(hipGetLastError());
        exclusive_scan(len.p, reinterpret_cast<u64 *>(tptr.p), (size_t)rows + 1);
        i64 total = 0;
        SLP_HIP(hipMemcpyAsync(&total, tptr.p + rows, sizeof(i64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        SLP_REQUIRE(total < (i64)0x7fffffffll, "standard_form: more than 2^31-1 stored entries");
        DevBuf<u64> key((size_t)total), key2((size_t)total);
        DevBuf<double> val((size_t)total), val2((size_t)total);
        if (rows)
            hipLaunchKernelGGL(k_sf_fill, dim3(g), dim3(kBlock), 0, st, me, mi, n, ae ? ae->a.ptr.p : nullptr, ae ? ae->a.idx.p : nullptr,
                               ae ? ae->a.val.p : nullptr, ai->a.ptr.p, ai->a.idx.p, ai->a.val.p, tptr.p, key.p, val.p);
        SLP_HIP(hipGetLastError());
        unsigned int bits = 1;
        while (bits < 32 && ((i64)1 << bits) < n + mi) ++bits;
        if (total) {  // every row sorted by (column, position in the row)
            size_t bytes = 0;
            SLP_HIP(rocprim::segmented_radix_sort_pairs(nullptr, bytes, key.p, key2.p, val.p, val2.p, (unsigned int)total, (unsigned int)rows,
                                                        tptr.p, tptr.p + 1, 0u, 32u + bits, st));
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::segmented_radix_sort_pairs(tmp.p, bytes, key.p, key2.p, val.p, val2.p, (unsigned int)total, (unsigned int)rows,
                                                        tptr.p, tptr.p + 1, 0u, 32u + bits, st));
            SLP_HIP(hipStreamSynchronize(st));
        }
        len.zero();
        if (rows) hipLaunchKernelGGL(k_sf_unique, dim3(g), dim3(kBlock), 0, st, rows, tptr.p, key2.p, len.p);
        SLP_HIP(hipGetLastError());
        m->a.nrow = rows;
        m->a.ncol = n + mi;
        m->a.ptr.alloc((size_t)rows + 1);
        exclusive_scan(len.p, reinterpret_cast<u64 *>(m->a.ptr.p), (size_t)rows + 1);
        i64 nnz = 0;
        SLP_HIP(hipMemcpyAsync(&nnz, m->a.ptr.p + rows, sizeof(i64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        m->a.nnz = nnz;
        m->a.idx.alloc((size_t)nnz);
        m->a.val.alloc((size_t)nnz);
        if (rows) hipLaunchKernelGGL(k_sf_merge, dim3(g), dim3(kBlock), 0, st, rows, tptr.p, key2.p, val2.p, m->a.ptr.p, m->a.idx.p, m->a.val.p);
        SLP_HIP(hipGetLastError());
        SLP_HIP(hipStreamSynchronize(st));
        finish_stats(m->a);
    } catch (...) {
        delete m;
        throw;
    }
    return m;
}

}  // namespace slp

using namespace slp;

extern "C" {

slp_matrix *slp_matrix_precondition_rows(slp_matrix *a, double *b, double *b2) {
    SLP_API_PTR({
        SLP_REQUIRE(a, "slp_matrix_precondition_rows: NULL matrix");
        const size_t rows = (size_t)a->a.nrow;
        DevBuf<double> db, db2;
        if (b) db.upload(b, rows);
        if (b2) db2.upload(b2, rows);
        slp_matrix *m = matrix_precondition_rows(a, b ? db.p : nullptr, b2 ? db2.p : nullptr);
        if (b) db.download(b, rows);
        if (b2) db2.download(b2, rows);
        return m;
    })
}

slp_matrix *slp_matrix_standard_form(slp_matrix *a_eq, slp_matrix *a_ineq) {
    SLP_API_PTR({ return matrix_standard_form(a_eq, a_ineq); })
}

slp_matrix *slp_matrix_normal(slp_matrix *a, double gamma_eq, double gamma_ineq) {
    SLP_API_PTR({ return matrix_normal(a, gamma_eq, gamma_ineq); })
}

slp_matrix *slp_matrix_remove_columns(slp_matrix *a, const unsigned char *keep, const double *shift, double *a_shift) {
    SLP_API_PTR({ return matrix_remove_columns(a, keep, shift, a_shift); })
}

}  // extern "C"
