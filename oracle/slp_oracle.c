/*
 * slp_oracle.c -- CPU restatement of the numeric kernels under PySparseLP's
 * first-order solvers.  TEST INFRASTRUCTURE ONLY: nothing in pysparselp_amd/
 * may link, import or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do, and only as the checker / the timed CPU
 * baseline, never as the product path.
 *
 * Parity: pinned.  Every function here is checked bit-for-bit against the
 * reference imported in the build container (tests/golden/make_golden.py) and
 * against the reference's own golden curves (tests/golden/ref_*.json).
 *
 * The arithmetic restated here lives partly in the reference tree
 * (pysparselp/gaussSiedel.pyx) and partly in a third-party dependency that is
 * NOT vendored under /root/reference: scipy.sparse._sparsetools, pinned
 * scipy==1.4.1 in the reference's requirements.txt:12 (the build container
 * has scipy 1.15.3, which reproduces the reference goldens bit-exactly).
 * Its published algorithms (csr_matvec, csc_matvec, csr_matmat = SMMP) are
 * restated from their documented behaviour; the call sites that fix the
 * semantics are cited per function.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: one rounding per
 * multiply and per add, like the x86-64 builds of scipy and of the Cython
 * module, which carry no FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef int64_t i64;
typedef int32_t i32;

/* Threads for the loops whose iterations are independent of each other (rows of a CSR product, columns of a CSC
 * product, rows of a setup transform).  1 by default -- the reference is single-threaded end to end and the timed
 * cpu_baseline keeps 1.  More threads never change a result: no sum is split or reordered, a thread owns whole rows
 * (or whole columns) and walks them in the order stated per function.  Used by the full-size parity tests
 * (tests/test_gpu_c3_full.py), where one thread would take minutes per iteration. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

/* y = A x, A in CSR.  scipy csr_matvec: for each row, sum = y[i] (zero on
 * entry for `A * x`), then sum += data[k] * x[indices[k]] in storage order.
 * Call sites: ChambollePockPPD.py:235,240,267-272 ; ADMM.py:220,262 ;
 * SparseLP.py:193-202 ; tools.py:276,284. */
void orc_csr_matvec(i64 nrow, const i64 *indptr, const i32 *indices,
                    const double *data, const double *x, double *y)
{
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(dynamic, 256)
    for (i64 i = 0; i < nrow; ++i) {
        double sum = 0.0;
        for (i64 k = indptr[i]; k < indptr[i + 1]; ++k)
            sum += data[k] * x[indices[k]];
        y[i] = sum;
    }
}

/* out = A^T y, computed the way `y * A` is for a CSR A: scipy forms A.T (a
 * CSC view over the same three arrays) and runs csc_matvec, i.e. it walks the
 * ROWS of A in order and scatters out[indices[k]] += data[k] * y[i].  Every
 * out[j] therefore accumulates its terms in increasing row order starting
 * from 0.  Call sites: ChambollePockPPD.py:206,216 (y_eq * a_eq),
 * ChambollePockPPD.py:134,144 (ones * |A|), ADMM.py:95,148 (A.T * b,
 * lambda_eq * a_eq). */
void orc_csr_rmatvec(i64 nrow, i64 ncol, const i64 *indptr, const i32 *indices,
                     const double *data, const double *y, double *out)
{
    for (i64 j = 0; j < ncol; ++j)
        out[j] = 0.0;
    for (i64 i = 0; i < nrow; ++i) {
        const double yi = y[i];
        for (i64 k = indptr[i]; k < indptr[i + 1]; ++k)
            out[indices[k]] += data[k] * yi;
    }
}

/* out += y * A, CONTINUING the chains of additions `out` already holds: the form a row-chunked matrix needs -- chunk k of the
 * stacked rows carries on the column sums chunks 0 .. k-1 left, so that out[j] = ((0 + t_1) + t_2) + ... over ALL rows in
 * order, exactly what orc_csr_rmatvec computes on the stacked matrix (tests/test_oracle_golden.py).  Threads own column
 * ranges (a column's chain is walked by one thread, rows in order): the rows must be sorted by column, the range of a
 * thread inside a row is found by binary search.  Returns 0, or 1 + the first unsorted row. */
i64 orc_csr_rmatvec_acc(i64 nrow, i64 ncol, const i64 *indptr, const i32 *indices, const double *data, const double *y,
                        double *out)
{
    for (i64 i = 0; i < nrow; ++i)
        for (i64 k = indptr[i] + 1; k < indptr[i + 1]; ++k)
            if (indices[k - 1] >= indices[k])
                return 1 + i;
    const int nt = g_threads < 1 ? 1 : g_threads;
#pragma omp parallel num_threads(nt) if (nt > 1)
    {
        const int t = omp_get_thread_num(), tt = omp_get_num_threads();
        const i64 c0 = ncol * t / tt, c1 = ncol * (t + 1) / tt;
        for (i64 i = 0; i < nrow; ++i) {
            i64 lo = indptr[i], hi = indptr[i + 1];
            while (lo < hi) {                       /* first entry of the row with column >= c0 */
                const i64 mid = (lo + hi) >> 1;
                if ((i64)indices[mid] < c0) lo = mid + 1;
                else hi = mid;
            }
            const double yi = y[i];
            for (i64 k = lo; k < indptr[i + 1] && (i64)indices[k] < c1; ++k)
                out[indices[k]] += data[k] * yi;
        }
    }
    return 0;
}

/* The same `y * A` from the CSC arrays of A as orc_csr_to_csc builds them (stable: rows increasing inside every
 * column, ties in storage order): out[j] = ((0 + t_1) + t_2) + ... over the column's entries -- exactly the chain
 * of additions orc_csr_rmatvec performs on out[j], so the two agree bit for bit (tests/test_oracle_golden.py).
 * Columns are independent: this is the form the multi-threaded runs use. */
void orc_csc_rmatvec(i64 ncol, const i64 *cptr, const i32 *crow, const double *cdata, const double *y, double *out)
{
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(dynamic, 256)
    for (i64 j = 0; j < ncol; ++j) {
        double sum = 0.0;
        for (i64 p = cptr[j]; p < cptr[j + 1]; ++p)
            sum += cdata[p] * y[crow[p]];
        out[j] = sum;
    }
}

/* Projected (box-clamped) SOR sweep, in place on x.
 * gaussSiedel.pyx:131-152: natural order i = 0..N-1 (the `order` argument is
 * ignored, :132-134), v = sum_k x[indices[k]] * data[k] in storage order
 * including the diagonal, then v = w*(b[i]-v)*invD[i] + x[i], clamp to
 * [lo[i], hi[i]] with `if v<l: v=l elif v>u: v=u`, store. */
void orc_bounded_gauss_seidel(i64 n, const i64 *indptr, const i32 *indices,
                              const double *data, const double *invD,
                              const double *b, const double *lo,
                              const double *hi, double *x, int maxiter,
                              double w)
{
    for (int it = 0; it < maxiter; ++it) {
        for (i64 i = 0; i < n; ++i) {
            double v = 0.0;
            for (i64 k = indptr[i]; k < indptr[i + 1]; ++k)
                v += x[indices[k]] * data[k];
            v = w * (b[i] - v) * invD[i] + x[i];
            const double l = lo[i], u = hi[i];
            if (v < l)
                v = l;
            else if (v > u)
                v = u;
            x[i] = v;
        }
    }
}

/* Stable CSR -> CSC conversion (scipy csr_tocsc: counting sort by column,
 * rows visited in order, so every column lists its rows increasingly).
 * With several threads: thread t owns a contiguous range of rows, counts its entries per column, and is handed, for
 * every column, the slots right behind those of the threads before it -- the same placement as the one-thread loop. */
void orc_csr_to_csc(i64 nrow, i64 ncol, const i64 *indptr, const i32 *indices,
                    const double *data, i64 *cptr, i32 *crow, double *cdata)
{
    const i64 nnz = indptr[nrow];
    const int T = (g_threads > 1 && nnz > 100000) ? g_threads : 1;
    if (T == 1) {
        memset(cptr, 0, (size_t)(ncol + 1) * sizeof(i64));
        for (i64 k = 0; k < nnz; ++k)
            cptr[indices[k] + 1]++;
        for (i64 j = 0; j < ncol; ++j)
            cptr[j + 1] += cptr[j];
        i64 *next = (i64 *)malloc((size_t)(ncol + 1) * sizeof(i64));
        memcpy(next, cptr, (size_t)(ncol + 1) * sizeof(i64));
        for (i64 i = 0; i < nrow; ++i)
            for (i64 k = indptr[i]; k < indptr[i + 1]; ++k) {
                const i64 p = next[indices[k]]++;
                crow[p] = (i32)i;
                cdata[p] = data[k];
            }
        free(next);
        return;
    }
    i64 *hist = (i64 *)calloc((size_t)T * (size_t)ncol, sizeof(i64));
    i64 *row0 = (i64 *)malloc((size_t)(T + 1) * sizeof(i64));
    for (int t = 0; t <= T; ++t) { /* row ranges of about equal entry counts */
        const i64 want = nnz / T * t;
        i64 lo = 0, hi = nrow;
        while (lo < hi) {
            const i64 mid = (lo + hi) / 2;
            if (indptr[mid] < want) lo = mid + 1; else hi = mid;
        }
        row0[t] = (t == T) ? nrow : lo;
    }
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        i64 *h = hist + (size_t)t * (size_t)ncol;
        for (i64 k = indptr[row0[t]]; k < indptr[row0[t + 1]]; ++k)
            h[indices[k]]++;
    }
    cptr[0] = 0;
#pragma omp parallel for num_threads(T) schedule(static)
    for (i64 j = 0; j < ncol; ++j) {
        i64 tot = 0;
        for (int t = 0; t < T; ++t) tot += hist[(size_t)t * (size_t)ncol + j];
        cptr[j + 1] = tot;
    }
    for (i64 j = 0; j < ncol; ++j)
        cptr[j + 1] += cptr[j];
#pragma omp parallel for num_threads(T) schedule(static)
    for (i64 j = 0; j < ncol; ++j) {
        i64 at = cptr[j];
        for (int t = 0; t < T; ++t) {
            const i64 c = hist[(size_t)t * (size_t)ncol + j];
            hist[(size_t)t * (size_t)ncol + j] = at;
            at += c;
        }
    }
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        i64 *next = hist + (size_t)t * (size_t)ncol;
        for (i64 i = row0[t]; i < row0[t + 1]; ++i)
            for (i64 k = indptr[i]; k < indptr[i + 1]; ++k) {
                const i64 p = next[indices[k]]++;
                crow[p] = (i32)i;
                cdata[p] = data[k];
            }
    }
    free(hist);
    free(row0);
}

/* M = gamma_eq * A^T A + gamma_ineq * I as CSR with sorted column indices
 * (ADMM.py:93,96,100-101).  scipy evaluates A.T * A with SMMP (csr_matmat on
 * the CSC operands): entry (i,j) accumulates A[k,j]*A[k,i] over the shared
 * rows k in increasing k, one rounding per product and per add, and entries
 * whose sum is exactly 0 are dropped; `gamma_eq * a_t_a` scales each stored
 * value, the sparse `+` adds gamma_ineq on the diagonal (result entries equal
 * to 0 are dropped by the binop), `.tocsr()` leaves sorted rows.
 *
 * Two calls: with Mj == NULL it only counts (returns nnz(M)); otherwise it
 * fills Mp/Mj/Mx.  cptr/crow/cdata = CSC of A from orc_csr_to_csc. */
i64 orc_normal_matrix(i64 nrow, i64 ncol, const i64 *indptr, const i32 *indices,
                      const double *data, const i64 *cptr, const i32 *crow,
                      const double *cdata, double gamma_eq, double gamma_ineq,
                      i64 *Mp, i32 *Mj, double *Mx)
{
    (void)nrow;
    double *acc = (double *)calloc((size_t)ncol, sizeof(double));
    unsigned char *seen = (unsigned char *)calloc((size_t)ncol, 1);
    i32 *cols = (i32 *)malloc((size_t)ncol * sizeof(i32));
    i64 nnz = 0;
    if (Mp)
        Mp[0] = 0;
    for (i64 i = 0; i < ncol; ++i) {
        i64 cnt = 0;
        for (i64 p = cptr[i]; p < cptr[i + 1]; ++p) {
            const i64 k = crow[p];
            const double aki = cdata[p];
            for (i64 q = indptr[k]; q < indptr[k + 1]; ++q) {
                const i32 j = indices[q];
                if (!seen[j]) {
                    seen[j] = 1;
                    cols[cnt++] = j;
                }
                acc[j] += data[q] * aki;
            }
        }
        if (!seen[i]) { /* identity contributes the diagonal even if A has an empty column */
            seen[i] = 1;
            cols[cnt++] = (i32)i;
        }
        /* insertion sort of the touched columns (rows of M are short) */
        for (i64 a = 1; a < cnt; ++a) {
            const i32 v = cols[a];
            i64 b = a - 1;
            while (b >= 0 && cols[b] > v) {
                cols[b + 1] = cols[b];
                --b;
            }
            cols[b + 1] = v;
        }
        for (i64 a = 0; a < cnt; ++a) {
            const i32 j = cols[a];
            double v;
            const double ata = acc[j];
            if (j == i)
                v = (ata != 0.0) ? (gamma_eq * ata + gamma_ineq) : gamma_ineq;
            else
                v = (ata != 0.0) ? gamma_eq * ata : 0.0;
            acc[j] = 0.0;
            seen[j] = 0;
            if (v != 0.0) {
                if (Mj) {
                    Mj[nnz] = j;
                    Mx[nnz] = v;
                }
                ++nnz;
            }
        }
        if (Mp)
            Mp[i + 1] = nnz;
    }
    free(acc);
    free(seen);
    free(cols);
    return nnz;
}

/* Row p-norm scaling factors of tools.py:272-281 (precondition_constraints,
 * alpha=2): s_i = (sum_k |a_ik|^alpha)^(1/alpha) with the row sum taken by
 * csr_matvec against a vector of ones (so each term is |a|^alpha * 1.0),
 * s_i == 0 -> 1, returns 1/s_i.  numpy evaluates |a|**2 as a*a and s**(0.5)
 * as sqrt(s) (both exact equivalents for alpha == 2, the only value the hot
 * path uses: ADMM.py:77,82,91). */
void orc_row_scale_l2(i64 nrow, const i64 *indptr, const double *data, double *inv_s)
{
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(dynamic, 256)
    for (i64 i = 0; i < nrow; ++i) {
        double sum = 0.0;
        for (i64 k = indptr[i]; k < indptr[i + 1]; ++k) {
            const double a = fabs(data[k]);
            sum += (a * a) * 1.0;
        }
        double s = sqrt(sum);
        if (s == 0.0)
            s = 1.0;
        inv_s[i] = 1.0 / s;
    }
}

/* Unbounded SOR sweep, gaussSiedel.pyx:21-79 (`GaussSeidel`), natural order (order=None -> arange):
 * v = sum_k x[indices[k]] * data[k] ; nv = (b[i] - v + D[i]*x[i]) * invD[i] ; x[i] = w*nv + (1-w)*x[i].
 * Only reached through the reference's flag-selected branch ADMM.py:164-181. */
void orc_gauss_seidel(i64 n, const i64 *indptr, const i32 *indices, const double *data, const double *D,
                      const double *invD, const double *b, double *x, int maxiter, double w)
{
    for (int it = 0; it < maxiter; ++it) {
        for (i64 i = 0; i < n; ++i) {
            double v = 0.0;
            for (i64 k = indptr[i]; k < indptr[i + 1]; ++k)
                v += x[indices[k]] * data[k];
            double nv = (b[i] - v + D[i] * x[i]) * invD[i];
            nv = w * nv + (1 - w) * x[i];
            x[i] = nv;
        }
    }
}

/* `diags(inv_s) * A` of tools.py:283-284 as scipy evaluates it: a sparse-sparse product (SMMP, csr_matmat) whose
 * output lists every row in the REVERSE of the order in which its columns were first touched -- with one term per
 * entry that is the reverse of the input row -- with the values inv_s[i] * a_ik (one rounding) and entries whose value
 * is exactly 0 dropped.  Two calls: counts[i] = entries row i keeps (out_indices == NULL), then, with out_indptr the
 * running sum of the counts, the fill. */
void orc_scale_rows_reversed(i64 nrow, const i64 *indptr, const i32 *indices, const double *data, const double *inv_s,
                             i64 *counts, const i64 *out_indptr, i32 *out_indices, double *out_data)
{
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(dynamic, 256)
    for (i64 i = 0; i < nrow; ++i) {
        const double f = inv_s[i];
        if (!out_indices) {
            i64 c = 0;
            for (i64 k = indptr[i]; k < indptr[i + 1]; ++k)
                if (f * data[k] != 0.0) ++c;
            counts[i] = c;
        } else {
            i64 o = out_indptr[i];
            for (i64 k = indptr[i + 1] - 1; k >= indptr[i]; --k) {
                const double v = f * data[k];
                if (v != 0.0) {
                    out_indices[o] = indices[k];
                    out_data[o] = v;
                    ++o;
                }
            }
        }
    }
}

/* Rows of [[A_eq, 0], [A_ineq, -I]] before they are sorted (tools.py:112-118, scipy hstack / vstack through COO):
 * an equality row keeps its entries, inequality row i keeps its entries followed by (n + i, -1.0).
 * out_indptr is given (row lengths are known to the caller). */
void orc_stack_standard_form(i64 me, i64 ni, i64 n, const i64 *ep, const i32 *ej, const double *ex, const i64 *ip,
                             const i32 *ij, const double *ix, const i64 *out_indptr, i32 *out_indices, double *out_data)
{
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(dynamic, 256)
    for (i64 r = 0; r < me + ni; ++r) {
        i64 o = out_indptr[r];
        if (r < me) {
            for (i64 k = ep[r]; k < ep[r + 1]; ++k, ++o) { out_indices[o] = ej[k]; out_data[o] = ex[k]; }
        } else {
            const i64 i = r - me;
            for (i64 k = ip[i]; k < ip[i + 1]; ++k, ++o) { out_indices[o] = ij[k]; out_data[o] = ix[k]; }
            out_indices[o] = (i32)(n + i);
            out_data[o] = -1.0;
        }
    }
}

/* Every row stably sorted by column, in place (what scipy's COO -> CSR conversion + sort_indices leave: the stable
 * order np.lexsort((indices, rows)) gives).  Returns the number of adjacent equal-column pairs left (duplicates; the
 * caller sums them left to right).  Rows already increasing are left alone, strictly decreasing rows are reversed,
 * everything else goes through a stable merge sort. */
static void merge_sort_row(i32 *j, double *v, i32 *tj, double *tv, i64 n)
{
    for (i64 w = 1; w < n; w *= 2) {
        for (i64 lo = 0; lo < n; lo += 2 * w) {
            const i64 mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            i64 a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                if (j[b] < j[a]) { tj[o] = j[b]; tv[o] = v[b]; ++b; }
                else { tj[o] = j[a]; tv[o] = v[a]; ++a; }
                ++o;
            }
            while (a < mid) { tj[o] = j[a]; tv[o] = v[a]; ++a; ++o; }
            while (b < hi) { tj[o] = j[b]; tv[o] = v[b]; ++b; ++o; }
        }
        memcpy(j, tj, (size_t)n * sizeof(i32));
        memcpy(v, tv, (size_t)n * sizeof(double));
    }
}

i64 orc_sort_rows(i64 nrow, const i64 *indptr, i32 *indices, double *data)
{
    i64 dups = 0;
#pragma omp parallel num_threads(g_threads) if (g_threads > 1) reduction(+ : dups)
    {
        i64 cap = 0;
        i32 *tj = NULL;
        double *tv = NULL;
#pragma omp for schedule(dynamic, 256)
        for (i64 i = 0; i < nrow; ++i) {
            i32 *j = indices + indptr[i];
            double *v = data + indptr[i];
            const i64 n = indptr[i + 1] - indptr[i];
            int inc = 1, dec = 1;
            for (i64 k = 1; k < n; ++k) {
                if (j[k] < j[k - 1]) inc = 0;
                if (j[k] >= j[k - 1]) dec = 0;
            }
            if (!inc && dec) {
                for (i64 a = 0, b = n - 1; a < b; ++a, --b) {
                    const i32 tji = j[a]; j[a] = j[b]; j[b] = tji;
                    const double tvi = v[a]; v[a] = v[b]; v[b] = tvi;
                }
            } else if (!inc) {
                if (n > cap) {
                    cap = 2 * n;
                    tj = (i32 *)realloc(tj, (size_t)cap * sizeof(i32));
                    tv = (double *)realloc(tv, (size_t)cap * sizeof(double));
                }
                merge_sort_row(j, v, tj, tv, n);
            }
            for (i64 k = 1; k < n; ++k)
                if (j[k] == j[k - 1]) ++dups;
        }
        free(tj);
        free(tv);
    }
    return dups;
}
