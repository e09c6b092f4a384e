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

typedef int64_t i64;
typedef int32_t i32;

/* y = A x, A in CSR.  scipy csr_matvec: for each row, sum = y[i] (zero on
 * entry for `A * x`), then sum += data[k] * x[indices[k]] in storage order.
 * Call sites: ChambollePockPPD.py:235,240,267-272 ; ADMM.py:220,262 ;
 * SparseLP.py:193-202 ; tools.py:276,284. */
void orc_csr_matvec(i64 nrow, const i64 *indptr, const i32 *indices,
                    const double *data, const double *x, double *y)
{
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
 * rows visited in order, so every column lists its rows increasingly). */
void orc_csr_to_csc(i64 nrow, i64 ncol, const i64 *indptr, const i32 *indices,
                    const double *data, i64 *cptr, i32 *crow, double *cdata)
{
    const i64 nnz = indptr[nrow];
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
