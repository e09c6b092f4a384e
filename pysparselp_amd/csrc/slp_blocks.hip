// slp_blocks.hip -- ADMM with one copy of the variables per block of constraints
// (reference ADMMBlocks.py:45-352, `lp_admm_block_decomposition`).
//
// The reference factorises, for every block g of rows, the KKT matrix [[gamma I, A_g^T], [A_g, 0]] with a
// sparse LU and solves it once per iteration (:178-243, :268-284).  That solve is the projection of
// v = xp[ids_g] - lambda_g / gamma onto {A_g x = b_g}:  x = v - A_g^T nu,  (A_g A_g^T) nu = A_g v - b_g.
// Here all blocks are handled at once and matrix-free: the host lays the copies side by side (P = sum of the
// blocks' variable counts) and builds the "split" matrix A^ (m x P, block diagonal in those coordinates), and
// one conjugate-gradient run on  S = A^ A^T  (block diagonal, SPD for full-rank blocks) solves every block's
// system together -- two SpMV passes per CG step through the library's SpMV kernels, warm-started from the
// previous iteration's nu.  The consensus average, the clamp and the multiplier update are elementwise.
// Parity with the LU form is a tolerance (the CG residual bound), not bit-exact; every reduction is two-stage
// with a fixed order, so runs are reproducible.
#include <algorithm>
#include <cmath>
#include <vector>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

constexpr int kBlkPartials = 1024;

// out[0] = sum a_i b_i  (two-stage, fixed order)
__global__ __launch_bounds__(kBlock) void k_blk_dot(i64 n, const double *__restrict__ a, const double *__restrict__ b,
                                                    double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double s = 0.0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) s += a[i] * b[i];
    const double r = block_reduce<false>(s, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// scal[slot] = sum of the partials; then the CG scalars that depend on it:
//   mode 1 (slot = PQ): alpha = rs / pq            (0 when the residual is already 0)
//   mode 2 (slot = RSNEW): beta = rsnew / rs ; rs = rsnew
enum { B_RS = 0, B_PQ = 1, B_RSNEW = 2, B_ALPHA = 3, B_BETA = 4, B_RHS2 = 5, B_RR = 6, B_COUNT = 8 };
__global__ __launch_bounds__(kBlock) void k_blk_finish(int nparts, const double *__restrict__ part, double *__restrict__ scal, int slot,
                                                       int mode) {
    __shared__ double lds[kBlock / kWave];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) s += part[i];
    const double r = block_reduce<false>(s, lds);
    if (threadIdx.x == 0) {
        scal[slot] = r;
        if (mode == 1) scal[B_ALPHA] = (scal[B_RS] > 0.0 && r > 0.0) ? scal[B_RS] / r : 0.0;
        if (mode == 2) {
            scal[B_BETA] = scal[B_RS] > 0.0 ? r / scal[B_RS] : 0.0;
            scal[B_RS] = r;
        }
    }
}

// v_p = xp[owner_p] - lambda_p / gamma     (:270-276 rewritten: y / gamma)
__global__ void k_blk_v(i64 P, const i32 *__restrict__ owner, const double *__restrict__ xp, const double *__restrict__ lam, double gamma,
                        double *__restrict__ v) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (i64)gridDim.x * blockDim.x) v[p] = xp[owner[p]] - lam[p] / gamma;
}

// r = w - b - q   (w = A^ v, q = S nu: residual of the warm start) ; dir = r
__global__ void k_blk_resid0(i64 m, const double *__restrict__ w, const double *__restrict__ b, const double *__restrict__ q,
                             double *__restrict__ r, double *__restrict__ dir, double *__restrict__ rhs) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        const double f = w[i] - b[i];
        rhs[i] = f;
        const double ri = f - q[i];
        r[i] = ri;
        dir[i] = ri;
    }
}

// nu += alpha dir ; r -= alpha q
__global__ void k_blk_step(i64 m, const double *__restrict__ scal, const double *__restrict__ dir, const double *__restrict__ q,
                           double *__restrict__ nu, double *__restrict__ r) {
    const double alpha = scal[B_ALPHA];
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        nu[i] = nu[i] + alpha * dir[i];
        r[i] = r[i] - alpha * q[i];
    }
}

// Jacobi preconditioner (slp_blocks_set_precond): z = r / diag(S)
__global__ void k_blk_precond(i64 m, const double *__restrict__ dinv, const double *__restrict__ r, double *__restrict__ z) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) z[i] = dinv[i] * r[i];
}

// 1 / diag(S): S = I + A^T A over the columns (rows of the transposed copy: 1 + sum a_ij^2), or A A^T (+ I on inequality
// rows) over the rows
__global__ void k_blk_diag(i64 rows, i64 first_identity, const i64 *__restrict__ ptr, const double *__restrict__ val, double base,
                           double *__restrict__ dinv) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (i64)gridDim.x * blockDim.x) {
        double s = (r >= first_identity) ? base : 0.0;
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) s += val[k] * val[k];
        dinv[r] = s > 0.0 ? 1.0 / s : 1.0;
    }
}

// dir = r + beta dir
__global__ void k_blk_dir(i64 m, const double *__restrict__ scal, const double *__restrict__ r, double *__restrict__ dir) {
    const double beta = scal[B_BETA];
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) dir[i] = r[i] + beta * dir[i];
}

// x_p = alpha (v_p - u_p) + (1 - alpha) xp[owner_p]     (u = A^T nu; :280-284)
__global__ void k_blk_x(i64 P, const i32 *__restrict__ owner, const double *__restrict__ xp, const double *__restrict__ v,
                        const double *__restrict__ u, double alpha, double one_minus_alpha, double *__restrict__ x) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (i64)gridDim.x * blockDim.x)
        x[p] = alpha * (v[p] - u[p]) + one_minus_alpha * xp[owner[p]];
}

// consensus (:290-299): xp_j = clamp((sum over the copies of j, in block order, of (x_p + lambda_p / gamma) - c_j / gamma)
//                                    / max(copies, 1));  a variable no block uses keeps its xp in the sum
__global__ void k_blk_consensus(i64 N, const i64 *__restrict__ cptr, const i32 *__restrict__ cidx, const double *__restrict__ x,
                                const double *__restrict__ lam, const double *__restrict__ c, const double *__restrict__ lb,
                                const double *__restrict__ ub, double gamma, double *__restrict__ xp) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (i64)gridDim.x * blockDim.x) {
        const i64 s = cptr[j], e = cptr[j + 1];
        double a = (e > s) ? 0.0 : xp[j];
        for (i64 k = s; k < e; ++k) {
            const i32 p = cidx[k];
            a += x[p] + lam[p] / gamma;
        }
        a = a - c[j] / gamma;
        const double cnt = (double)(e - s);
        a = a / (cnt > 1.0 ? cnt : 1.0);
        const double l = lb[j], u = ub[j];
        a = (a > l) ? a : l;  // np.maximum / np.minimum
        a = (a < u) ? a : u;
        xp[j] = a;
    }
}

// lambda_p += gamma (x_p - xp[owner_p])     (:302-307)
__global__ void k_blk_lambda(i64 P, const i32 *__restrict__ owner, const double *__restrict__ x, const double *__restrict__ xp, double gamma,
                             double *__restrict__ lam) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (i64)gridDim.x * blockDim.x)
        lam[p] = lam[p] + gamma * (x[p] - xp[owner[p]]);
}

// energy terms (:246-253): part[b] = sum_p 0.5 gamma d_p^2 + lambda_p d_p,  d = x - xp[owner]
__global__ __launch_bounds__(kBlock) void k_blk_energy(i64 P, const i32 *__restrict__ owner, const double *__restrict__ x,
                                                       const double *__restrict__ xp, const double *__restrict__ lam, double gamma,
                                                       double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double s = 0.0;
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (i64)gridDim.x * blockDim.x) {
        const double d = x[p] - xp[owner[p]];
        s += 0.5 * gamma * (d * d) + lam[p] * d;
    }
    const double r = block_reduce<false>(s, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// ---- one block per rank over a device-resident row block (at scale / multi-GPU) -------------------------
// The rank's rows  [A_eq; A_ineq] x {=, <=}  are ONE block of the standard form [A_eq 0; A_ineq -I]: its copies are the n
// original variables (those whose column is not empty in this row block) plus one slack per inequality row, which no
// other block shares.  The slack column stays implicit:  A^ v = A v_x - v_s,  A^^T nu = [A^T nu; -nu],
// S nu = A (A^T nu) + nu  on inequality rows.  The only exchange is the consensus sum over the ranks (n doubles).
bool comm_active();
void comm_allreduce_dev(double *buf, i64 count, int op);
void comm_allreduce_dev_async(double *buf, i64 count, int op);
void comm_join();
bool comm_library_collective_pending();
void comm_sync_side();

// v = xp - lambda / gamma  (original part)
__global__ void k_rb_v(i64 n, const double *__restrict__ xp, const double *__restrict__ lam, double gamma, double *__restrict__ v) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) v[j] = xp[j] - lam[j] / gamma;
}

// rows: vs = xps - lams / gamma ; rhs = (A v)_i - [ineq] vs_i - b_i ; r = rhs - (q_i + [ineq] nu_i) ; dir = r
__global__ void k_rb_resid0(i64 m, i64 m_eq, const double *__restrict__ w, const double *__restrict__ b, const double *__restrict__ q,
                            const double *__restrict__ nu, const double *__restrict__ xps, const double *__restrict__ lams, double gamma,
                            double *__restrict__ vs, double *__restrict__ r, double *__restrict__ dir, double *__restrict__ rhs) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        const bool ineq = i >= m_eq;
        const double s = ineq ? xps[i] - lams[i] / gamma : 0.0;
        vs[i] = s;
        const double f = ineq ? (w[i] - s) - b[i] : w[i] - b[i];
        rhs[i] = f;
        const double ri = f - (ineq ? q[i] + nu[i] : q[i]);
        r[i] = ri;
        dir[i] = ri;
    }
}

// q += dir on inequality rows (the slack column's share of S dir)
__global__ void k_rb_add_identity(i64 m, i64 m_eq, const double *__restrict__ dir, double *__restrict__ q) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x + m_eq; i < m; i += (i64)gridDim.x * blockDim.x) q[i] = q[i] + dir[i];
}

// original part: x = alpha (v - u) + (1 - alpha) xp ; acc = used ? x + lambda / gamma : 0   (the consensus summand)
__global__ void k_rb_x(i64 n, const unsigned char *__restrict__ used, const double *__restrict__ xp, const double *__restrict__ v,
                       const double *__restrict__ u, const double *__restrict__ lam, double alpha, double one_minus_alpha, double gamma,
                       double *__restrict__ x, double *__restrict__ acc) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const double xj = alpha * (v[j] - u[j]) + one_minus_alpha * xp[j];
        x[j] = xj;
        acc[j] = used[j] ? (0.0 + (xj + lam[j] / gamma)) : 0.0;
    }
}

// slack part, all local: xs = alpha (vs + nu) + (1 - alpha) xps ; xps = clamp(xs + lams / gamma) ; lams += gamma (xs - xps)
__global__ void k_rb_slack(i64 m, i64 m_eq, const double *__restrict__ vs, const double *__restrict__ nu, const double *__restrict__ slo,
                           const double *__restrict__ shi, double alpha, double one_minus_alpha, double gamma, double *__restrict__ xs,
                           double *__restrict__ xps, double *__restrict__ lams) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x + m_eq; i < m; i += (i64)gridDim.x * blockDim.x) {
        const double x = alpha * (vs[i] + nu[i]) + one_minus_alpha * xps[i];
        double a = (0.0 + (x + lams[i] / gamma)) - 0.0 / gamma;  // cost 0, one copy
        a = a / 1.0;
        const double l = slo[i], h = shi[i];
        a = (a > l) ? a : l;
        a = (a < h) ? a : h;
        xs[i] = x;
        xps[i] = a;
        lams[i] = lams[i] + gamma * (x - a);
    }
}

// xp = clamp((acc - c / gamma) / max(copies, 1))  (acc = sum over the ranks; a column no rank uses keeps xp in the sum) ;
// lambda += gamma (x - xp) where this rank holds a copy
__global__ void k_rb_consensus(i64 n, const unsigned char *__restrict__ used, const double *__restrict__ copies, const double *__restrict__ acc,
                               const double *__restrict__ c, const double *__restrict__ lb, const double *__restrict__ ub,
                               const double *__restrict__ x, double gamma, double *__restrict__ xp, double *__restrict__ lam) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const double cnt = copies[j];
        double a = cnt > 0.0 ? acc[j] : xp[j];
        a = a - c[j] / gamma;
        a = a / (cnt > 1.0 ? cnt : 1.0);
        const double l = lb[j], h = ub[j];
        a = (a > l) ? a : l;
        a = (a < h) ? a : h;
        xp[j] = a;
        if (used[j]) lam[j] = lam[j] + gamma * (x[j] - a);
    }
}

// ---- primal form of the projection, all rows inequalities and at least as many rows as columns:
// min |x - v|^2 + |A x - vs|^2  <=>  (I + A^T A) x = v + A^T vs ,  s = A x.  Same projection as the dual form
// (A A^T + I) nu = A v - vs, x = v - A^T nu, but the matrix has no unit eigenvalues from the rank deficit of A A^T:
// condition (1 + smax^2) / (1 + smin^2) instead of 1 + smax^2 -- about a third of the CG steps on the C3 LP.
__global__ void k_rb_vs(i64 m, const double *__restrict__ xps, const double *__restrict__ lams, double gamma, double *__restrict__ vs) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) vs[i] = xps[i] - lams[i] / gamma;
}

// rhs = v + u (u = A^T vs) ; r = rhs - (xsol + q) (q = A^T A xsol) ; dir = r
__global__ void k_rb_resid0_primal(i64 n, const double *__restrict__ v, const double *__restrict__ u, const double *__restrict__ xsol,
                                   const double *__restrict__ q, double *__restrict__ r, double *__restrict__ dir, double *__restrict__ rhs) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const double f = v[j] + u[j];
        rhs[j] = f;
        const double rj = f - (xsol[j] + q[j]);
        r[j] = rj;
        dir[j] = rj;
    }
}

__global__ void k_rb_true_resid(i64 n, const double *__restrict__ rhs, const double *__restrict__ q, double *__restrict__ r) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) r[i] = rhs[i] - q[i];
}

__global__ void k_rb_add_vec(i64 n, const double *__restrict__ a, double *__restrict__ q) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) q[j] = q[j] + a[j];
}

// x = alpha xv + (1 - alpha) xp ; acc = used ? x + lambda / gamma : 0
__global__ void k_rb_x_primal(i64 n, const unsigned char *__restrict__ used, const double *__restrict__ xp, const double *__restrict__ xv,
                              const double *__restrict__ lam, double alpha, double one_minus_alpha, double gamma, double *__restrict__ x,
                              double *__restrict__ acc) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const double xj = alpha * xv[j] + one_minus_alpha * xp[j];
        x[j] = xj;
        acc[j] = used[j] ? (0.0 + (xj + lam[j] / gamma)) : 0.0;
    }
}

__global__ void k_rb_used(i64 n, const i64 *__restrict__ tptr, unsigned char *__restrict__ used, double *__restrict__ copies) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const unsigned char u = tptr[j + 1] > tptr[j] ? 1 : 0;
        used[j] = u;
        copies[j] = (double)u;
    }
}

// the same from the column counts (|A|^0)^T 1 -- a matrix whose CSR entries are gone (chunked, released) only has its products
__global__ void k_rb_used_counts(i64 n, const double *__restrict__ cnt, unsigned char *__restrict__ used, double *__restrict__ copies) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const unsigned char u = cnt[j] > 0.0 ? 1 : 0;
        used[j] = u;
        copies[j] = (double)u;
    }
}

__global__ void k_rb_fill(i64 n, double v, double *__restrict__ out) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) out[j] = v;
}

// energy terms of this rank's block: part = sum_{used j} 0.5 g d^2 + lambda d  +  the same over the slacks
__global__ __launch_bounds__(kBlock) void k_rb_energy(i64 n, i64 m, i64 m_eq, const unsigned char *__restrict__ used,
                                                      const double *__restrict__ x, const double *__restrict__ xp,
                                                      const double *__restrict__ lam, const double *__restrict__ xs,
                                                      const double *__restrict__ xps, const double *__restrict__ lams, double gamma,
                                                      double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double s = 0.0;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x)
        if (used[j]) {
            const double d = x[j] - xp[j];
            s += 0.5 * gamma * (d * d) + lam[j] * d;
        }
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x + m_eq; i < m; i += (i64)gridDim.x * blockDim.x) {
        const double d = xs[i] - xps[i];
        s += 0.5 * gamma * (d * d) + lams[i] * d;
    }
    const double r = block_reduce<false>(s, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

}  // namespace slp

using namespace slp;

struct slp_blocks {
    slp_matrix *a = nullptr;  // the split matrix A^ (m x P), owned
    i64 P = 0, m = 0, N = 0;
    double gamma = 0.7, alpha = 1.95, tol = 1e-13;
    int max_cg = 500, check_every = 10;
    long long cg_steps = 0;   // CG steps taken so far (all iterations)
    DevBuf<i32> owner, cidx;
    DevBuf<i64> cptr;
    DevBuf<double> b, c, lb, ub, xp, x, lam, nu, v, u, w, q, r, dir, rhs, part, scal;
    // one block per rank over a caller-owned row block (slp_blocks_create_on)
    IterGraph cg_graph;       // `check_every` CG steps captured once (launch-bound problems)
    bool row_block = false, distributed = false;
    bool grouped = false;     // linked into a group (slp_blocks_group_link): its copy counts are the group's totals
    i64 m_eq = 0;
    DevBuf<unsigned char> used;
    DevBuf<double> copies, acc, vs, xs, xps, lams, slo, shi;
    bool primal = false;      // projection solved in the primal form (I + A^T A), see k_rb_resid0_primal
    DevBuf<double> xsol, zero_m;
    bool precond = false;     // Jacobi-preconditioned CG (slp_blocks_set_precond); dinv = 1 / diag(S), z = preconditioned residual
    DevBuf<double> dinv, z;
    // the conjugate-gradient loop as begin / rounds (blk_cg_begin, blk_cg_round): the blocks of a group take their rounds in turn,
    // each on a stream of its own (slp_blocks_group_iterate)
    struct { i64 m = 0; double *sol = nullptr; bool pc = false; int it = 0; } cg;
    bool warmed = false;      // has run a projection on the library's stream (slp_blocks_group_iterate)
};

namespace slp {

static void blk_dot(slp_blocks *s, i64 n, const double *a, const double *b, int slot, int mode) {
    int grid = std::min(grid_for(n, kBlock), kBlkPartials);
    hipLaunchKernelGGL(k_blk_dot, dim3(grid), dim3(kBlock), 0, ctx().stream, n, a, b, s->part.p);
    hipLaunchKernelGGL(k_blk_finish, dim3(1), dim3(kBlock), 0, ctx().stream, grid, s->part.p, s->scal.p, slot, mode);
    SLP_HIP(hipGetLastError());
}

// q = S dir = A^ (A^T dir)
static void blk_apply(slp_blocks *s, const double *dir, double *q) {
    matrix_spmv(s->a, true, dir, s->u.p, SLP_ORDER_AUTO);
    matrix_spmv(s->a, false, s->u.p, q, SLP_ORDER_AUTO);
}

// the conjugate-gradient loop shared by both layouts; `apply(dir, q)` computes q = S dir.  In two pieces: blk_cg_begin (the first
// inner products) and blk_cg_round (the stopping test on the scalars of the steps so far -- one 64-byte read -- then `check_every`
// more steps enqueued; true once the test says stop or the step limit is reached).  blk_cg runs them to the end.
static void blk_cg_begin(slp_blocks *s, i64 len = -1, double *sol = nullptr) {
    hipStream_t st = ctx().stream;
    const i64 m = len < 0 ? s->m : len;   // length of the system (rows: dual form; original variables: primal form)
    if (!sol) sol = s->nu.p;
    const int gm = grid_for(m, kBlock);
    blk_dot(s, m, s->rhs.p, s->rhs.p, B_RHS2, 0);
    const bool pc = s->precond && s->dinv.n >= (size_t)m;
    if (pc) {  // z = D^-1 r ; dir = z ; rs := r.z ; the stopping test keeps |r|^2 in its own slot
        hipLaunchKernelGGL(k_blk_precond, dim3(gm), dim3(kBlock), 0, st, m, s->dinv.p, s->r.p, s->z.p);
        SLP_HIP(hipMemcpyAsync(s->dir.p, s->z.p, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, st));
        blk_dot(s, m, s->r.p, s->z.p, B_RS, 0);
        blk_dot(s, m, s->r.p, s->r.p, B_RR, 0);
    } else {
        blk_dot(s, m, s->r.p, s->r.p, B_RS, 0);
    }
    s->cg.m = m; s->cg.sol = sol; s->cg.pc = pc; s->cg.it = 0;
}

template <class Apply>
static bool blk_cg_round(slp_blocks *s, Apply apply, bool may_capture = true) {
    hipStream_t st = ctx().stream;
    const i64 m = s->cg.m;
    double *sol = s->cg.sol;
    const bool pc = s->cg.pc;
    const int gm = grid_for(m, kBlock);
    if (s->cg.it >= s->max_cg) return true;
    double h[B_COUNT];
    s->scal.download(h, B_COUNT);  // one 64-byte read every `check_every` steps
    if (!(h[pc ? B_RR : B_RS] > s->tol * s->tol * h[B_RHS2])) return true;
    auto step = [&]() {
        apply(s->dir.p, s->q.p);
        blk_dot(s, m, s->dir.p, s->q.p, B_PQ, 1);
        hipLaunchKernelGGL(k_blk_step, dim3(gm), dim3(kBlock), 0, st, m, s->scal.p, s->dir.p, s->q.p, sol, s->r.p);
        if (pc) {
            hipLaunchKernelGGL(k_blk_precond, dim3(gm), dim3(kBlock), 0, st, m, s->dinv.p, s->r.p, s->z.p);
            blk_dot(s, m, s->r.p, s->z.p, B_RSNEW, 2);
            blk_dot(s, m, s->r.p, s->r.p, B_RR, 0);
            hipLaunchKernelGGL(k_blk_dir, dim3(gm), dim3(kBlock), 0, st, m, s->scal.p, s->z.p, s->dir.p);
        } else {
            blk_dot(s, m, s->r.p, s->r.p, B_RSNEW, 2);
            hipLaunchKernelGGL(k_blk_dir, dim3(gm), dim3(kBlock), 0, st, m, s->scal.p, s->r.p, s->dir.p);
        }
    };
    const int chunk = std::min(s->check_every, s->max_cg - s->cg.it);
    // all kernel arguments are fixed pointers.  Not while an asynchronous RCCL all-reduce of the previous block is in flight: what
    // the library's own threads call meanwhile is not ours to order against an open capture (see capture_mutex, slp_common.h);
    // not on a group's side streams either (the graph was captured on the library's stream)
    if (may_capture && s->a->a.nnz <= 20000000 && !comm_library_collective_pending()) s->cg_graph.run(chunk, s->check_every, step);
    else for (int k = 0; k < chunk; ++k) step();
    s->cg.it += chunk;
    s->cg_steps += chunk;
    return false;
}

template <class Apply>
static void blk_cg(slp_blocks *s, Apply apply, i64 len = -1, double *sol = nullptr) {
    blk_cg_begin(s, len, sol);
    while (!blk_cg_round(s, apply)) {}
}

// One block's share of an iteration up to the exchange: its projection (matrix-free CG, no exchange inside), the
// over-relaxed x, the slack update and the consensus summand acc_j = used_j ? x_j + lambda_j / gamma : 0.
// In three pieces (begin / rounds of conjugate-gradient steps / end) so that the blocks of a group can take their rounds in turn
// on streams of their own; rb_project runs them back to back on the library's stream.
static void rb_apply_dual(slp_blocks *s, const double *dir, double *q) {   // q = (A A^T + [0; I]) dir
    blk_apply(s, dir, q);
    if (s->m > s->m_eq)
        hipLaunchKernelGGL(k_rb_add_identity, dim3(grid_for(s->m - s->m_eq, kBlock)), dim3(kBlock), 0, ctx().stream, s->m, s->m_eq, dir, q);
}
static void rb_apply_primal(slp_blocks *s, const double *dir, double *q) {  // q = dir + A^T (A dir)
    matrix_spmv(s->a, false, dir, s->w.p, SLP_ORDER_AUTO);
    matrix_spmv(s->a, true, s->w.p, q, SLP_ORDER_AUTO);
    hipLaunchKernelGGL(k_rb_add_vec, dim3(grid_for(s->N, kBlock)), dim3(kBlock), 0, ctx().stream, s->N, dir, q);
}

static void rb_project_begin(slp_blocks *s) {
    hipStream_t st = ctx().stream;
    const i64 n = s->N, m = s->m, me = s->m_eq;
    const int gn = grid_for(n, kBlock), gm = grid_for(m, kBlock);
    hipLaunchKernelGGL(k_rb_v, dim3(gn), dim3(kBlock), 0, st, n, s->xp.p, s->lam.p, s->gamma, s->v.p);
    if (s->primal) {
        hipLaunchKernelGGL(k_rb_vs, dim3(gm), dim3(kBlock), 0, st, m, s->xps.p, s->lams.p, s->gamma, s->vs.p);
        matrix_spmv(s->a, true, s->vs.p, s->u.p, SLP_ORDER_AUTO);
        matrix_spmv(s->a, false, s->xsol.p, s->w.p, SLP_ORDER_AUTO);
        matrix_spmv(s->a, true, s->w.p, s->q.p, SLP_ORDER_AUTO);
        hipLaunchKernelGGL(k_rb_resid0_primal, dim3(gn), dim3(kBlock), 0, st, n, s->v.p, s->u.p, s->xsol.p, s->q.p, s->r.p, s->dir.p, s->rhs.p);
        blk_cg_begin(s, n, s->xsol.p);
    } else if (m > 0) {
        matrix_spmv(s->a, false, s->v.p, s->w.p, SLP_ORDER_AUTO);
        blk_apply(s, s->nu.p, s->q.p);
        hipLaunchKernelGGL(k_rb_resid0, dim3(gm), dim3(kBlock), 0, st, m, me, s->w.p, s->b.p, s->q.p, s->nu.p, s->xps.p, s->lams.p, s->gamma,
                           s->vs.p, s->r.p, s->dir.p, s->rhs.p);
        blk_cg_begin(s);
    }
    SLP_HIP(hipGetLastError());
}

// the next `check_every` conjugate-gradient steps of the block's projection; true when it has none left
static bool rb_project_round(slp_blocks *s, bool may_capture = true) {
    if (s->primal) return blk_cg_round(s, [&](const double *dir, double *q) { rb_apply_primal(s, dir, q); }, may_capture);
    if (s->m > 0) return blk_cg_round(s, [&](const double *dir, double *q) { rb_apply_dual(s, dir, q); }, may_capture);
    return true;
}

static void rb_project_end(slp_blocks *s) {
    hipStream_t st = ctx().stream;
    const i64 n = s->N, m = s->m, me = s->m_eq;
    const int gn = grid_for(n, kBlock), gm = grid_for(m, kBlock);
    if (s->primal) {
        matrix_spmv(s->a, false, s->xsol.p, s->w.p, SLP_ORDER_AUTO);  // the projected slacks s = A x
        hipLaunchKernelGGL(k_rb_x_primal, dim3(gn), dim3(kBlock), 0, st, n, s->used.p, s->xp.p, s->xsol.p, s->lam.p, s->alpha, 1.0 - s->alpha,
                           s->gamma, s->x.p, s->acc.p);
        hipLaunchKernelGGL(k_rb_slack, dim3(gm), dim3(kBlock), 0, st, m, (i64)0, s->w.p, s->zero_m.p, s->slo.p, s->shi.p, s->alpha,
                           1.0 - s->alpha, s->gamma, s->xs.p, s->xps.p, s->lams.p);
        SLP_HIP(hipGetLastError());
        return;
    }
    if (m > 0) matrix_spmv(s->a, true, s->nu.p, s->u.p, SLP_ORDER_AUTO);
    else s->u.zero();
    hipLaunchKernelGGL(k_rb_x, dim3(gn), dim3(kBlock), 0, st, n, s->used.p, s->xp.p, s->v.p, s->u.p, s->lam.p, s->alpha, 1.0 - s->alpha,
                       s->gamma, s->x.p, s->acc.p);
    if (m > me)
        hipLaunchKernelGGL(k_rb_slack, dim3(grid_for(m - me, kBlock)), dim3(kBlock), 0, st, m, me, s->vs.p, s->nu.p, s->slo.p, s->shi.p,
                           s->alpha, 1.0 - s->alpha, s->gamma, s->xs.p, s->xps.p, s->lams.p);
    SLP_HIP(hipGetLastError());
}

static void rb_project(slp_blocks *s) {
    rb_project_begin(s);
    while (!rb_project_round(s)) {}
    rb_project_end(s);
}

// ... and after it: xp = clamp((sum of the summands over every block of every rank - c / gamma) / copies), this block's lambda
static void rb_consensus(slp_blocks *s, const double *acc_total) {
    const i64 n = s->N;
    hipLaunchKernelGGL(k_rb_consensus, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, ctx().stream, n, s->used.p, s->copies.p, acc_total, s->c.p,
                       s->lb.p, s->ub.p, s->x.p, s->gamma, s->xp.p, s->lam.p);
    SLP_HIP(hipGetLastError());
}

static void rb_iteration(slp_blocks *s) {
    rb_project(s);
    if (s->distributed) comm_allreduce_dev(s->acc.p, s->N, 0);  // the consensus sum: the only exchange of an iteration
    rb_consensus(s, s->acc.p);
}

__global__ void k_rb_accumulate(i64 n, const double *__restrict__ a, double *__restrict__ total) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) total[j] = total[j] + a[j];
}

static void blk_iteration(slp_blocks *s) {
    if (s->row_block) { rb_iteration(s); return; }
    hipStream_t st = ctx().stream;
    const i64 P = s->P, m = s->m, N = s->N;
    const int gp = grid_for(P, kBlock), gm = grid_for(m, kBlock), gn = grid_for(N, kBlock);
    hipLaunchKernelGGL(k_blk_v, dim3(gp), dim3(kBlock), 0, st, P, s->owner.p, s->xp.p, s->lam.p, s->gamma, s->v.p);
    if (m > 0) {
        // (A^ A^T) nu = A^ v - b, conjugate gradients from the previous nu
        matrix_spmv(s->a, false, s->v.p, s->w.p, SLP_ORDER_AUTO);
        blk_apply(s, s->nu.p, s->q.p);
        hipLaunchKernelGGL(k_blk_resid0, dim3(gm), dim3(kBlock), 0, st, m, s->w.p, s->b.p, s->q.p, s->r.p, s->dir.p, s->rhs.p);
        blk_cg(s, [&](const double *dir, double *q) { blk_apply(s, dir, q); });
        matrix_spmv(s->a, true, s->nu.p, s->u.p, SLP_ORDER_AUTO);
    } else {
        s->u.zero();
    }
    hipLaunchKernelGGL(k_blk_x, dim3(gp), dim3(kBlock), 0, st, P, s->owner.p, s->xp.p, s->v.p, s->u.p, s->alpha, 1.0 - s->alpha, s->x.p);
    hipLaunchKernelGGL(k_blk_consensus, dim3(gn), dim3(kBlock), 0, st, N, s->cptr.p, s->cidx.p, s->x.p, s->lam.p, s->c.p, s->lb.p, s->ub.p,
                       s->gamma, s->xp.p);
    hipLaunchKernelGGL(k_blk_lambda, dim3(gp), dim3(kBlock), 0, st, P, s->owner.p, s->x.p, s->xp.p, s->gamma, s->lam.p);
    SLP_HIP(hipGetLastError());
}

}  // namespace slp

extern "C" {

slp_blocks *slp_blocks_create(int64_t P, int64_t m, int64_t N, const int64_t *indptr, const int32_t *indices, const double *data,
                              const double *b, const double *c, const double *lb, const double *ub, const double *xp0,
                              const int32_t *owner, const int64_t *copy_ptr, const int32_t *copy_idx, double gamma) {
    SLP_API_PTR({
        SLP_REQUIRE(P >= 0 && m >= 0 && N >= 0 && indptr && b && c && lb && ub && xp0 && owner && copy_ptr && copy_idx,
                    "slp_blocks_create: NULL argument");
        SLP_REQUIRE(gamma > 0.0, "slp_blocks_create: gamma must be positive");
        auto *s = new slp_blocks();
        try {
            s->a = slp_matrix_create(m, P, indptr, indices, data);
            if (!s->a) throw Error(slp_last_error());
            build_transpose(s->a);
            fast_format(s->a, false);  // derived formats are settled here, never inside a captured CG step
            fast_format(s->a, true);
            s->P = P; s->m = m; s->N = N; s->gamma = gamma;
            s->owner.upload(owner, (size_t)P);
            s->cptr.upload(copy_ptr, (size_t)N + 1);
            s->cidx.upload(copy_idx, (size_t)P);
            s->b.upload(b, (size_t)m); s->c.upload(c, (size_t)N); s->lb.upload(lb, (size_t)N); s->ub.upload(ub, (size_t)N);
            s->xp.upload(xp0, (size_t)N);
            const size_t sp = (size_t)P, sm = (size_t)m;
            s->x.alloc(sp); s->lam.alloc(sp); s->lam.zero(); s->v.alloc(sp); s->u.alloc(sp);
            s->nu.alloc(sm); s->nu.zero(); s->w.alloc(sm); s->q.alloc(sm); s->r.alloc(sm); s->dir.alloc(sm); s->rhs.alloc(sm);
            s->part.alloc(kBlkPartials); s->scal.alloc(B_COUNT); s->scal.zero();
            SLP_HIP(hipStreamSynchronize(ctx().stream));
        } catch (...) {
            delete s->a;
            delete s;
            throw;
        }
        return s;
    })
}

slp_blocks *slp_blocks_create_on(slp_matrix *a, int64_t m_eq, const double *b_lower, const double *b_upper, const double *c,
                                 const double *lb, const double *ub, double gamma) {
    SLP_API_PTR({
        SLP_REQUIRE(a && b_upper && c && lb && ub, "slp_blocks_create_on: NULL argument");
        SLP_REQUIRE(m_eq >= 0 && m_eq <= a->a.nrow, "slp_blocks_create_on: m_eq out of range");
        SLP_REQUIRE(gamma > 0.0, "slp_blocks_create_on: gamma must be positive");
        // A matrix without CSR entries (a chunked matrix, or one whose CSR was released) serves as long as both products run on
        // its strip / tall-cell copies: the block ADMM only ever multiplies (BASELINE config 5: eight 5e5 x 5e7 blocks on one
        // GPU exist as 26 GB of tall cells each, never as 60 GB of CSR).
        const bool csrless = !a->chunks.empty() || a->csr_released;
        if (csrless)
            SLP_REQUIRE(fast_format(a, false) && fast_format(a, true),
                        "slp_blocks_create_on: the CSR entries of this matrix are gone and it has no strip copies in both orientations");
        auto *s = new slp_blocks();
        try {
            hipStream_t st = ctx().stream;
            const i64 m = a->a.nrow, n = a->a.ncol;
            s->a = a; s->row_block = true; s->m = m; s->N = n; s->P = n; s->m_eq = m_eq; s->gamma = gamma;
            s->distributed = comm_active();
            if (!csrless) {
                ensure_transposed(a);  // a copy of A^T (tall cells come straight from the CSR of A), else the transposed CSR
                fast_format(a, false);
                fast_format(a, true);
            }
            const size_t sn = (size_t)n, sm = (size_t)m;
            // standard form (tools.py:88-127): b = [b_eq; 0], slack bounds [b_lower, b_upper]; x0 = 0 so xp0 = clamp(0) (:84-86)
            std::vector<double> hb(sm, 0.0), hlo(sm, 0.0), hhi(sm, 0.0), hx(sn), hs(sm, 0.0);
            for (i64 i = 0; i < m; ++i) {
                if (i < m_eq) { hb[(size_t)i] = b_upper[i]; continue; }
                hlo[(size_t)i] = b_lower ? b_lower[i] : -INFINITY;
                hhi[(size_t)i] = b_upper[i];
                hs[(size_t)i] = std::min(std::max(0.0, hlo[(size_t)i]), hhi[(size_t)i]);
            }
            for (i64 j = 0; j < n; ++j) hx[(size_t)j] = std::min(std::max(0.0, lb[j]), ub[j]);
            s->b.upload(hb.data(), sm); s->slo.upload(hlo.data(), sm); s->shi.upload(hhi.data(), sm); s->xps.upload(hs.data(), sm);
            s->c.upload(c, sn); s->lb.upload(lb, sn); s->ub.upload(ub, sn); s->xp.upload(hx.data(), sn);
            s->x.alloc(sn); s->lam.alloc(sn); s->lam.zero(); s->v.alloc(sn); s->u.alloc(sn); s->acc.alloc(sn);
            s->used.alloc(sn); s->copies.alloc(sn);
            // all rows inequalities and m >= n: the better-conditioned primal form of the projection (SLP_BLOCKS_PRIMAL=0/1 forces)
            const char *ep = getenv("SLP_BLOCKS_PRIMAL");
            s->primal = m_eq == 0 && m > 0 && (ep ? ep[0] == '1' : m >= n);
            const size_t sv = s->primal ? sn : sm;  // the CG runs over the rows (dual form) or over the original variables (primal form)
            s->nu.alloc(sm); s->nu.zero(); s->w.alloc(sm); s->q.alloc(sv); s->r.alloc(sv); s->dir.alloc(sv); s->rhs.alloc(sv);
            if (s->primal) { s->xsol.alloc(sn); s->xsol.zero(); s->zero_m.alloc(sm); s->zero_m.zero(); }
            s->vs.alloc(sm); s->xs.alloc(sm); s->xs.zero(); s->lams.alloc(sm); s->lams.zero();
            s->part.alloc(kBlkPartials); s->scal.alloc(B_COUNT); s->scal.zero();
            // which columns this row block uses (a copy exists only for those, :183-185) and in how many ranks' blocks
            const StripJds *ft = a->a.nnz > 0 ? fast_format(a, true) : nullptr;
            if (!(a->at.ptr.p && a->at.ptr.n == sn + 1) && ft && strip_abs_pow_supported(*ft)) {
                // no transposed CSR: the column counts are the product (|A|^0)^T 1 over the copy of A^T (exact: sums of ones)
                DevBuf<double> ones(sm), cnt(sn);
                hipLaunchKernelGGL(k_rb_fill, dim3(grid_for(m, kBlock)), dim3(kBlock), 0, st, m, 1.0, ones.p);
                strip_spmv_abs_pow(*ft, 0.0, ones.p, cnt.p);
                hipLaunchKernelGGL(k_rb_used_counts, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, n, cnt.p, s->used.p, s->copies.p);
                SLP_HIP(hipGetLastError());
                SLP_HIP(hipStreamSynchronize(st));  // (ones / cnt go back to the cache)
            } else {
                build_transpose(a);
                hipLaunchKernelGGL(k_rb_used, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, n, a->at.ptr.p, s->used.p, s->copies.p);
            }
            SLP_HIP(hipGetLastError());
            if (s->distributed) comm_allreduce_dev(s->copies.p, n, 0);
            SLP_HIP(hipStreamSynchronize(st));
        } catch (...) {
            delete s;
            throw;
        }
        ++a->borrowers;
        return s;
    })
}

void slp_blocks_destroy(slp_blocks *s) {
    if (!s) return;
    if (!s->row_block) delete s->a;
    else --s->a->borrowers;
    delete s;
}

int slp_blocks_projection_residual(slp_blocks *s, double out[2]) {
    SLP_API_INT({
        SLP_REQUIRE(s && out && s->row_block, "slp_blocks_projection_residual: a solver from slp_blocks_create_on");
        hipStream_t st = ctx().stream;
        const i64 n = s->N, m = s->m, me = s->m_eq;
        const i64 len = s->primal ? n : m;
        if (len == 0) { out[0] = out[1] = 0.0; return 0; }
        // the TRUE residual of the projection system of the last block update, rhs - S sol, with the operator applied afresh
        // (the conjugate-gradient loop only ever sees its recurrence)
        if (s->primal) {
            matrix_spmv(s->a, false, s->xsol.p, s->w.p, SLP_ORDER_AUTO);
            matrix_spmv(s->a, true, s->w.p, s->q.p, SLP_ORDER_AUTO);
            hipLaunchKernelGGL(k_rb_add_vec, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, n, s->xsol.p, s->q.p);
        } else {
            blk_apply(s, s->nu.p, s->q.p);
            if (m > me) hipLaunchKernelGGL(k_rb_add_identity, dim3(grid_for(m - me, kBlock)), dim3(kBlock), 0, st, m, me, s->nu.p, s->q.p);
        }
        hipLaunchKernelGGL(k_rb_true_resid, dim3(grid_for(len, kBlock)), dim3(kBlock), 0, st, len, s->rhs.p, s->q.p, s->r.p);
        SLP_HIP(hipGetLastError());
        blk_dot(s, len, s->r.p, s->r.p, B_RR, 0);
        blk_dot(s, len, s->rhs.p, s->rhs.p, B_RHS2, 0);
        double h[B_COUNT];
        s->scal.download(h, B_COUNT);
        out[0] = sqrt(h[B_RR]);
        out[1] = sqrt(h[B_RHS2]);
    })
}

int slp_blocks_set_cg(slp_blocks *s, double tol, int max_steps) {
    SLP_API_INT({
        SLP_REQUIRE(s && tol > 0.0 && max_steps > 0, "slp_blocks_set_cg: bad arguments");
        s->tol = tol;
        s->max_cg = max_steps;
    })
}

int slp_blocks_set_precond(slp_blocks *s, int jacobi) {
    SLP_API_INT({
        SLP_REQUIRE(s && s->row_block, "slp_blocks_set_precond: a solver from slp_blocks_create_on is required");
        s->precond = jacobi != 0;
        s->cg_graph.reset();
        if (!s->precond) return 0;
        const i64 len = s->primal ? s->N : s->m;
        require_csr(s->a, "slp_blocks_set_precond");
        if (s->primal) build_transpose(s->a);  // the column norms walk the transposed CSR
        s->dinv.alloc((size_t)len);
        s->z.alloc((size_t)len);
        const CsrDev &c = s->primal ? s->a->at : s->a->a;  // primal: 1 + column norms^2; dual: row norms^2 (+ 1 on inequality rows)
        if (len)
            hipLaunchKernelGGL(k_blk_diag, dim3(grid_for(len, kBlock)), dim3(kBlock), 0, ctx().stream, len, s->primal ? (i64)0 : s->m_eq,
                               c.ptr.p, c.val.p, 1.0, s->dinv.p);
        SLP_HIP(hipGetLastError());
        SLP_HIP(hipStreamSynchronize(ctx().stream));
    })
}

int slp_blocks_iterate(slp_blocks *s, int64_t k) {
    SLP_API_INT({
        SLP_REQUIRE(s && k >= 0, "slp_blocks_iterate: bad arguments");
        for (i64 it = 0; it < k; ++it) blk_iteration(s);
    })
}

// ---- several row blocks on one rank (ADMMBlocks.py's `blocks` metadata at scale) ---------------------------------------
// Every block is its own slp_blocks over its own row-block matrix (slp_blocks_create_on): own copy of the variables it
// uses, own multipliers, own projection.  One iteration of the group: every block's projection; with the rows partitioned
// over several ranks each block's summand is all-reduced on a second stream WHILE the next block's projection computes
// (asynchronous block updates overlapped with the exchange: G all-reduces of n doubles, all but the last hidden); the
// reduced summands added in block order; the consensus update in every block.
int slp_blocks_group_link(slp_blocks **blocks, int count) {
    SLP_API_INT({
        SLP_REQUIRE(blocks && count >= 1, "slp_blocks_group_link: bad arguments");
        for (int g = 0; g < count; ++g) {
            SLP_REQUIRE(blocks[g] && blocks[g]->row_block, "slp_blocks_group_link: blocks must come from slp_blocks_create_on");
            SLP_REQUIRE(blocks[g]->N == blocks[0]->N && blocks[g]->gamma == blocks[0]->gamma, "slp_blocks_group_link: blocks differ in n / gamma");
            // linking replaces every block's copy counts by the group totals: a second link (or a block in two groups) would
            // count its copies twice and the consensus would divide by the wrong number, silently
            SLP_REQUIRE(!blocks[g]->grouped, "slp_blocks_group_link: a block is already part of a group");
            for (int h = 0; h < g; ++h) SLP_REQUIRE(blocks[h] != blocks[g], "slp_blocks_group_link: the same block twice");
        }
        if (blocks[0]->distributed) {
            // the overlapped iteration issues `count` all-reduces per iteration: every rank must bring the same number of
            // blocks, or the ranks deadlock inside the collective library instead of failing here
            double mm[2] = {(double)count, -(double)count};
            SLP_REQUIRE(slp_comm_allreduce_host(mm, 2, 1) == 0, slp_last_error());
            SLP_REQUIRE(mm[0] == (double)count && -mm[1] == (double)count, "slp_blocks_group_link: the ranks link different numbers of blocks");
        }
        // copies_j = number of blocks (of all ranks) that use variable j: each block holds its own count summed over the
        // ranks (slp_blocks_create_on); the group total goes into every block
        const i64 n = blocks[0]->N;
        DevBuf<double> total((size_t)n);
        total.zero();
        for (int g = 0; g < count; ++g)
            hipLaunchKernelGGL(k_rb_accumulate, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, ctx().stream, n, blocks[g]->copies.p, total.p);
        SLP_HIP(hipGetLastError());
        for (int g = 0; g < count; ++g) { blocks[g]->copies.copy_from(total); blocks[g]->grouped = true; }
        SLP_HIP(hipStreamSynchronize(ctx().stream));
    })
}

// The blocks' projections of ONE rank side by side (round 6; VERDICT r05 #5: "one grid over the eight blocks' products").  The
// blocks are independent inside an iteration (ADMMBlocks.py:264-307), but each projection is a chain of ~40 products with a
// stopping test on the host every `check_every` steps, and a product of config 5's shape is ONE round of 255 workgroups on 256
// compute units: block after block every launch ends at its slowest workgroup.  Here every block has a stream of its own and the
// blocks take their rounds of conjugate-gradient steps in turn (begin all; round-robin: read a block's scalars, test, enqueue its
// next `check_every` steps; end): the queues hold several blocks' products at once and a compute unit that finishes its workgroup
// of one block starts one of the next -- what a batched grid would do, without rewriting the loop around per-block scalars in
// device memory.  Every block's arithmetic is what it was (same kernels, same order on its stream): results bit for bit, step
// counts equal.  MEASURED at config 5 (tools/lab/c5_streams_ab.sh, profiles/r06_c5_streams_ab.log, interleaved on one box):
// 0.9602 / 0.9605 it/s block after block, 0.9385 / 0.9383 side by side -- 2.3 % SLOWER (two / four blocks at a time: 1.0 / 1.2 %),
// the objective equal to the last bit: the more products in flight, the slower -- there is no launch tail to win back (a product's
// 255 workgroups are balanced by construction), and workgroups of different blocks on one chip get in each other's way.  (Not
// through x: a lab build whose x-tiles all come from one 4 MB window runs a single product of the shape in 3.02 ms against 3.03.)
// So it is OPT-IN (SLP_BLOCKS_STREAMS=1 | w; the test of the path sets it), block after block is the default, and the batched grid
// this stands in for has nothing to gain either.  Never under a communicator (there the blocks' all-reduces overlap the next block's projection
// instead), never captured graphs on the side streams, and not for a group's first iteration (whatever a block builds lazily is
// built on the library's stream, where the caching allocator's stream order holds).
static std::vector<hipStream_t> g_blk_streams;
static std::vector<hipEvent_t> g_blk_events;   // [count] = the library's stream at the start of the iteration
static bool blocks_side_by_side(slp_blocks **blocks, int count) {
    if (count < 2 || blocks[0]->distributed) return false;
    for (int g = 0; g < count; ++g)
        if (!blocks[g]->warmed) return false;
    const char *e = getenv("SLP_BLOCKS_STREAMS");
    return e && atoi(e) >= 1;
}
// SLP_BLOCKS_STREAMS=1: all blocks side by side; = w >= 2: w at a time (block g on stream g % w, the groups one after the other)
static void blocks_project_side_by_side(slp_blocks **blocks, int count) {
    hipStream_t main_stream = ctx().stream;
    const char *e = getenv("SLP_BLOCKS_STREAMS");
    const int want = e ? atoi(e) : 1, width = want >= 2 && want < count ? want : count;
    while ((int)g_blk_streams.size() < count) {
        hipStream_t st = nullptr;
        SLP_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        g_blk_streams.push_back(st);
    }
    while ((int)g_blk_events.size() < count + 1) {
        hipEvent_t ev = nullptr;
        SLP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g_blk_events.push_back(ev);
    }
    SLP_HIP(hipEventRecord(g_blk_events[count], main_stream));
    try {
        for (int g = 0; g < width; ++g) SLP_HIP(hipStreamWaitEvent(g_blk_streams[g], g_blk_events[count], 0));
        for (int g0 = 0; g0 < count; g0 += width) {
            const int g1 = std::min(count, g0 + width);
            for (int g = g0; g < g1; ++g) {
                ctx().stream = g_blk_streams[g % width];
                rb_project_begin(blocks[g]);
            }
            std::vector<char> active((size_t)count, 1);
            for (int left = g1 - g0; left > 0;)
                for (int g = g0; g < g1; ++g) {
                    if (!active[g]) continue;
                    ctx().stream = g_blk_streams[g % width];
                    if (!rb_project_round(blocks[g], false)) continue;
                    rb_project_end(blocks[g]);
                    SLP_HIP(hipEventRecord(g_blk_events[g], g_blk_streams[g % width]));
                    active[g] = 0;
                    --left;
                }
        }
    } catch (...) {
        ctx().stream = main_stream;
        for (int g = 0; g < count; ++g) (void)hipStreamSynchronize(g_blk_streams[g]);   // nothing of the blocks may still run when the error unwinds
        throw;
    }
    ctx().stream = main_stream;
    for (int g = 0; g < count; ++g) SLP_HIP(hipStreamWaitEvent(main_stream, g_blk_events[g], 0));
}

int slp_blocks_group_iterate(slp_blocks **blocks, int count, int64_t k) {
    SLP_API_INT({
        SLP_REQUIRE(blocks && count >= 1 && k >= 0, "slp_blocks_group_iterate: bad arguments");
        slp_blocks *s0 = blocks[0];
        const i64 n = s0->N;
        for (i64 it = 0; it < k; ++it) {
            if (blocks_side_by_side(blocks, count)) {
                blocks_project_side_by_side(blocks, count);
                for (int g = 1; g < count; ++g)  // fixed order: deterministic sums
                    hipLaunchKernelGGL(k_rb_accumulate, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, ctx().stream, n, blocks[g]->acc.p, s0->acc.p);
                SLP_HIP(hipGetLastError());
                for (int g = 0; g < count; ++g) rb_consensus(blocks[g], s0->acc.p);
                continue;
            }
            // Asynchronous block updates: as soon as a block's summand exists its all-reduce starts on the second stream and
            // travels over xGMI while the NEXT block's projection computes; only the last block's exchange is exposed.
            // (One rank, or one block: a single all-reduce of the sum, as before.)
            const bool overlap = s0->distributed && count > 1;
            try {
                for (int g = 0; g < count; ++g) {
                    rb_project(blocks[g]);
                    blocks[g]->warmed = true;
                    if (overlap) comm_allreduce_dev_async(blocks[g]->acc.p, n, 0);
                }
            } catch (...) {
                comm_sync_side();  // no asynchronous all-reduce may still be writing a summand when the error unwinds
                throw;
            }
            if (overlap) comm_join();
            for (int g = 1; g < count; ++g)  // fixed order: deterministic sums
                hipLaunchKernelGGL(k_rb_accumulate, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, ctx().stream, n, blocks[g]->acc.p, s0->acc.p);
            SLP_HIP(hipGetLastError());
            if (s0->distributed && !overlap) comm_allreduce_dev(s0->acc.p, n, 0);
            for (int g = 0; g < count; ++g) rb_consensus(blocks[g], s0->acc.p);
        }
    })
}

int slp_blocks_report(slp_blocks *s, double out[2]) {
    SLP_API_INT({
        SLP_REQUIRE(s && out, "slp_blocks_report: NULL argument");
        hipStream_t st = ctx().stream;
        double h[B_COUNT];
        blk_dot(s, s->N, s->c.p, s->xp.p, B_PQ, 0);
        s->scal.download(h, B_COUNT);
        double e = h[B_PQ];
        if (s->row_block) {
            int grid = std::min(grid_for(std::max(s->N, s->m), kBlock), kBlkPartials);
            hipLaunchKernelGGL(k_rb_energy, dim3(grid), dim3(kBlock), 0, st, s->N, s->m, s->m_eq, s->used.p, s->x.p, s->xp.p, s->lam.p,
                               s->xs.p, s->xps.p, s->lams.p, s->gamma, s->part.p);
            hipLaunchKernelGGL(k_blk_finish, dim3(1), dim3(kBlock), 0, st, grid, s->part.p, s->scal.p, (int)B_PQ, 0);
            SLP_HIP(hipGetLastError());
            s->scal.download(h, B_COUNT);
            double blocks = h[B_PQ];
            if (s->distributed) SLP_REQUIRE(slp_comm_allreduce_host(&blocks, 1, 0) == 0, slp_last_error());
            e += blocks;
        } else if (s->P > 0) {
            int grid = std::min(grid_for(s->P, kBlock), kBlkPartials);
            hipLaunchKernelGGL(k_blk_energy, dim3(grid), dim3(kBlock), 0, st, s->P, s->owner.p, s->x.p, s->xp.p, s->lam.p, s->gamma,
                               s->part.p);
            hipLaunchKernelGGL(k_blk_finish, dim3(1), dim3(kBlock), 0, st, grid, s->part.p, s->scal.p, (int)B_PQ, 0);
            SLP_HIP(hipGetLastError());
            s->scal.download(h, B_COUNT);
            e += h[B_PQ];
        }
        out[0] = e;                      // ADMMBlocks.py:246-253
        out[1] = (double)s->cg_steps;    // conjugate-gradient steps taken so far
    })
}

int64_t slp_blocks_cg_steps(const slp_blocks *s) { return s ? (int64_t)s->cg_steps : -1; }  // local, no exchange

int slp_blocks_get_xp(slp_blocks *s, double *xp, int64_t count) {
    SLP_API_INT({
        SLP_REQUIRE(s && xp && count >= 0 && count <= s->N, "slp_blocks_get_xp: bad arguments");
        s->xp.download(xp, (size_t)count);
    })
}

}  // extern "C"
