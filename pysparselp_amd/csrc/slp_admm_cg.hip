// slp_admm_cg.hip -- ADMM with the reference's conjugate-gradient x-step
// (ADMM.py:182-201 + conjugateGradientLinearSolver.py:30-52, the branch its
// hard-coded flags at ADMM.py:66-71 select with use_cg=True), matrix-free:
//     M v = gamma_eq A^T (A v) + gamma_ineq v
// so that it exists where M = gamma_eq A^T A + gamma_ineq I cannot be formed
// (1e6 x 2e6 at 1000 entries per row: M would be dense).  Ten passes over the
// constraint matrix per iteration (5 x A v, 5 x A^T w), everything else is
// elementwise work and dot products on length-N vectors.
//
// The standard-form matrix is A = [ At | diag(sc) ]: `At` is a CSR block over
// the n_o original variables (both orientations resident), the optional
// diagonal block holds one slack variable per row (ADMM.py:84-86: [A_ineq -I],
// row-scaled).  Without `sc` the caller passes the whole standard-form matrix
// as `At` (host path, any mix of equalities and inequalities).
//
// Multi-GPU: rows (and their slack variables) are partitioned; the n_o original
// variables are replicated.  Each A^T w needs one all-reduce of n_o partial
// sums; each dot product one scalar all-reduce of its slack part.
#include <cmath>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {
bool comm_active();
void comm_allreduce_dev(double *buf, i64 count, int op);
void comm_reduce_scatter_dev(double *buf, i64 cnt);
void comm_all_gather_dev(double *buf, i64 cnt);
int comm_rank();
int comm_size();

constexpr int kCgPartials = 2048;
constexpr int kCgTail = 8;       // doubles reserved behind u / u2 for the packed exchange
enum { S_T = 0, S_DMD = 2, S_RS = 4, S_PAP = 6, S_TMP = 8, S_COUNT = 32 };

// w_i = At_i . v_o + sc_i v_s,i          (A v)
template <int L>
__global__ __launch_bounds__(kBlock) void k_cg_rows(i64 m, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                    const double *__restrict__ val, const double *__restrict__ v,
                                                    const double *__restrict__ sc, i64 n_o, double *__restrict__ w) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 i = group; i < m; i += ngroups) {
        double s = row_dot<L>(ptr, idx, val, v, i, sub);
        if (sub == 0) {
            if (sc) s = s + sc[i] * v[n_o + i];
            w[i] = s;
        }
    }
}

// u_j = sum_i At_ij w_i  for the n_o original variables (partial sum on this rank)
template <int L>
__global__ __launch_bounds__(kBlock) void k_cg_cols(i64 n_o, const i64 *__restrict__ tptr, const i32 *__restrict__ tidx,
                                                    const double *__restrict__ tval, const double *__restrict__ w,
                                                    double *__restrict__ u) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 j = group; j < n_o; j += ngroups) {
        const double s = row_dot<L>(tptr, tidx, tval, w, j, sub);
        if (sub == 0) u[j] = s;
    }
}

// (A^T w)_j for any j in [0, N): column sums for the original variables, sc_i w_i for slack i
__device__ __forceinline__ double at_elem(i64 j, i64 n_o, const double *__restrict__ u, const double *__restrict__ sc,
                                          const double *__restrict__ w) {
    return j < n_o ? u[j] : sc[j - n_o] * w[j - n_o];
}

// Elementwise passes over the N unknowns.  Each also produces one dot product, split into the part over
// the replicated original variables (slot) and the part over this rank's slack variables (slot + 1).
enum { E_RHS = 0, E_LINE_T = 1, E_LINE_DMD = 2, E_LINE_STEP = 3, E_RESID = 4, E_PAP = 5, E_UPDATE = 6, E_PROJECT = 7, E_RESID_REUSE = 8, E_GRAD = 9, E_RESID_FUSED = 10, E_LINE_DMD_MD = 11,
       E_GRAD_DMD = 12,    // E_GRAD and E_LINE_DMD_MD in one pass (two dot products: part[b] and part[kCgPartials + b])
       E_STEP_RESID = 13   // E_LINE_STEP and E_RESID_FUSED in one pass (both need the line-search step only)
};

struct CgVecs {
    const double *q, *c, *lb, *ub, *u, *sc, *w;
    double *x, *xp, *y, *dir, *xprev, *r, *lin, *mx, *md;
    int keep;  // store M x and M dir for E_RESID_REUSE
    int carry_md;  // reuse level 4: E_PAP stores M r (in y), E_UPDATE advances M dir = step M dir + a_cg M r
    const double *scal;
    i64 n_o, N;
    i64 o0, o1;  // the original variables this launch walks: [0, n_o), or this rank's slice of them (sharded updates)
    double gamma_eq, gamma_ineq, alpha, one_minus_alpha;
};

// Blocks [0, go) walk the n_o replicated original variables, blocks [go, gridDim.x) this rank's slack variables: `go`
// depends on n_o only, so the original-variable part of every dot product is summed in the same order on every rank
// whatever the size of its row block (replicas must stay bit-identical), and a block's partial sum belongs to one part.
template <int OP>
__global__ __launch_bounds__(kBlock) void k_cg_elem(CgVecs a, double *__restrict__ part, int go) {
    __shared__ double lds[kBlock / kWave];
    double acc = 0.0, acc2 = 0.0;
    double f = 0.0;
    bool on = true;
    if (OP == E_LINE_STEP || OP == E_RESID_REUSE || OP == E_RESID_FUSED || OP == E_STEP_RESID) {
        const double t = -(a.scal[S_T] + a.scal[S_T + 1]);
        on = fabs(t) > 0.0;                                      // ADMM.py:192
        f = t / (a.scal[S_DMD] + a.scal[S_DMD + 1]);             // :193
    }
    if (OP == E_UPDATE) f = (a.scal[S_RS] + a.scal[S_RS + 1]) / (a.scal[S_PAP] + a.scal[S_PAP + 1]);  // conjgrad :38
    double step = 0.0;  // E_UPDATE with carry_md: the line-search step of this iteration (0 when it was skipped)
    if (OP == E_UPDATE && a.carry_md) {
        const double t = -(a.scal[S_T] + a.scal[S_T + 1]);
        step = fabs(t) > 0.0 ? t / (a.scal[S_DMD] + a.scal[S_DMD + 1]) : 0.0;
    }
    const bool slack = (int)blockIdx.x >= go;
    const i64 j_begin = slack ? a.n_o + ((i64)blockIdx.x - go) * kBlock + threadIdx.x : a.o0 + (i64)blockIdx.x * kBlock + threadIdx.x;
    const i64 j_end = slack ? a.N : a.o1;
    const i64 j_stride = (slack ? (i64)gridDim.x - go : (i64)go) * kBlock;
    for (i64 j = j_begin; j < j_end; j += j_stride) {
        double term = 0.0;
        if (OP == E_RHS) {  // y = -c + g_eq A^T b + g_ineq xp - A^T lambda_eq - lambda_ineq (:148) ; xprev = x (:184)
            const double atl = at_elem(j, a.n_o, a.u, a.sc, a.w);
            a.y[j] = ((a.q[j] + a.gamma_ineq * a.xp[j]) - atl) - a.lin[j];
            a.xprev[j] = a.x[j];
        } else if (OP == E_LINE_T) {  // t = -dir.(M x - y) (:191)
            const double mx = a.gamma_eq * at_elem(j, a.n_o, a.u, a.sc, a.w) + a.gamma_ineq * a.x[j];
            if (a.keep) a.mx[j] = mx;
            term = a.dir[j] * (mx - a.y[j]);
        } else if (OP == E_LINE_DMD) {  // dir.(M dir) (:193)
            const double md = a.gamma_eq * at_elem(j, a.n_o, a.u, a.sc, a.w) + a.gamma_ineq * a.dir[j];
            if (a.keep) a.md[j] = md;
            term = a.dir[j] * md;
        } else if (OP == E_LINE_DMD_MD) {  // the same with M dir carried by the recurrence (reuse level 4)
            term = a.dir[j] * a.md[j];
        } else if (OP == E_LINE_STEP) {  // x = x + step * dir (:194)
            if (on) a.x[j] = a.x[j] + f * a.dir[j];
        } else if (OP == E_RESID) {  // r = y - M x ; p = r ; rsold = r.r (conjgrad :33-35)
            const double mx = a.gamma_eq * at_elem(j, a.n_o, a.u, a.sc, a.w) + a.gamma_ineq * a.x[j];
            const double r = a.y[j] - mx;
            a.r[j] = r;
            term = r * r;
        } else if (OP == E_RESID_REUSE) {  // same residual from the stored products: M(x + step dir) = M x + step M dir
            const double mxn = on ? a.mx[j] + f * a.md[j] : a.mx[j];
            const double r = a.y[j] - mxn;
            a.r[j] = r;
            term = r * r;
        } else if (OP == E_GRAD) {  // fused: g = M x - y with A^T (g_eq A x + lambda_eq) taken as ONE product (u, w = v1)
            const double g = (at_elem(j, a.n_o, a.u, a.sc, a.w) + a.gamma_ineq * a.x[j]) - ((a.q[j] + a.gamma_ineq * a.xp[j]) - a.lin[j]);
            a.mx[j] = g;
            a.xprev[j] = a.x[j];
            term = a.dir[j] * g;  // t = -dir.g
        } else if (OP == E_RESID_FUSED) {  // r = y - M(x + step dir) = -(g + step M dir)
            const double r = on ? -(a.mx[j] + f * a.md[j]) : -a.mx[j];
            a.r[j] = r;
            term = r * r;
        } else if (OP == E_GRAD_DMD) {  // E_GRAD, then E_LINE_DMD_MD on the same j (same arithmetic, one pass over dir)
            const double g = (at_elem(j, a.n_o, a.u, a.sc, a.w) + a.gamma_ineq * a.x[j]) - ((a.q[j] + a.gamma_ineq * a.xp[j]) - a.lin[j]);
            a.mx[j] = g;
            a.xprev[j] = a.x[j];
            const double dj = a.dir[j];
            term = dj * g;
            acc2 += dj * a.md[j];
        } else if (OP == E_STEP_RESID) {  // E_LINE_STEP, then E_RESID_FUSED on the same j
            if (on) a.x[j] = a.x[j] + f * a.dir[j];
            const double r = on ? -(a.mx[j] + f * a.md[j]) : -a.mx[j];
            a.r[j] = r;
            term = r * r;
        } else if (OP == E_PAP) {  // p.(M p) (conjgrad :37-38)
            const double ap = a.gamma_eq * at_elem(j, a.n_o, a.u, a.sc, a.w) + a.gamma_ineq * a.r[j];
            if (a.carry_md) a.y[j] = ap;  // y is free in the fused form
            term = a.r[j] * ap;
        } else if (OP == E_UPDATE) {  // x += alpha_cg p (:39) ; speed = x - xprev (:200) ; x = 1.4 x + (1-1.4) xp (:201)
            const double xn = a.x[j] + f * a.r[j];
            if (a.carry_md) a.md[j] = step * a.md[j] + f * a.y[j];  // dir_new = step dir + a_cg r  =>  M dir_new likewise
            a.dir[j] = xn - a.xprev[j];
            a.x[j] = a.alpha * xn + a.one_minus_alpha * a.xp[j];
        } else if (OP == E_PROJECT) {  // xp = clip(x + lambda_ineq / g_ineq) ; lambda_ineq += g_ineq (x - xp) (:253-256)
            const double xj = a.x[j];
            double p = xj + a.lin[j] / a.gamma_ineq;
            const double l = a.lb[j], u = a.ub[j];
            p = (p < l) ? l : p;
            p = (p > u) ? u : p;
            a.xp[j] = p;
            a.lin[j] = a.lin[j] + a.gamma_ineq * (xj - p);
        }
        acc += term;
    }
    if (OP == E_LINE_T || OP == E_LINE_DMD || OP == E_RESID || OP == E_RESID_REUSE || OP == E_PAP || OP == E_GRAD ||
        OP == E_RESID_FUSED || OP == E_LINE_DMD_MD || OP == E_GRAD_DMD || OP == E_STEP_RESID) {
        const double r = block_reduce<false>(acc, lds);
        if (threadIdx.x == 0) part[blockIdx.x] = r;
    }
    if (OP == E_GRAD_DMD) {
        const double r = block_reduce<false>(acc2, lds);
        if (threadIdx.x == 0) part[kCgPartials + blockIdx.x] = r;
    }
}

// scal[slot] = sum of the original-variable partials (blocks [0, go)), scal[slot + 1] = this rank's slack partials.
// Packed exchange (row-partitioned level-4 iteration): the slack part was summed over the ranks in the tail of the
// preceding vector all-reduce and replaces the local one --
//   tail_mode 1: scal[slot + 1] = tail[0]
//   tail_mode 2: r = -(g + step M dir), so r.r over the slack unknowns = g.g + 2 step g.Md + step^2 Md.Md from the three
//                reduced coefficients tail[0..3) (step is known only after the exchange that carried them)
__global__ __launch_bounds__(kBlock) void k_cg_finish(int go, int nparts, const double *__restrict__ part, double *__restrict__ scal,
                                                      int slot, const double *__restrict__ tail, int tail_mode) {
    __shared__ double lds[kBlock / kWave];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < go; i += kBlock) a += part[i];
    for (int i = go + threadIdx.x; i < nparts; i += kBlock) b += part[i];
    const double ra = block_reduce<false>(a, lds), rb = block_reduce<false>(b, lds);
    if (threadIdx.x == 0) {
        scal[slot] = ra;
        double sl = rb;
        if (tail_mode == 1) sl = tail[0];
        if (tail_mode == 2) {
            const double t = -(scal[S_T] + scal[S_T + 1]);
            const double f = t / (scal[S_DMD] + scal[S_DMD + 1]);
            sl = fabs(t) > 0.0 ? (tail[0] + 2.0 * f * tail[1]) + (f * f) * tail[2] : tail[0];
        }
        scal[slot + 1] = sl;
    }
}

// k_cg_finish for the two dot products of E_GRAD_DMD: (S_T from part[.], S_DMD from part[kCgPartials + .]); with the
// packed exchange their slack parts are tail[0] and tail[1]
__global__ __launch_bounds__(kBlock) void k_cg_finish2(int go, int nparts, const double *__restrict__ part, double *__restrict__ scal,
                                                       const double *__restrict__ tail, int tail_mode) {
    __shared__ double lds[kBlock / kWave];
    for (int which = 0; which < 2; ++which) {
        const double *pp = part + which * kCgPartials;
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < go; i += kBlock) a += pp[i];
        for (int i = go + threadIdx.x; i < nparts; i += kBlock) b += pp[i];
        const double ra = block_reduce<false>(a, lds), rb = block_reduce<false>(b, lds);
        if (threadIdx.x == 0) {
            const int slot = which == 0 ? S_T : S_DMD;
            scal[slot] = ra;
            scal[slot + 1] = tail_mode == 1 ? tail[which] : rb;
        }
    }
}

// ---- sharded updates (SLP_SHARD_UPDATES=1, rows partitioned, level 4): every rank runs the elementwise passes over ITS slice of
// the original variables only; the slice's share of a dot product goes into a small exchange vector `ex` that is all-reduced:
//   ex[0] dir.g  ex[1] dir.Md   (this rank's slice)        ex[2..7) the five rank-local slack sums of k_cg_slack_pre
//   ex[8] r.r    ex[9] r.Mr     (this rank's slice)        ex[10]   the rank-local slack part of r.Mr (k_cg_slack_pap)
__global__ __launch_bounds__(kBlock) void k_cg_ex_finish(int go, const double *__restrict__ part, double *__restrict__ ex, int i0, int two) {
    __shared__ double lds[kBlock / kWave];
    for (int which = 0; which <= two; ++which) {
        double a = 0.0;
        for (int i = threadIdx.x; i < go; i += kBlock) a += part[which * kCgPartials + i];
        const double r = block_reduce<false>(a, lds);
        if (threadIdx.x == 0) ex[i0 + which] = r;
    }
}

// the iteration's scalars from the all-reduced exchange vector: mode 0 after ex[0..7), mode 1 after ex[8..11)
__global__ void k_cg_scal_from_ex(const double *__restrict__ ex, double *__restrict__ scal, int mode) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (mode == 0) {
        scal[S_T] = ex[0]; scal[S_T + 1] = ex[2];
        scal[S_DMD] = ex[1]; scal[S_DMD + 1] = ex[3];
    } else {
        const double t = -(scal[S_T] + scal[S_T + 1]);
        const double f = t / (scal[S_DMD] + scal[S_DMD + 1]);
        scal[S_RS] = ex[8];   // r = -(g + step M dir): r.r over the slack unknowns from the three reduced coefficients (k_cg_finish, mode 2)
        scal[S_RS + 1] = fabs(t) > 0.0 ? (ex[4] + 2.0 * f * ex[5]) + (f * f) * ex[6] : ex[4];
        scal[S_PAP] = ex[9]; scal[S_PAP + 1] = ex[10];
    }
}

// Packed exchange, rank-local pre-pass over this rank's slack unknowns j = n_o + i, BEFORE the all-reduce of the A^T
// product of the line search: everything the iteration needs from the slack part of its first three dot products only
// depends on rank-local data (the slack column of A^T w is sc_i w_i), so it rides behind the vector.
//   g_j  = (sc_i v1_i + g_ineq x_j) - ((q_j + g_ineq xp_j) - lin_j)          = what E_GRAD computes for j
//   Md_j = refresh ? g_eq sc_i wd_i + g_ineq dir_j (E_LINE_DMD) : carried md_j (E_LINE_DMD_MD)
// partial sums per block: [0] dir.g  [1] dir.Md  [2] g.g  [3] g.Md  [4] Md.Md
__global__ __launch_bounds__(kBlock) void k_cg_slack_pre(CgVecs a, const double *__restrict__ v1, const double *__restrict__ wd,
                                                         int refresh, double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double v[5] = {0, 0, 0, 0, 0};
    for (i64 j = a.n_o + (i64)blockIdx.x * kBlock + threadIdx.x; j < a.N; j += (i64)gridDim.x * kBlock) {
        const i64 i = j - a.n_o;
        const double g = (a.sc[i] * v1[i] + a.gamma_ineq * a.x[j]) - ((a.q[j] + a.gamma_ineq * a.xp[j]) - a.lin[j]);
        const double d = a.dir[j];
        const double md = refresh ? a.gamma_eq * (a.sc[i] * wd[i]) + a.gamma_ineq * d : a.md[j];
        v[0] += d * g;
        v[1] += d * md;
        v[2] += g * g;
        v[3] += g * md;
        v[4] += md * md;
    }
    for (int k = 0; k < 5; ++k) {
        const double r = block_reduce<false>(v[k], lds);
        if (threadIdx.x == 0) part[blockIdx.x * 5 + k] = r;
    }
}

// ... and before the all-reduce of A^T (A r): the slack part of r.(M r), M r over slack j = g_eq sc_i w_i + g_ineq r_j, w = A r
__global__ __launch_bounds__(kBlock) void k_cg_slack_pap(CgVecs a, const double *__restrict__ w, double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double v = 0.0;
    for (i64 j = a.n_o + (i64)blockIdx.x * kBlock + threadIdx.x; j < a.N; j += (i64)gridDim.x * kBlock) {
        const i64 i = j - a.n_o;
        const double ap = a.gamma_eq * (a.sc[i] * w[i]) + a.gamma_ineq * a.r[j];
        v += a.r[j] * ap;
    }
    const double r = block_reduce<false>(v, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// tail[k] = sum over blocks of part[block * K + k]  (fixed order)
__global__ __launch_bounds__(kBlock) void k_cg_tail(int nparts, int K, const double *__restrict__ part, double *__restrict__ tail) {
    __shared__ double lds[kBlock / kWave];
    for (int k = 0; k < K; ++k) {
        double a = 0.0;
        for (int i = threadIdx.x; i < nparts; i += kBlock) a += part[i * K + k];
        const double r = block_reduce<false>(a, lds);
        if (threadIdx.x == 0) tail[k] = r;
    }
}

// reuse level 3: dir_new = step dir + a_cg r (E_LINE_STEP, E_UPDATE), so A dir_new = step (A dir) + a_cg (A r) without a product
__global__ void k_cg_wd_update(i64 m, const double *__restrict__ scal, const double *__restrict__ w, double *__restrict__ wd) {
    const double t = -(scal[S_T] + scal[S_T + 1]);
    const bool on = fabs(t) > 0.0;
    const double step = t / (scal[S_DMD] + scal[S_DMD + 1]);
    const double acg = (scal[S_RS] + scal[S_RS + 1]) / (scal[S_PAP] + scal[S_PAP + 1]);
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x)
        wd[i] = (on ? step * wd[i] : 0.0) + acg * w[i];
}

// lambda_eq_i += gamma_eq (w_i - b_i)  (:261-263), w = A x
// (rs != NULL: deferred row scaling -- the copy rs o v1 that the A^T product of the next x-step reads is written here too,
// instead of by a pass of its own in front of that product)
__global__ void k_cg_multiplier(i64 m, const double *__restrict__ w, const double *__restrict__ b, double gamma_eq,
                                double *__restrict__ lam, double *__restrict__ v1, const double *__restrict__ rs,
                                double *__restrict__ v1s) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        const double l = lam[i] + gamma_eq * (w[i] - b[i]);
        lam[i] = l;
        if (v1) {  // fused mode: A^T (g_eq A x + lambda) is one product in the next iteration
            const double v = gamma_eq * w[i] + l;
            v1[i] = v;
            if (rs) v1s[i] = rs[i] * v;
        }
    }
}

__global__ void k_cg_v1(i64 m, const double *__restrict__ w, const double *__restrict__ lam, double gamma_eq, double *__restrict__ v1,
                        const double *__restrict__ rs, double *__restrict__ v1s) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        const double v = gamma_eq * w[i] + lam[i];
        v1[i] = v;
        if (rs) v1s[i] = rs[i] * v;
    }
}

__global__ void k_cg_q(i64 N, i64 n_o, const double *__restrict__ c, const double *__restrict__ u, const double *__restrict__ sc,
                       const double *__restrict__ b, double gamma_eq, double *__restrict__ q) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (i64)gridDim.x * blockDim.x)
        q[j] = (-c[j]) + gamma_eq * at_elem(j, n_o, u, sc, b);  // -c + gamma_eq A^T b
}

// report partials over rows: [0] sum r^2  [1] sum lambda r  [2] max |r|   (w = A x)
__global__ __launch_bounds__(kBlock) void k_cg_report_rows(i64 m, const double *__restrict__ w, const double *__restrict__ b,
                                                           const double *__restrict__ lam, double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double s0 = 0.0, s1 = 0.0, mx = -__builtin_inf();
    for (i64 i = (i64)blockIdx.x * kBlock + threadIdx.x; i < m; i += (i64)gridDim.x * kBlock) {
        const double r = w[i] - b[i];
        s0 += r * r;
        s1 += lam[i] * r;
        mx = fabs(r) > mx ? fabs(r) : mx;
    }
    const double r0 = block_reduce<false>(s0, lds), r1 = block_reduce<false>(s1, lds), r2 = block_reduce<true>(mx, lds);
    if (threadIdx.x == 0) { part[blockIdx.x * 3] = r0; part[blockIdx.x * 3 + 1] = r1; part[blockIdx.x * 3 + 2] = r2; }
}

// over unknowns, original / slack parts apart: [0,1] sum c x  [2,3] sum (x-xp)^2  [4,5] sum lin (x-xp)  [6,7] max(-x)
__global__ __launch_bounds__(kBlock) void k_cg_report_cols(i64 N, i64 n_o, const double *__restrict__ c, const double *__restrict__ x,
                                                           const double *__restrict__ xp, const double *__restrict__ lin,
                                                           double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double v[8] = {0, 0, 0, 0, 0, 0, -__builtin_inf(), -__builtin_inf()};
    for (i64 j = (i64)blockIdx.x * kBlock + threadIdx.x; j < N; j += (i64)gridDim.x * kBlock) {
        const int o = j < n_o ? 0 : 1;
        const double dx = x[j] - xp[j];
        v[0 + o] += c[j] * x[j];
        v[2 + o] += dx * dx;
        v[4 + o] += lin[j] * dx;
        v[6 + o] = (-x[j]) > v[6 + o] ? (-x[j]) : v[6 + o];
    }
    for (int k = 0; k < 8; ++k) {
        const double r = k < 6 ? block_reduce<false>(v[k], lds) : block_reduce<true>(v[k], lds);
        if (threadIdx.x == 0) part[blockIdx.x * 8 + k] = r;
    }
}

__global__ __launch_bounds__(kBlock) void k_cg_report_final(int nr, const double *__restrict__ rp, int nc, const double *__restrict__ cp,
                                                            double *__restrict__ out) {
    __shared__ double lds[kBlock / kWave];
    double r[3] = {0, 0, -__builtin_inf()};
    double c[8] = {0, 0, 0, 0, 0, 0, -__builtin_inf(), -__builtin_inf()};
    for (int i = threadIdx.x; i < nr; i += kBlock) {
        r[0] += rp[i * 3];
        r[1] += rp[i * 3 + 1];
        r[2] = rp[i * 3 + 2] > r[2] ? rp[i * 3 + 2] : r[2];
    }
    for (int i = threadIdx.x; i < nc; i += kBlock)
        for (int k = 0; k < 8; ++k) {
            if (k < 6) c[k] += cp[i * 8 + k];
            else c[k] = cp[i * 8 + k] > c[k] ? cp[i * 8 + k] : c[k];
        }
    for (int k = 0; k < 3; ++k) {
        const double v = k < 2 ? block_reduce<false>(r[k], lds) : block_reduce<true>(r[k], lds);
        if (threadIdx.x == 0) out[k] = v;
    }
    for (int k = 0; k < 8; ++k) {
        const double v = k < 6 ? block_reduce<false>(c[k], lds) : block_reduce<true>(c[k], lds);
        if (threadIdx.x == 0) out[3 + k] = v;
    }
}

// ---- device-side setup of the all-inequality standard form (ADMM.py:76-91, tools.py:272-290,88-127)
// pass 1: inv1_i = 1/||a_i||_2 (0 -> 1);  a_ij *= inv1_i ; bu_i *= inv1_i
// pass 2: inv2_i = 1/sqrt(sum_j a'_ij^2 + 1) ; a'_ij *= inv2_i ; sc_i = -1 * inv2_i   (the slack entry of [A' -I])
// Rows i < m_eq are equalities [A_eq' 0] (tools.py:96-107): no slack entry (sc_i = 0, no "+ 1"), and their
// right-hand side b_eq is scaled in both passes; inequality rows end with right-hand side 0 and the scaled
// upper bound bu_i on their slack variable.
template <int L>
__global__ __launch_bounds__(kBlock) void k_cg_scale_rows(i64 m, i64 m_eq, const i64 *__restrict__ ptr, double *__restrict__ val, int pass,
                                                          double *__restrict__ bu, double *__restrict__ sc, double *__restrict__ bl) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 i = group; i < m; i += ngroups) {
        const i64 s = ptr[i], e = ptr[i + 1];
        double acc = 0.0;
        for (i64 k = s + sub; k < e; k += L) acc += (val[k] * val[k]) * 1.0;
        acc = group_sum<L>(acc);
        if (pass == 2 && i >= m_eq) acc = acc + 1.0;
        double nrm = sqrt(acc);
        if (nrm == 0.0) nrm = 1.0;
        const double inv = 1.0 / nrm;
        for (i64 k = s + sub; k < e; k += L) val[k] = inv * val[k];
        if (sub == 0) {
            if (pass == 1) { bu[i] = inv * bu[i]; if (bl && i >= m_eq) bl[i] = inv * bl[i]; }  // b_lower is scaled like b_upper (tools.py:286-288)
            else if (i >= m_eq) sc[i] = inv * -1.0;
            else { sc[i] = 0.0; bu[i] = inv * bu[i]; }
        }
    }
}

// The same two passes WITHOUT touching the matrix: rs_i = inv2_i * inv1_i is kept as a row-scale vector and
// applied around the products (A v = rs o (A0 v), A^T w = A0^T (rs o w)).  Used when the unscaled matrix has few
// distinct values and runs on the value-dictionary strips (slp_strip.hip): scaling the entries would
// make every row's values unique.  Differs from the in-place form by the rounding of rs_i * sum versus
// sum of (rs_i * a_ij) terms only.
// In two steps: the sums over a row's entries (k_cg_row_sq: all that is ever needed of the CSR -- a chunked matrix takes
// them chunk by chunk while a chunk's CSR exists, slp_chunked.hip), then an elementwise pass (k_cg_row_scales_from).
template <int L>
__global__ __launch_bounds__(kBlock) void k_cg_row_sq(i64 m, const i64 *__restrict__ ptr, const double *__restrict__ val,
                                                      double *__restrict__ sq) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 i = group; i < m; i += ngroups) {
        const i64 s = ptr[i], e = ptr[i + 1];
        double acc = 0.0;
        for (i64 k = s + sub; k < e; k += L) acc += (val[k] * val[k]) * 1.0;
        acc = group_sum<L>(acc);
        const double sq1 = acc;
        double nrm = sqrt(acc);
        if (nrm == 0.0) nrm = 1.0;
        const double inv1 = 1.0 / nrm;
        acc = 0.0;
        for (i64 k = s + sub; k < e; k += L) {
            const double v = inv1 * val[k];
            acc += (v * v) * 1.0;
        }
        acc = group_sum<L>(acc);
        if (sub == 0) { sq[2 * i] = sq1; sq[2 * i + 1] = acc; }
    }
}

__global__ void k_cg_row_scales_from(i64 m, i64 m_eq, const double *__restrict__ sq, double *__restrict__ bu, double *__restrict__ sc,
                                     double *__restrict__ rs, double *__restrict__ bl) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        double nrm = sqrt(sq[2 * i]);
        if (nrm == 0.0) nrm = 1.0;
        const double inv1 = 1.0 / nrm;
        double acc = sq[2 * i + 1];
        if (i >= m_eq) acc = acc + 1.0;
        nrm = sqrt(acc);
        if (nrm == 0.0) nrm = 1.0;
        const double inv2 = 1.0 / nrm;
        rs[i] = inv2 * inv1;
        if (i >= m_eq) { sc[i] = inv2 * -1.0; bu[i] = inv1 * bu[i]; if (bl) bl[i] = inv1 * bl[i]; }
        else { sc[i] = 0.0; bu[i] = inv2 * (inv1 * bu[i]); }
    }
}

// after the two scaling passes: rhs_i = b_eq'' (equalities) or 0; slack bounds [0,0] (equalities) or [bl' (default -inf), bu']
__global__ void k_cg_split_rows(i64 m, i64 m_eq, const double *__restrict__ bu, const double *__restrict__ bl, double *__restrict__ rhs,
                                double *__restrict__ slo, double *__restrict__ shi) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        const bool eq = i < m_eq;
        rhs[i] = eq ? bu[i] : 0.0;
        slo[i] = eq ? 0.0 : (bl ? bl[i] : -__builtin_inf());
        shi[i] = eq ? 0.0 : bu[i];
    }
}

__global__ void k_fill(i64 n, double *__restrict__ p, double v) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) p[j] = v;
}

}  // namespace slp

using namespace slp;

struct slp_admm_cg {
    slp_matrix *a = nullptr;
    bool owns_a = false;
    i64 n_o = 0, m = 0, ns = 0, N = 0;
    double gamma_eq = 2, gamma_ineq = 3, alpha = 1.4;
    int order = SLP_ORDER_AUTO, lanes_rows = 1, lanes_cols = 1;
    bool distributed = false;
    int since_refresh = 0; // level 3: iterations since A dir was last taken as a product
    bool need_md = true;   // level 4: M dir must come from a product in the next x-step (start, and after every refresh)
    int reuse = 0;        // 0: ten products as written; 1: CG residual from the line search's products (8);
                          // 3: additionally A dir by recurrence from A dir_old and A r (5 products; exact refresh every 64 iterations)
                          // 2: additionally A^T (g_eq A x + lambda_eq) as one product (6 products, 4 passes with strips)
    DevBuf<double> sc, b, lam, w;                                         // rows
    DevBuf<double> rs, ws0, ws1;   // deferred row scaling (value-dictionary strips): A = diag(rs) A0; scratch rs o w
    DevBuf<double> wsw, wsv1;      // ... rs o w and rs o v1, written by the kernels that produce w and v1
    DevBuf<double> c, lb, ub, x, xp, y, q, dir, xprev, r, lin, u, mx, md;        // unknowns (u: n_o)
    DevBuf<double> part, rowpart, colpart, scal, out;
    DevBuf<double> wx, wd, u2, v1;   // batched form: A x, A dir, g_eq A x + lambda (rows) and the two A^T products (2 n_o)
    bool have_w = false;
    // sharded updates (cg_xstep_sharded): this rank's slice [o0, o1) of the original variables; slice_on: the elementwise
    // launches walk the slice only; shard_dirty: the slice-only vectors (dir, md, g, xprev, xp, lin -- and x between the two
    // halves of an iteration) are current on their owner only (cg_gather_state makes them whole again)
    bool sharded = false, slice_on = false, shard_dirty = false;
    i64 o0 = 0, o1 = 0, scnt = 0;
    DevBuf<double> ex;
    bool started = false;   // at least one full iteration done: the steady-state kernel sequence can be replayed
    IterGraph graph;        // launch-bound problems: iterations replayed as a captured graph
};

namespace slp {

// w = rs o w (deferred row scale, optional) + sc o vs (the slack column, optional)
// (wsc != NULL: also rs o w of the finished w, the vector the A^T product that follows reads)
__global__ void k_cg_add_slack(i64 m, const double *__restrict__ rs, const double *__restrict__ sc, const double *__restrict__ vs,
                               double *__restrict__ w, double *__restrict__ wsc) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        double v = w[i];
        if (rs) v = rs[i] * v;
        if (sc) v = v + sc[i] * vs[i];
        w[i] = v;
        if (wsc) wsc[i] = rs[i] * v;
    }
}

__global__ void k_cg_row_scaled(i64 m, const double *__restrict__ rs, const double *__restrict__ w, double *__restrict__ out) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) out[i] = rs[i] * w[i];
}

void matrix_row_squares(const CsrDev &a, double *sq) {
    if (a.nrow == 0) return;
    const int lanes = lanes_for(a, SLP_ORDER_TREE);
    SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cg_row_sq<L>), dim3(grid_for(a.nrow * lanes, kBlock)), dim3(kBlock), 0, ctx().stream,
                                                 a.nrow, a.ptr.p, a.val.p, sq));
    SLP_HIP(hipGetLastError());
}

static void cg_finish_rows(slp_admm_cg *s, const double *v, double *w) {
    if (!s->ns && !s->rs.p) return;
    // the solver's own w = A v is what cg_cols multiplies next: its row-scaled copy is written in the same pass
    double *wsc = (s->rs.p && w == s->w.p) ? s->wsw.p : nullptr;
    hipLaunchKernelGGL(k_cg_add_slack, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, ctx().stream, s->m, s->rs.p,
                       s->ns ? s->sc.p : (double *)nullptr, v + s->n_o, w, wsc);
    SLP_HIP(hipGetLastError());
}

// rs o w in scratch (deferred row scaling) or w itself
static const double *cg_scaled_rows(slp_admm_cg *s, const double *w, DevBuf<double> &tmp) {
    if (!s->rs.p) return w;
    if (w == s->w.p) return s->wsw.p;                 // written by cg_finish_rows together with w
    if (w == s->v1.p && s->v1.p) return s->wsv1.p;    // written by k_cg_multiplier / k_cg_v1 together with v1
    if (tmp.n < (size_t)s->m) tmp.alloc((size_t)s->m);
    hipLaunchKernelGGL(k_cg_row_scaled, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, ctx().stream, s->m, s->rs.p, w, tmp.p);
    SLP_HIP(hipGetLastError());
    return tmp.p;
}

// w_out = A v   (v over the N unknowns: original part, then this rank's slack part)
static void cg_rows(slp_admm_cg *s, const double *v, double *w_out = nullptr) {
    if (s->m == 0) return;
    double *w = w_out ? w_out : s->w.p;
    const CsrDev &a = s->a->a;
    if (const StripJds *f = fast_format(s->a, false)) {  // long rows: LDS-tiled product, then the slack column
        strip_spmv(*f, v, w);
        cg_finish_rows(s, v, w);
        return;
    }
    SLP_REQUIRE(!s->rs.p, "deferred row scaling needs the strip format in both orientations");
    require_csr(s->a, "matrix-free ADMM row product (CSR walk)");
    const int lanes = s->lanes_rows;
    SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cg_rows<L>), dim3(grid_for(s->m * lanes, kBlock)), dim3(kBlock), 0, ctx().stream,
                                                 s->m, a.ptr.p, a.idx.p, a.val.p, v, s->ns ? s->sc.p : nullptr, s->n_o, w));
    SLP_HIP(hipGetLastError());
}

// [w0, w1] = A [v0, v1]: ONE pass over the matrix when the strip format is available
static void cg_rows2(slp_admm_cg *s, const double *v0, const double *v1, double *w0, double *w1) {
    if (s->m == 0) return;
    if (const StripJds *f = fast_format(s->a, false)) {
        strip_spmv2(*f, v0, v1, w0, w1);
        cg_finish_rows(s, v0, w0);
        cg_finish_rows(s, v1, w1);
        return;
    }
    cg_rows(s, v0, w0);
    cg_rows(s, v1, w1);
}

// u_out = (A^T w) restricted to the original variables, summed over the ranks; `tail` more doubles stored right behind the
// n_o sums (packed exchange: rank-local slack parts of dot products) are summed in the same all-reduce
static void cg_cols(slp_admm_cg *s, const double *w, double *u_out = nullptr, bool reduce = true, i64 tail = 0) {
    if (s->n_o == 0) return;
    double *u = u_out ? u_out : s->u.p;
    const CsrDev &at = s->a->at;
    if (const StripJds *f = fast_format(s->a, true)) {
        strip_spmv(*f, cg_scaled_rows(s, w, s->ws0), u);
    } else {
        SLP_REQUIRE(!s->rs.p, "deferred row scaling needs the strip format in both orientations");
        require_csr(s->a, "matrix-free ADMM column product (CSR walk)");
        const int lanes = s->lanes_cols;
        SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cg_cols<L>), dim3(grid_for(s->n_o * lanes, kBlock)), dim3(kBlock), 0,
                                                     ctx().stream, s->n_o, at.ptr.p, at.idx.p, at.val.p, w, u));
        SLP_HIP(hipGetLastError());
    }
    if (s->distributed && reduce) comm_allreduce_dev(u, s->n_o + tail, 0);
}

// [u2[0..n_o), u2[n_o..2 n_o)] = A^T [w0, w1]: one pass, one all-reduce of 2 n_o values
static void cg_cols2(slp_admm_cg *s, const double *w0, const double *w1, double *u2, i64 tail = 0) {
    if (s->n_o == 0) return;
    // the same ONE collective on every rank, whichever kernel a rank's block runs on (ranks must never disagree on the
    // sequence of all-reduces)
    if (const StripJds *f = fast_format(s->a, true)) {
        strip_spmv2(*f, cg_scaled_rows(s, w0, s->ws0), cg_scaled_rows(s, w1, s->ws1), u2, u2 + s->n_o);
    } else {
        cg_cols(s, w0, u2, false);
        cg_cols(s, w1, u2 + s->n_o, false);
    }
    if (s->distributed) comm_allreduce_dev(u2, 2 * s->n_o + tail, 0);
}

static CgVecs cg_vecs(slp_admm_cg *s, const double *u, const double *w) {
    CgVecs v;
    v.q = s->q.p; v.c = s->c.p; v.lb = s->lb.p; v.ub = s->ub.p; v.sc = s->sc.p;
    v.u = u ? u : s->u.p;   // (A^T .) over the original variables ...
    v.w = w ? w : s->w.p;   // ... and the row vector whose slack column gives the rest: sc_i * w_i
    v.x = s->x.p; v.xp = s->xp.p; v.y = s->y.p; v.dir = s->dir.p; v.xprev = s->xprev.p; v.r = s->r.p; v.lin = s->lin.p;
    v.mx = s->mx.p; v.md = s->md.p; v.keep = s->reuse ? 1 : 0; v.carry_md = s->reuse >= 4 ? 1 : 0;
    v.scal = s->scal.p; v.n_o = s->n_o; v.N = s->N;
    v.o0 = s->slice_on ? s->o0 : 0; v.o1 = s->slice_on ? s->o1 : s->n_o;
    v.gamma_eq = s->gamma_eq; v.gamma_ineq = s->gamma_ineq; v.alpha = s->alpha; v.one_minus_alpha = 1.0 - s->alpha;
    return v;
}

// grid of the elementwise passes: `go` blocks for the replicated original variables (a function of n_o alone: the same on
// every rank), the rest for this rank's slack variables
static void cg_grids(const slp_admm_cg *s, int *go, int *gs) {
    *go = std::min(grid_for(s->n_o, kBlock), kCgPartials / 2);
    *gs = std::min(grid_for(s->N - s->n_o, kBlock), kCgPartials / 2);
}

// tail / tail_mode: packed exchange -- the slack part of the dot product was reduced over the ranks behind a vector
// all-reduce (k_cg_finish); otherwise, when the rows are partitioned, one scalar all-reduce of the slack part.
template <int OP>
static void cg_elem(slp_admm_cg *s, int slot, const double *u = nullptr, const double *w = nullptr, const double *tail = nullptr,
                    int tail_mode = 0) {
    int go, gs;
    cg_grids(s, &go, &gs);
    hipLaunchKernelGGL((k_cg_elem<OP>), dim3(go + gs), dim3(kBlock), 0, ctx().stream, cg_vecs(s, u, w), s->part.p, go);
    SLP_HIP(hipGetLastError());
    if (slot >= 0) {
        hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(kBlock), 0, ctx().stream, go, go + gs, s->part.p, s->scal.p, slot, tail, tail_mode);
        SLP_HIP(hipGetLastError());
        if (s->distributed && !tail_mode) comm_allreduce_dev(s->scal.p + slot + 1, 1, 0);  // slack part lives on its rank
    }
}

// E_GRAD and E_LINE_DMD_MD in one pass, both dot products finished by one launch
static void cg_elem2_grad_dmd(slp_admm_cg *s, const double *u, const double *w, const double *tail, int tail_mode) {
    int go, gs;
    cg_grids(s, &go, &gs);
    hipLaunchKernelGGL((k_cg_elem<E_GRAD_DMD>), dim3(go + gs), dim3(kBlock), 0, ctx().stream, cg_vecs(s, u, w), s->part.p, go);
    hipLaunchKernelGGL(k_cg_finish2, dim3(1), dim3(kBlock), 0, ctx().stream, go, go + gs, s->part.p, s->scal.p, tail, tail_mode);
    SLP_HIP(hipGetLastError());
    if (s->distributed && !tail_mode) {  // (not reached at level 4: the packed exchange carries the slack parts)
        comm_allreduce_dev(s->scal.p + S_T + 1, 1, 0);
        comm_allreduce_dev(s->scal.p + S_DMD + 1, 1, 0);
    }
}

// Packed exchange: the rank-local slack sums land in `tail` (device, right behind the vector the next all-reduce sends)
static void cg_slack_pre(slp_admm_cg *s, bool refresh, double *tail) {
    const int g = std::min(grid_for(s->N - s->n_o, kBlock), kCgPartials / 8);
    hipLaunchKernelGGL(k_cg_slack_pre, dim3(g), dim3(kBlock), 0, ctx().stream, cg_vecs(s, nullptr, nullptr), s->v1.p, s->wd.p, refresh ? 1 : 0,
                       s->part.p);
    hipLaunchKernelGGL(k_cg_tail, dim3(1), dim3(kBlock), 0, ctx().stream, g, 5, s->part.p, tail);
    SLP_HIP(hipGetLastError());
}

static void cg_slack_pap(slp_admm_cg *s, const double *w, double *tail) {
    const int g = std::min(grid_for(s->N - s->n_o, kBlock), kCgPartials / 8);
    hipLaunchKernelGGL(k_cg_slack_pap, dim3(g), dim3(kBlock), 0, ctx().stream, cg_vecs(s, nullptr, nullptr), w, s->part.p);
    hipLaunchKernelGGL(k_cg_tail, dim3(1), dim3(kBlock), 0, ctx().stream, g, 1, s->part.p, tail);
    SLP_HIP(hipGetLastError());
}

// Batched form (reuse mode + strip kernels): the products A x and A dir that the NEXT line search needs are
// taken in one two-vector pass at the end of an iteration (the multiplier update needs A x anyway), and
// A^T (A x), A^T (A dir) in one two-vector pass: 5 passes over the matrix per iteration instead of 8.
// Every product is the same arithmetic as in the unbatched form; only the passes are shared.
static bool cg_batched(slp_admm_cg *s) {
    if (!s->reuse) return false;
    // rows partitioned: never let the sequence (or the size) of the collectives depend on rank-local data -- a rank whose
    // row block is empty (m == 0: fewer rows than ranks) runs the same batched sequence over zero-sized products
    if (s->distributed) return true;
    if (s->m <= 0 || s->n_o <= 0) return false;
    if (s->reuse >= 2) return true;  // the fused form is written on top of the shared products
    return fast_format(s->a, false) && fast_format(s->a, true);
}

static void cg_refresh_products(slp_admm_cg *s) {  // wx = A x, wd = A dir
    if (!s->u2.p) {  // (+ kCgTail: the packed exchange appends rank-local scalars to the vectors it sends)
        s->wx.alloc((size_t)s->m); s->wd.alloc((size_t)s->m); s->v1.alloc((size_t)s->m); s->u2.alloc(2 * (size_t)s->n_o + kCgTail);
    }
    cg_rows2(s, s->x.p, s->dir.p, s->wx.p, s->wd.p);
    s->have_w = true;
}

// Every slice-only vector whole again on every rank (before anything that reads them outside the sharded steady state: the
// refresh iteration, a report, a change of mode).  Collective: every rank calls it at the same points.
static void cg_gather_state(slp_admm_cg *s) {
    if (!s->sharded || !s->shard_dirty) return;
    for (double *v : {s->x.p, s->dir.p, s->md.p, s->mx.p, s->xprev.p, s->xp.p, s->lin.p}) comm_all_gather_dev(v, s->scnt);
    s->shard_dirty = false;
}

template <int OP>
static void cg_elem_launch(slp_admm_cg *s, const double *u, const double *w) {
    int go, gs;
    cg_grids(s, &go, &gs);
    hipLaunchKernelGGL((k_cg_elem<OP>), dim3(go + gs), dim3(kBlock), 0, ctx().stream, cg_vecs(s, u, w), s->part.p, go);
    SLP_HIP(hipGetLastError());
}

static void cg_ex_finish(slp_admm_cg *s, int i0, int two) {
    int go, gs;
    cg_grids(s, &go, &gs);
    hipLaunchKernelGGL(k_cg_ex_finish, dim3(1), dim3(kBlock), 0, ctx().stream, go, s->part.p, s->ex.p, i0, two);
    SLP_HIP(hipGetLastError());
}

// The level-4 steady-state x-step with the replicated work SHARDED: the two all-reduces of the variable
// vector become reduce-scatter ... all-gather pairs (the same bytes over the links) and the elementwise passes between them
// run over n / N variables instead of n; the slice-partial dot products travel in two small all-reduces.  Six collectives
// per iteration instead of two.  Replicas stay bit-identical by construction: a slice is computed by its owner alone and
// gathered; every scalar is the result of an all-reduce.
static void cg_xstep_sharded(slp_admm_cg *s) {
    hipStream_t st = ctx().stream;
    double *ex = s->ex.p;
    cg_slack_pre(s, false, ex + 2);                                   // five rank-local slack sums
    cg_cols(s, s->v1.p, s->u2.p, false);                              // this rank's partial A^T v1 ...
    comm_reduce_scatter_dev(s->u2.p, s->scnt);                        // ... summed, my slice of it
    s->slice_on = true;
    cg_elem_launch<E_GRAD_DMD>(s, s->u2.p, s->v1.p);                  // g, xprev, and the slice's dir.g / dir.Md
    cg_ex_finish(s, 0, 1);
    comm_allreduce_dev(ex, 7, 0);
    hipLaunchKernelGGL(k_cg_scal_from_ex, dim3(1), dim3(64), 0, st, ex, s->scal.p, 0);
    cg_elem_launch<E_STEP_RESID>(s, nullptr, nullptr);                // x += step dir ; r = -(g + step M dir) ; the slice's r.r
    cg_ex_finish(s, 8, 0);
    s->slice_on = false;
    comm_all_gather_dev(s->r.p, s->scnt);                             // A r needs all of r
    cg_rows(s, s->r.p);
    cg_slack_pap(s, s->w.p, ex + 10);
    cg_cols(s, s->w.p, nullptr, false);
    comm_reduce_scatter_dev(s->u.p, s->scnt);
    s->slice_on = true;
    cg_elem_launch<E_PAP>(s, nullptr, nullptr);                       // M r on the slice, the slice's r.Mr
    cg_ex_finish(s, 9, 0);
    comm_allreduce_dev(ex + 8, 3, 0);
    hipLaunchKernelGGL(k_cg_scal_from_ex, dim3(1), dim3(64), 0, st, ex, s->scal.p, 1);
    cg_elem_launch<E_UPDATE>(s, nullptr, nullptr);
    s->slice_on = false;
    hipLaunchKernelGGL(k_cg_wd_update, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, s->scal.p, s->w.p, s->wd.p);
    SLP_HIP(hipGetLastError());
    s->have_w = false;
    s->shard_dirty = true;
}

// first half of an iteration: everything up to (and including) the over-relaxed x (:148-201)
static void cg_xstep(slp_admm_cg *s) {
    if (s->sharded && s->reuse >= 4 && !s->need_md && s->have_w && cg_batched(s)) {
        cg_xstep_sharded(s);
        return;
    }
    cg_gather_state(s);
    if (s->reuse >= 2 && cg_batched(s)) {
        // fused: [A^T (g_eq A x + lambda), A^T (A dir)] in one two-vector pass; y is never formed
        if (!s->have_w) {
            cg_refresh_products(s);
            hipLaunchKernelGGL(k_cg_v1, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, ctx().stream, s->m, s->wx.p, s->lam.p,
                               s->gamma_eq, s->v1.p, s->rs.p, s->wsv1.p);
            SLP_HIP(hipGetLastError());
        }
        // Rows partitioned, level 4: the slack parts of ALL dot products of the iteration ride behind the two vector
        // all-reduces (5 doubles behind A^T v1, 1 behind A^T (A r)): two collectives per iteration, no scalar ones.
        const bool packed = s->distributed && s->reuse >= 4;
        if (s->reuse >= 4 && !s->need_md) {
            // level 4: M dir = step M dir_old + a_cg M r was advanced in E_UPDATE: the A^T pass carries one vector
            double *tail = s->u2.p + s->n_o;
            if (packed) cg_slack_pre(s, false, tail);
            cg_cols(s, s->v1.p, s->u2.p, true, packed ? 5 : 0);
            cg_elem2_grad_dmd(s, s->u2.p, s->v1.p, tail, packed ? 1 : 0);
            cg_elem<E_STEP_RESID>(s, S_RS, nullptr, nullptr, tail + 2, packed ? 2 : 0);
        } else {
            double *tail = s->u2.p + 2 * s->n_o;
            if (packed) cg_slack_pre(s, true, tail);
            cg_cols2(s, s->v1.p, s->wd.p, s->u2.p, packed ? 5 : 0);
            cg_elem<E_GRAD>(s, S_T, s->u2.p, s->v1.p, tail, packed ? 1 : 0);
            cg_elem<E_LINE_DMD>(s, S_DMD, s->u2.p + s->n_o, s->wd.p, tail + 1, packed ? 1 : 0);
            s->need_md = false;
            cg_elem<E_STEP_RESID>(s, S_RS, nullptr, nullptr, tail + 2, packed ? 2 : 0);
        }
        cg_rows(s, s->r.p);
        if (packed) cg_slack_pap(s, s->w.p, s->u.p + s->n_o);
        cg_cols(s, s->w.p, nullptr, true, packed ? 1 : 0);
        cg_elem<E_PAP>(s, S_PAP, nullptr, nullptr, s->u.p + s->n_o, packed ? 1 : 0);
        cg_elem<E_UPDATE>(s, -1);
        if (s->reuse >= 3) {  // w still holds A r
            hipLaunchKernelGGL(k_cg_wd_update, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, ctx().stream, s->m, s->scal.p, s->w.p, s->wd.p);
            SLP_HIP(hipGetLastError());
        }
        s->have_w = false;
        return;
    }
    cg_cols(s, s->lam.p);                                   // A^T lambda_eq over the original variables
    cg_elem<E_RHS>(s, -1, s->u.p, s->lam.p);                // slack part of A^T lambda: sc_i * lambda_i
    if (cg_batched(s)) {
        if (!s->have_w) cg_refresh_products(s);
        cg_cols2(s, s->wx.p, s->wd.p, s->u2.p);
        cg_elem<E_LINE_T>(s, S_T, s->u2.p, s->wx.p);
        cg_elem<E_LINE_DMD>(s, S_DMD, s->u2.p + s->n_o, s->wd.p);
        cg_elem<E_LINE_STEP>(s, -1);
        cg_elem<E_RESID_REUSE>(s, S_RS);
    } else {
        cg_rows(s, s->x.p); cg_cols(s, s->w.p); cg_elem<E_LINE_T>(s, S_T);
        cg_rows(s, s->dir.p); cg_cols(s, s->w.p); cg_elem<E_LINE_DMD>(s, S_DMD);
        cg_elem<E_LINE_STEP>(s, -1);
        if (s->reuse) {
            cg_elem<E_RESID_REUSE>(s, S_RS);
        } else {
            cg_rows(s, s->x.p); cg_cols(s, s->w.p); cg_elem<E_RESID>(s, S_RS);
        }
    }
    cg_rows(s, s->r.p); cg_cols(s, s->w.p); cg_elem<E_PAP>(s, S_PAP);
    cg_elem<E_UPDATE>(s, -1);
    s->have_w = false;  // x and dir changed
}

// second half: projection step and both multiplier updates (:253-263)
static void cg_multipliers(slp_admm_cg *s) {
    if (s->sharded && s->shard_dirty) {     // the projection on my slice, then x whole again for A x
        s->slice_on = true;
        cg_elem_launch<E_PROJECT>(s, nullptr, nullptr);
        s->slice_on = false;
        comm_all_gather_dev(s->x.p, s->scnt);
        if (!(s->reuse >= 3 && cg_batched(s) && s->u2.p && s->since_refresh + 1 < 64)) cg_gather_state(s);  // the refresh walks whole vectors
    } else {
        cg_elem<E_PROJECT>(s, -1);
    }
    const double *ax = s->w.p;
    if (s->reuse >= 3 && cg_batched(s) && s->u2.p && ++s->since_refresh < 64) {
        cg_rows(s, s->x.p, s->wx.p);  // A dir came from the recurrence (k_cg_wd_update): one product, one vector
        s->have_w = true;
        ax = s->wx.p;
    } else if (cg_batched(s)) {
        cg_refresh_products(s);   // A x for the multiplier AND, with A dir, for the next line search
        s->since_refresh = 0;
        s->need_md = true;
        ax = s->wx.p;
    } else {
        cg_rows(s, s->x.p);
    }
    if (s->m) {
        hipLaunchKernelGGL(k_cg_multiplier, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, ctx().stream, s->m, ax, s->b.p,
                           s->gamma_eq, s->lam.p, (s->reuse >= 2 && ax == s->wx.p) ? s->v1.p : (double *)nullptr, s->rs.p, s->wsv1.p);
        SLP_HIP(hipGetLastError());
    }
}

static void cg_alloc_state(slp_admm_cg *s) {
    const size_t N = (size_t)s->N, m = (size_t)s->m;
    s->lam.alloc(m); s->lam.zero();
    s->w.alloc(m);
    s->xp.alloc(N); s->y.alloc(N); s->q.alloc(N); s->dir.alloc(N); s->dir.zero(); s->xprev.alloc(N); s->r.alloc(N);
    s->lin.alloc(N); s->lin.zero();
    s->u.alloc((size_t)s->n_o + kCgTail);
    s->part.alloc((size_t)kCgPartials * 2); s->rowpart.alloc((size_t)kCgPartials * 3); s->colpart.alloc((size_t)kCgPartials * 8);
    s->scal.alloc(S_COUNT); s->scal.zero(); s->out.alloc(16);
    s->distributed = comm_active();
    {
        const char *es = getenv("SLP_SHARD_UPDATES");
        const int nr = comm_size();
        s->sharded = s->distributed && es && es[0] == '1' && nr > 1 && s->ns > 0 && s->n_o > 0 && s->n_o % nr == 0;
        if (s->sharded) {
            s->scnt = s->n_o / nr;
            s->o0 = (i64)comm_rank() * s->scnt;
            s->o1 = s->o0 + s->scnt;
            s->ex.alloc(16);
            s->ex.zero();
        }
    }
    s->lanes_rows = lanes_for(s->a->a, s->order);
    s->lanes_cols = lanes_for(s->a->at, s->order);
    hipStream_t st = ctx().stream;
    // q = -c + gamma_eq A^T b (:95,:148)
    cg_cols(s, s->b.p);
    if (s->N) {
        hipLaunchKernelGGL(k_cg_q, dim3(grid_for(s->N, kBlock)), dim3(kBlock), 0, st, s->N, s->n_o, s->c.p, s->u.p, s->sc.p, s->b.p,
                           s->gamma_eq, s->q.p);
        SLP_HIP(hipGetLastError());
    }
}

__global__ void k_cg_max0(i64 n, const double *__restrict__ x, double *__restrict__ xp) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x)
        xp[j] = (x[j] < 0.0) ? 0.0 : x[j];
}

}  // namespace slp

extern "C" {

slp_admm_cg *slp_admm_cg_create(int64_t N, int64_t m, const int64_t *indptr, const int32_t *indices, const double *data,
                                const double *b, const double *c, const double *lb, const double *ub, const double *x0,
                                double gamma_eq, double gamma_ineq, int order) {
    SLP_API_PTR({
        SLP_REQUIRE(indptr && b && c && lb && ub && x0, "slp_admm_cg_create: NULL argument");
        auto *s = new slp_admm_cg();
        try {
            s->a = slp_matrix_create(m, N, indptr, indices, data);
            if (!s->a) throw Error(slp_last_error());
            s->owns_a = true;
            ensure_transposed(s->a);
            s->n_o = N; s->m = m; s->ns = 0; s->N = N;
            s->gamma_eq = gamma_eq; s->gamma_ineq = gamma_ineq; s->order = order;
            s->b.upload(b, (size_t)m); s->c.upload(c, (size_t)N); s->lb.upload(lb, (size_t)N); s->ub.upload(ub, (size_t)N);
            s->x.upload(x0, (size_t)N);
            cg_alloc_state(s);
            if (N) hipLaunchKernelGGL(k_cg_max0, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, ctx().stream, N, s->x.p, s->xp.p);
            SLP_HIP(hipGetLastError());
            SLP_HIP(hipStreamSynchronize(ctx().stream));
        } catch (...) {
            if (s->owns_a) delete s->a;
            delete s;
            throw;
        }
        return s;
    })
}

slp_admm_cg *slp_admm_cg_create_on(slp_matrix *a_ineq, const double *b_upper, const double *c, const double *lb,
                                   const double *ub, double gamma_eq, double gamma_ineq, int order) {
    return slp_admm_cg_create_on_mixed(a_ineq, 0, b_upper, c, lb, ub, gamma_eq, gamma_ineq, order);
}

slp_admm_cg *slp_admm_cg_create_on_mixed(slp_matrix *a_ineq, int64_t m_eq, const double *b_upper, const double *c, const double *lb,
                                         const double *ub, double gamma_eq, double gamma_ineq, int order) {
    return slp_admm_cg_create_on_two_sided(a_ineq, m_eq, nullptr, b_upper, c, lb, ub, gamma_eq, gamma_ineq, order);
}

slp_admm_cg *slp_admm_cg_create_on_two_sided(slp_matrix *a_ineq, int64_t m_eq, const double *b_lower, const double *b_upper,
                                             const double *c, const double *lb, const double *ub, double gamma_eq, double gamma_ineq,
                                             int order) {
    SLP_API_PTR({
        SLP_REQUIRE(a_ineq && b_upper && c && lb && ub, "slp_admm_cg_create_on: NULL argument");
        SLP_REQUIRE(m_eq >= 0 && m_eq <= a_ineq->a.nrow, "slp_admm_cg_create_on_mixed: m_eq out of range");
        const bool chunked = !a_ineq->chunks.empty();
        if (!chunked) require_csr(a_ineq, "slp_admm_cg_create_on");  // the row norms are taken over the CSR entries
        Phase ph("slp_admm_cg_create_on");
        auto *s = new slp_admm_cg();
        try {
            hipStream_t st = ctx().stream;
            s->a = a_ineq;
            s->owns_a = false;
            CsrDev &a = a_ineq->a;
            const i64 m = a.nrow, n = a.ncol;
            s->n_o = n; s->m = m; s->ns = m; s->N = n + m;
            s->gamma_eq = gamma_eq; s->gamma_ineq = gamma_ineq; s->order = order;
            DevBuf<double> bu((size_t)m);
            bu.upload(b_upper, (size_t)m);
            DevBuf<double> bl;  // optional lower bounds of the inequality rows (entries of equality rows are ignored)
            if (b_lower) bl.upload(b_lower, (size_t)m);
            s->sc.alloc((size_t)m);
            // Few distinct stored values and long rows: keep the matrix as it is (value-dictionary strips) and carry
            // the two row scalings as a vector.  Otherwise: rows scaled in place, twice; the transposed copy is
            // (re)built from the scaled values.
            // the in-place normalisation below is not idempotent (pass 2 adds the slack entry's 1)
            SLP_REQUIRE(!a_ineq->scaled, "slp_admm_cg_create_on: this matrix was already row-normalised in place by an earlier ADMM "
                                         "setup; scaling it again would solve a different problem -- build the solver on a fresh matrix");
            bool deferred = false;
            if (chunked) {
                // a chunked matrix holds no CSR: only the deferred form exists, from the row sums its chunks kept
                const StripJds *f0 = fast_format(a_ineq, false), *f1 = fast_format(a_ineq, true);
                deferred = f0->D > 0 && f1->D > 0;
                for (const slp_matrix *ch : a_ineq->chunks) deferred = deferred && ch->rowsq.n == 2 * (size_t)ch->a.nrow;
                SLP_REQUIRE(deferred, "slp_admm_cg_create_on: a chunked matrix runs the matrix-free ADMM on value-dictionary copies only "
                                      "(its rows cannot be scaled in place: no CSR is held)");
            } else if (m && n && (strip_wanted(a, 2) || strip_wanted(a, 1) || strip_wanted(a, 3) || tall_wanted(a.nrow, a.ncol, a.nnz)) &&
                       matrix_dictionary(a_ineq)) {
                const StripJds *f0 = fast_format(a_ineq, false), *f1 = fast_format(a_ineq, true);
                deferred = f0 && f1 && f0->D > 0 && f1->D > 0;
            }
            if (deferred) {
                s->rs.alloc((size_t)m);
                s->wsw.alloc((size_t)m);
                s->wsv1.alloc((size_t)m);
                DevBuf<double> sq(2 * (size_t)m);
                if (chunked) {
                    for (size_t k = 0; k < a_ineq->chunks.size(); ++k) {
                        const slp_matrix *ch = a_ineq->chunks[k];
                        SLP_HIP(hipMemcpyAsync(sq.p + 2 * a_ineq->chunk_row0[k], ch->rowsq.p, 2 * (size_t)ch->a.nrow * sizeof(double),
                                               hipMemcpyDeviceToDevice, st));
                    }
                } else {
                    matrix_row_squares(a, sq.p);
                }
                hipLaunchKernelGGL(k_cg_row_scales_from, dim3(grid_for(m, kBlock)), dim3(kBlock), 0, st, m, (i64)m_eq, sq.p, bu.p, s->sc.p,
                                   s->rs.p, bl.p);
                SLP_HIP(hipGetLastError());
                SLP_HIP(hipStreamSynchronize(st));
            } else if (m) {
                // in place: the matrix then holds the row-normalised values and every derived copy is rebuilt -- refuse when
                // that would pull the data from under another solver
                SLP_REQUIRE(a_ineq->borrowers == 0, "slp_admm_cg_create_on: another solver created on this matrix is still alive; the "
                                                    "in-place row normalisation would change the values it iterates on");
                invalidate_derived(a_ineq);
                a_ineq->scaled = true;
                const int lanes = lanes_for(a, SLP_ORDER_TREE);
                for (int pass = 1; pass <= 2; ++pass) {
                    SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cg_scale_rows<L>), dim3(grid_for(m * lanes, kBlock)), dim3(kBlock), 0,
                                                                 st, m, (i64)m_eq, a.ptr.p, a.val.p, pass, bu.p, s->sc.p, bl.p));
                    SLP_HIP(hipGetLastError());
                }
            }
            if (!deferred) ensure_transposed(a_ineq);
            // c2 = [c; 0]  lb2 = [lb; -inf]  ub2 = [ub; bu']  b = 0  x0 = 0   (equality rows: b = b_eq'', slack pinned to 0)
            const size_t N = (size_t)s->N;
            s->c.alloc(N); s->lb.alloc(N); s->ub.alloc(N); s->x.alloc(N); s->b.alloc((size_t)m);
            s->c.zero(); s->x.zero(); s->b.zero();
            SLP_HIP(hipMemcpyAsync(s->c.p, c, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            SLP_HIP(hipMemcpyAsync(s->lb.p, lb, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            SLP_HIP(hipMemcpyAsync(s->ub.p, ub, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            if (m) {
                hipLaunchKernelGGL(k_cg_split_rows, dim3(grid_for(m, kBlock)), dim3(kBlock), 0, st, m, (i64)m_eq, bu.p, bl.p, s->b.p, s->lb.p + n,
                                   s->ub.p + n);
                SLP_HIP(hipGetLastError());
            }
            SLP_HIP(hipStreamSynchronize(st));
            cg_alloc_state(s);
            s->xp.zero();  // max(0, 0)
            SLP_HIP(hipStreamSynchronize(st));
        } catch (...) {
            delete s;
            throw;
        }
        ++a_ineq->borrowers;
        return s;
    })
}

void slp_admm_cg_destroy(slp_admm_cg *s) {
    if (!s) return;
    if (s->owns_a) delete s->a;
    else --s->a->borrowers;
    delete s;
}

int slp_admm_cg_iterate(slp_admm_cg *s, int64_t k) {
    SLP_API_INT({
        SLP_REQUIRE(s && k >= 0, "slp_admm_cg_iterate: bad arguments");
        auto one = [&]() { cg_xstep(s); cg_multipliers(s); s->started = true; };
        if (k > 0 && !s->started) {  // the first iteration builds the derived formats and the shared products
            one();
            --k;
        }
        // ~30 small kernels per iteration: replay them as a graph when the matrix is cache-sized (no collectives inside)
        if (!s->distributed && s->a->a.nnz <= 20000000 && s->reuse < 3) s->graph.run(k, 4, one);  // (level 3 alternates two sequences)
        else for (i64 it = 0; it < k; ++it) one();
    })
}

int slp_admm_cg_set_reuse(slp_admm_cg *s, int reuse) {
    SLP_API_INT({
        SLP_REQUIRE(s, "NULL handle");
        cg_gather_state(s);
        if (reuse && s->mx.n < (size_t)s->N) { s->mx.alloc((size_t)s->N); s->md.alloc((size_t)s->N); s->mx.zero(); s->md.zero(); }
        s->reuse = reuse < 0 ? 0 : (reuse > 4 ? 4 : reuse);
        s->since_refresh = 0;
        s->need_md = true;
        s->have_w = false;
        s->started = false;
        s->graph.reset();
    })
}

int slp_admm_cg_xstep(slp_admm_cg *s) {
    SLP_API_INT({
        SLP_REQUIRE(s, "NULL handle");
        cg_xstep(s);
        if (s->sharded && s->shard_dirty) comm_all_gather_dev(s->x.p, s->scnt);  // the caller may read x between the halves (report, callback)
    })
}

int slp_admm_cg_multiplier_step(slp_admm_cg *s) {
    SLP_API_INT({ SLP_REQUIRE(s, "NULL handle"); cg_multipliers(s); s->started = true; })
}

int slp_admm_cg_report(slp_admm_cg *s, double out[3]) {
    SLP_API_INT({
        SLP_REQUIRE(s && out, "slp_admm_cg_report: NULL argument");
        hipStream_t st = ctx().stream;
        if (s->sharded && s->shard_dirty) {   // x is whole (slp_admm_cg_xstep gathered it); the sums below also walk xp and lin
            comm_all_gather_dev(s->xp.p, s->scnt);
            comm_all_gather_dev(s->lin.p, s->scnt);
        }
        cg_rows(s, s->x.p);
        int gr = std::min(grid_for(s->m, kBlock), kCgPartials), gc = std::min(grid_for(s->N, kBlock), kCgPartials);
        hipLaunchKernelGGL(k_cg_report_rows, dim3(gr), dim3(kBlock), 0, st, s->m, s->w.p, s->b.p, s->lam.p, s->rowpart.p);
        hipLaunchKernelGGL(k_cg_report_cols, dim3(gc), dim3(kBlock), 0, st, s->N, s->n_o, s->c.p, s->x.p, s->xp.p, s->lin.p, s->colpart.p);
        hipLaunchKernelGGL(k_cg_report_final, dim3(1), dim3(kBlock), 0, st, gr, s->rowpart.p, gc, s->colpart.p, s->out.p);
        SLP_HIP(hipGetLastError());
        double h[11];
        s->out.download(h, 11);
        if (s->distributed) {
            double sums[5] = {h[0], h[1], h[4], h[6], h[8]}, maxs[2] = {h[2], h[10]};
            SLP_REQUIRE(slp_comm_allreduce_host(sums, 5, 0) == 0, slp_last_error());
            SLP_REQUIRE(slp_comm_allreduce_host(maxs, 2, 1) == 0, slp_last_error());
            h[0] = sums[0]; h[1] = sums[1]; h[4] = sums[2]; h[6] = sums[3]; h[8] = sums[4]; h[2] = maxs[0]; h[10] = maxs[1];
        }
        const double cx = h[3] + h[4], dx2 = h[5] + h[6], ldx = h[7] + h[8];
        const double mneg = h[9] > h[10] ? h[9] : h[10];
        out[0] = cx + 0.5 * s->gamma_eq * h[0] + 0.5 * s->gamma_ineq * dx2 + h[1] + ldx;  // ADMM.py:124-132
        out[1] = h[2];                                                                       // :221
        out[2] = mneg > 0.0 ? mneg : 0.0;                                                    // :222
    })
}

int slp_admm_cg_get_x(slp_admm_cg *s, double *x, int64_t count) {
    SLP_API_INT({ SLP_REQUIRE(s && x && count >= 0 && count <= s->N, "slp_admm_cg_get_x: bad arguments"); s->x.download(x, (size_t)count); })
}

}  // extern "C"
