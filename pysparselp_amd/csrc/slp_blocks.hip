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
enum { B_RS = 0, B_PQ = 1, B_RSNEW = 2, B_ALPHA = 3, B_BETA = 4, B_RHS2 = 5, B_COUNT = 8 };
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

static void blk_iteration(slp_blocks *s) {
    hipStream_t st = ctx().stream;
    const i64 P = s->P, m = s->m, N = s->N;
    const int gp = grid_for(P, kBlock), gm = grid_for(m, kBlock), gn = grid_for(N, kBlock);
    hipLaunchKernelGGL(k_blk_v, dim3(gp), dim3(kBlock), 0, st, P, s->owner.p, s->xp.p, s->lam.p, s->gamma, s->v.p);
    if (m > 0) {
        // (A^ A^T) nu = A^ v - b, conjugate gradients from the previous nu
        matrix_spmv(s->a, false, s->v.p, s->w.p, SLP_ORDER_AUTO);
        blk_apply(s, s->nu.p, s->q.p);
        hipLaunchKernelGGL(k_blk_resid0, dim3(gm), dim3(kBlock), 0, st, m, s->w.p, s->b.p, s->q.p, s->r.p, s->dir.p, s->rhs.p);
        blk_dot(s, m, s->rhs.p, s->rhs.p, B_RHS2, 0);
        blk_dot(s, m, s->r.p, s->r.p, B_RS, 0);
        double h[B_COUNT];
        for (int it = 0; it < s->max_cg;) {
            s->scal.download(h, B_COUNT);  // one 64-byte read every `check_every` steps
            if (!(h[B_RS] > s->tol * s->tol * h[B_RHS2])) break;
            for (int k = 0; k < s->check_every && it < s->max_cg; ++k, ++it) {
                blk_apply(s, s->dir.p, s->q.p);
                blk_dot(s, m, s->dir.p, s->q.p, B_PQ, 1);
                hipLaunchKernelGGL(k_blk_step, dim3(gm), dim3(kBlock), 0, st, m, s->scal.p, s->dir.p, s->q.p, s->nu.p, s->r.p);
                blk_dot(s, m, s->r.p, s->r.p, B_RSNEW, 2);
                hipLaunchKernelGGL(k_blk_dir, dim3(gm), dim3(kBlock), 0, st, m, s->scal.p, s->r.p, s->dir.p);
                ++s->cg_steps;
            }
        }
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

void slp_blocks_destroy(slp_blocks *s) {
    if (!s) return;
    delete s->a;
    delete s;
}

int slp_blocks_set_cg(slp_blocks *s, double tol, int max_steps) {
    SLP_API_INT({
        SLP_REQUIRE(s && tol > 0.0 && max_steps > 0, "slp_blocks_set_cg: bad arguments");
        s->tol = tol;
        s->max_cg = max_steps;
    })
}

int slp_blocks_iterate(slp_blocks *s, int64_t k) {
    SLP_API_INT({
        SLP_REQUIRE(s && k >= 0, "slp_blocks_iterate: bad arguments");
        for (i64 it = 0; it < k; ++it) blk_iteration(s);
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
        if (s->P > 0) {
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

int slp_blocks_get_xp(slp_blocks *s, double *xp, int64_t count) {
    SLP_API_INT({
        SLP_REQUIRE(s && xp && count >= 0 && count <= s->N, "slp_blocks_get_xp: bad arguments");
        s->xp.download(xp, (size_t)count);
    })
}

}  // extern "C"
