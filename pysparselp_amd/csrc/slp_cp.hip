// slp_cp.hip -- diagonally preconditioned Chambolle-Pock on the device.
// Replaces the setup and the loop of chambolle_pock_ppd
// (ChambollePockPPD.py:122-179 and :195-343).  One iteration = two kernels:
//   k_cp_primal : d = c + K^T y (walks the columns of K = rows of K^T),
//                 x+ = clip(x - T d), z = (1+theta) x+ - theta x   [:198-228]
//   k_cp_dual   : r = K z - b (walks the rows of K), y += Sigma r,
//                 inequality rows clamped at 0                     [:231-240,:333-342]
#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

constexpr int kMaxPartials = 4096;

// ---------------------------------------------------------------------------
// setup: T_j = 1 / (sum_i |Ke_ij|^(2-alpha) + sum_i |Ki_ij|^(2-alpha)),  0 -> 1   (:122-153)
//        S_i = 1 / sum_j |K_ij|^alpha, 0 -> 1                                      (:158-179)
// column sums over the transposed matrix (rows of K^T = columns of K), eq and ineq parts apart
__global__ void k_cp_colsum(i64 n, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const double *__restrict__ val,
                            i32 m_eq, i64 m_ineq, double p, double *__restrict__ part_or_t, int finalize) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        double se = 0.0, si = 0.0;
        for (i64 k = ptr[j]; k < ptr[j + 1]; ++k) {
            const double t = abs_pow(val[k], p) * 1.0;
            if (idx[k] < m_eq) se += t;
            else si += t;
        }
        double tmp;
        if (m_eq > 0 && m_ineq > 0) tmp = (0.0 + se) + si;
        else if (m_eq > 0) tmp = se;
        else tmp = si;
        if (finalize) {
            if (tmp == 0.0) tmp = 1.0;
            part_or_t[j] = 1.0 / tmp;
        } else {
            part_or_t[j] = tmp;  // multi-GPU: summed over ranks first
        }
    }
}

// |dict|^p for the strip copies' value table; fill; T from the two partial column sums
__global__ void k_cp_indicator(i64 m, i64 lo, i64 hi, double *__restrict__ y) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) y[i] = (i >= lo && i < hi) ? 1.0 : 0.0;
}
__global__ void k_cp_join_sums(i64 n, const double *__restrict__ se, const double *__restrict__ si, double *__restrict__ t) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) t[j] = (0.0 + se[j]) + si[j];
}

__global__ void k_invert_or_one(i64 n, double *__restrict__ v) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        double t = v[j];
        if (t == 0.0) t = 1.0;
        v[j] = 1.0 / t;
    }
}

__global__ void k_cp_rowsum(i64 m, const i64 *__restrict__ ptr, const double *__restrict__ val, double p,
                            double *__restrict__ sigma) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) s += abs_pow(val[k], p) * 1.0;
        if (s == 0.0) s = 1.0;
        sigma[i] = 1.0 / s;
    }
}

// ---------------------------------------------------------------------------
// primal half-iteration.  `pre` (optional) holds K^T y already formed (strip copies; multi-GPU: summed over the
// ranks), `pre2` the inequality rows' share when the two kinds of rows are multiplied apart; otherwise the column
// walk happens here.
template <int L, bool FROM_PRE>
__global__ __launch_bounds__(kBlock) void k_cp_primal(i64 n, const i64 *__restrict__ tptr, const i32 *__restrict__ tidx,
                                                      const double *__restrict__ tval, const double *__restrict__ y,
                                                      const double *__restrict__ pre, const double *__restrict__ pre2,
                                                      const double *__restrict__ c,
                                                      const double *__restrict__ t, const double *__restrict__ lb,
                                                      const double *__restrict__ ub, double *__restrict__ x,
                                                      double *__restrict__ z, double *__restrict__ d_out, i32 m_eq,
                                                      i64 m_ineq, double one_plus_theta, double theta) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 j = group; j < n; j += ngroups) {
        double d;
        if (FROM_PRE) {
            // pre2: equality and inequality rows were multiplied apart (pre = A_e^T y_e, pre2 = A_i^T y_i): the reference's
            // d = (c + y_eq * a_eq) + y_ineq * a_ineq  (:206,216)
            d = pre2 ? (c[j] + pre[j]) + pre2[j] : c[j] + pre[j];
        } else {
            double se, si;
            row_dot_split<L>(tptr, tidx, tval, y, j, sub, m_eq, &se, &si);
            if (m_eq > 0 && m_ineq > 0) d = (c[j] + se) + si;  // :206,216
            else if (m_eq > 0) d = c[j] + se;
            else d = c[j] + si;
        }
        if (sub == 0) {
            const double xo = x[j];
            double x2 = xo - t[j] * d;  // :220
            const double l = lb[j], u = ub[j];
            x2 = (x2 < l) ? l : x2;  // np.maximum(x2, lb)
            x2 = (x2 > u) ? u : x2;  // np.minimum(x2, ub)
            z[j] = one_plus_theta * x2 - theta * xo;  // :226
            x[j] = x2;
            if (d_out) d_out[j] = d;
        }
    }
}

// partial K_g^T y_g only (multi-GPU), before the all-reduce
template <int L>
__global__ __launch_bounds__(kBlock) void k_cp_colsum_y(i64 n, const i64 *__restrict__ tptr, const i32 *__restrict__ tidx,
                                                        const double *__restrict__ tval, const double *__restrict__ y,
                                                        double *__restrict__ out) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 j = group; j < n; j += ngroups) {
        const double s = row_dot<L>(tptr, tidx, tval, y, j, sub);
        if (sub == 0) out[j] = s;
    }
}

// dual half-iteration
template <int L>
__global__ __launch_bounds__(kBlock) void k_cp_dual(i64 m, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                    const double *__restrict__ val, const double *__restrict__ z,
                                                    const double *__restrict__ b, const double *__restrict__ sigma,
                                                    double *__restrict__ y, i64 m_eq) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 i = group; i < m; i += ngroups) {
        const double kz = row_dot<L>(ptr, idx, val, z, i, sub);
        if (sub == 0) {
            const double r = kz - b[i];          // :235,240
            double yn = y[i] + sigma[i] * r;     // :334,339
            if (i >= m_eq) yn = (yn < 0.0) ? 0.0 : yn;  // :341
            y[i] = yn;
        }
    }
}

// ---------------------------------------------------------------------------
// Short rows (Potts: 3 entries per row, <= 8 per column; netlib): ELL copies of K and K^T, column-major, padded to
// W entries per row.  A thread reads its row's W (index, value) pairs with coalesced, unconditional loads (pads:
// index 0, value 0, skipped in the sum) -- two dependent memory hops per half-iteration (entries -> gather)
// instead of the three of CSR (row pointer -> entries -> gather); these LPs are latency-bound, not bandwidth-bound.
// Sums run over e = 0 .. len-1 in storage order: the same sequential sums as k_cp_primal<1> / k_cp_dual<1>.
// Packed form (few distinct stored values, D <= 256, and fewer than 2^24 rows / columns -- Potts: +-1): an entry is ONE 32-bit
// word, index | value id << 24, the D values sit in LDS: 4 bytes per entry instead of 12.  These LPs live in the caches: the
// iteration's time is the bytes it pulls through the L2s from the Infinity Cache, and the ELL copies were two thirds of them.
// The looked-up value is the fp64 number the CSR holds: the same sums, bit for bit.
__host__ __device__ inline unsigned long long cp_value_key(unsigned long long bits) {  // order-preserving image (slp_strip.hip)
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
}

__global__ void k_ell_fill(i64 nrow, int W, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const double *__restrict__ val,
                           i32 *__restrict__ oidx, double *__restrict__ oval, unsigned char *__restrict__ olen, int *__restrict__ bad,
                           int D, const unsigned long long *__restrict__ dkeys) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        const i64 s = ptr[r];
        const i64 full = ptr[r + 1] - s;
        if (full > W) atomicOr(bad, 1);  // a row longer than the width: the copy must not be used
        const int len = (int)(full < W ? full : W);
        olen[r] = (unsigned char)len;
        for (int e = 0; e < W; ++e) {
            if (D > 0) {  // packed: index | value id << 24
                unsigned int w = 0;
                if (e < len) {
                    const unsigned long long key = cp_value_key((unsigned long long)__double_as_longlong(val[s + e]));
                    int lo = 0, hi = D - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (dkeys[mid] < key) lo = mid + 1;
                        else hi = mid;
                    }
                    if (dkeys[lo] != key || (unsigned)idx[s + e] >= (1u << 24)) atomicOr(bad, 2);
                    w = (unsigned int)idx[s + e] | ((unsigned int)lo << 24);
                }
                oidx[(i64)e * nrow + r] = (i32)w;
            } else {
                oidx[(i64)e * nrow + r] = e < len ? idx[s + e] : 0;
                oval[(i64)e * nrow + r] = e < len ? val[s + e] : 0.0;
            }
        }
    }
}

// PACKED: eidx holds index | value id << 24 and eval the D <= 256 values (k_ell_fill)
template <int W, bool PACKED>
__global__ __launch_bounds__(kBlock) void k_cp_primal_ell(i64 n, const unsigned char *__restrict__ len, const i32 *__restrict__ eidx,
                                                          const double *__restrict__ eval, const double *__restrict__ y,
                                                          const double *__restrict__ c, const double *__restrict__ t,
                                                          const double *__restrict__ lb, const double *__restrict__ ub,
                                                          double *__restrict__ x, double *__restrict__ z, double *__restrict__ d_out,
                                                          i32 m_eq, i64 m_ineq, double one_plus_theta, double theta, int D) {
    __shared__ double tab[PACKED ? 256 : 1];
    if (PACKED) {
        if ((int)threadIdx.x < D) tab[threadIdx.x] = eval[threadIdx.x];
        __syncthreads();
    }
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        i32 ix[W];
        double v[W], g[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const i32 raw = eidx[(i64)e * n + j];
            ix[e] = PACKED ? (i32)((unsigned int)raw & 0xffffffu) : raw;
            v[e] = PACKED ? tab[(unsigned int)raw >> 24] : eval[(i64)e * n + j];
        }
        const int L = len[j];
        const double cj = c[j], tj = t[j], l = lb[j], u = ub[j], xo = x[j];
#pragma unroll
        for (int e = 0; e < W; ++e) g[e] = y[ix[e]];
        double se = 0.0, si = 0.0;
#pragma unroll
        for (int e = 0; e < W; ++e)
            if (e < L) {
                if (ix[e] < m_eq) se += v[e] * g[e];
                else si += v[e] * g[e];
            }
        double d;
        if (m_eq > 0 && m_ineq > 0) d = (cj + se) + si;  // :206,216
        else if (m_eq > 0) d = cj + se;
        else d = cj + si;
        double x2 = xo - tj * d;  // :220
        x2 = (x2 < l) ? l : x2;
        x2 = (x2 > u) ? u : x2;
        z[j] = one_plus_theta * x2 - theta * xo;  // :226
        x[j] = x2;
        if (d_out) d_out[j] = d;
    }
}

template <int W, bool PACKED>
__global__ __launch_bounds__(kBlock) void k_cp_dual_ell(i64 m, const unsigned char *__restrict__ len, const i32 *__restrict__ eidx,
                                                        const double *__restrict__ eval, const double *__restrict__ z,
                                                        const double *__restrict__ b, const double *__restrict__ sigma,
                                                        double *__restrict__ y, i64 m_eq, int D) {
    __shared__ double tab[PACKED ? 256 : 1];
    if (PACKED) {
        if ((int)threadIdx.x < D) tab[threadIdx.x] = eval[threadIdx.x];
        __syncthreads();
    }
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        i32 ix[W];
        double v[W], g[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const i32 raw = eidx[(i64)e * m + i];
            ix[e] = PACKED ? (i32)((unsigned int)raw & 0xffffffu) : raw;
            v[e] = PACKED ? tab[(unsigned int)raw >> 24] : eval[(i64)e * m + i];
        }
        const int L = len[i];
        const double bi = b[i], sg = sigma[i], yo = y[i];
#pragma unroll
        for (int e = 0; e < W; ++e) g[e] = z[ix[e]];
        double kz = 0.0;
#pragma unroll
        for (int e = 0; e < W; ++e)
            if (e < L) kz += v[e] * g[e];
        const double r = kz - bi;       // :235,240
        double yn = yo + sg * r;        // :334,339
        if (i >= m_eq) yn = (yn < 0.0) ? 0.0 : yn;  // :341
        y[i] = yn;
    }
}

// dual half-iteration from a precomputed K z (LDS-tiled SpMV path)
__global__ void k_cp_dual_from(i64 m, const double *__restrict__ kz, const double *__restrict__ b,
                               const double *__restrict__ sigma, double *__restrict__ y, i64 m_eq) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        const double r = kz[i] - b[i];
        double yn = y[i] + sigma[i] * r;
        if (i >= m_eq) yn = (yn < 0.0) ? 0.0 : yn;
        y[i] = yn;
    }
}

// ---------------------------------------------------------------------------
// report (:242-329).  x4_j = ub_j if d_j < 0 else lb_j (:260-261).
// Row pass: per row three dot products (K x, K x4, K z); partial sums / maxima per workgroup:
//   part[0] sum y_i (Kx - b)_i   part[1] sum y_i (Kx4 - b)_i
//   part[2] max_{i<m_eq} |Kz - b|   part[3] max_{i>=m_eq} (Kx - b)   part[4] max_{i<m_eq} |Kx - b|
template <int L>
__global__ __launch_bounds__(kBlock) void k_cp_report_rows(i64 m, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                           const double *__restrict__ val, const double *__restrict__ x,
                                                           const double *__restrict__ x4, const double *__restrict__ z,
                                                           const double *__restrict__ b, const double *__restrict__ y,
                                                           i64 m_eq, double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    double s1 = 0.0, s2 = 0.0, veq = -__builtin_inf(), vin = -__builtin_inf(), veqx = -__builtin_inf();
    const i64 rounds = (m + ngroups - 1) / ngroups;
    for (i64 it = 0; it < rounds; ++it) {
        const i64 i = group + it * ngroups;
        if (i < m) {
            const double kx = row_dot<L>(ptr, idx, val, x, i, sub);
            const double kx4 = row_dot<L>(ptr, idx, val, x4, i, sub);
            const double kz = row_dot<L>(ptr, idx, val, z, i, sub);
            if (sub == 0) {
                const double bi = b[i], yi = y[i];
                s1 += yi * (kx - bi);
                s2 += yi * (kx4 - bi);
                if (i < m_eq) {
                    const double a = fabs(kz - bi), ax = fabs(kx - bi);
                    veq = a > veq ? a : veq;
                    veqx = ax > veqx ? ax : veqx;
                } else {
                    const double v = kx - bi;
                    vin = v > vin ? v : vin;
                }
            }
        }
    }
    const double r0 = block_reduce<false>(s1, lds);
    const double r1 = block_reduce<false>(s2, lds);
    const double r2 = block_reduce<true>(veq, lds);
    const double r3 = block_reduce<true>(vin, lds);
    const double r4 = block_reduce<true>(veqx, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 5 + 0] = r0;
        part[blockIdx.x * 5 + 1] = r1;
        part[blockIdx.x * 5 + 2] = r2;
        part[blockIdx.x * 5 + 3] = r3;
        part[blockIdx.x * 5 + 4] = r4;
    }
}

// The same partials from the three products taken beforehand (strip copies: K x, K x4, K z as vectors) -- the form the
// report takes whenever the matrix runs on strip copies, with or without its CSR arrays (slp_matrix_release_csr).
__global__ __launch_bounds__(kBlock) void k_cp_report_rows_from(i64 m, const double *__restrict__ kxv, const double *__restrict__ kx4v,
                                                                const double *__restrict__ kzv, const double *__restrict__ b,
                                                                const double *__restrict__ y, i64 m_eq, double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double s1 = 0.0, s2 = 0.0, veq = -__builtin_inf(), vin = -__builtin_inf(), veqx = -__builtin_inf();
    for (i64 i = (i64)blockIdx.x * kBlock + threadIdx.x; i < m; i += (i64)gridDim.x * kBlock) {
        const double kx = kxv[i], kx4 = kx4v[i], kz = kzv[i], bi = b[i], yi = y[i];
        s1 += yi * (kx - bi);
        s2 += yi * (kx4 - bi);
        if (i < m_eq) {
            const double a = fabs(kz - bi), ax = fabs(kx - bi);
            veq = a > veq ? a : veq;
            veqx = ax > veqx ? ax : veqx;
        } else {
            const double v = kx - bi;
            vin = v > vin ? v : vin;
        }
    }
    const double r0 = block_reduce<false>(s1, lds);
    const double r1 = block_reduce<false>(s2, lds);
    const double r2 = block_reduce<true>(veq, lds);
    const double r3 = block_reduce<true>(vin, lds);
    const double r4 = block_reduce<true>(veqx, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 5 + 0] = r0;
        part[blockIdx.x * 5 + 1] = r1;
        part[blockIdx.x * 5 + 2] = r2;
        part[blockIdx.x * 5 + 3] = r3;
        part[blockIdx.x * 5 + 4] = r4;
    }
}

// column pass: x4 and the two cost dot products; part[0] sum c x, part[1] sum c x4
__global__ __launch_bounds__(kBlock) void k_cp_report_cols(i64 n, const double *__restrict__ c, const double *__restrict__ x,
                                                           const double *__restrict__ d, const double *__restrict__ lb,
                                                           const double *__restrict__ ub, double *__restrict__ x4,
                                                           double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double s0 = 0.0, s1 = 0.0;
    for (i64 j = (i64)blockIdx.x * kBlock + threadIdx.x; j < n; j += (i64)gridDim.x * kBlock) {
        const double v4 = (d[j] < 0.0) ? ub[j] : lb[j];
        x4[j] = v4;
        s0 += c[j] * x[j];
        s1 += c[j] * v4;
    }
    const double r0 = block_reduce<false>(s0, lds);
    const double r1 = block_reduce<false>(s1, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = r0;
        part[blockIdx.x * 2 + 1] = r1;
    }
}

// out[0]=energy1 out[1]=energy2 out[2]=max eq out[3]=max ineq ; one workgroup, fixed order
__global__ __launch_bounds__(kBlock) void k_cp_report_final(int nrowparts, const double *__restrict__ rp, int ncolparts,
                                                            const double *__restrict__ cp, double *__restrict__ out) {
    __shared__ double lds[kBlock / kWave];
    double s1 = 0.0, s2 = 0.0, veq = -__builtin_inf(), vin = -__builtin_inf(), veqx = -__builtin_inf(), c0 = 0.0, c1 = 0.0;
    for (int i = threadIdx.x; i < nrowparts; i += kBlock) {
        s1 += rp[i * 5 + 0];
        s2 += rp[i * 5 + 1];
        veq = rp[i * 5 + 2] > veq ? rp[i * 5 + 2] : veq;
        vin = rp[i * 5 + 3] > vin ? rp[i * 5 + 3] : vin;
        veqx = rp[i * 5 + 4] > veqx ? rp[i * 5 + 4] : veqx;
    }
    for (int i = threadIdx.x; i < ncolparts; i += kBlock) {
        c0 += cp[i * 2 + 0];
        c1 += cp[i * 2 + 1];
    }
    const double r0 = block_reduce<false>(s1, lds), r1 = block_reduce<false>(s2, lds);
    const double r2 = block_reduce<true>(veq, lds), r3 = block_reduce<true>(vin, lds);
    const double r4 = block_reduce<false>(c0, lds), r5 = block_reduce<false>(c1, lds);
    const double r6 = block_reduce<true>(veqx, lds);
    if (threadIdx.x == 0) {
        out[6] = r6;
        out[0] = r4;  // c.x            (row sums are added on the host after the cross-rank reduction)
        out[1] = r5;  // c.x4
        out[2] = r0;  // y.(Kx-b)
        out[3] = r1;  // y.(Kx4-b)
        out[4] = r2;
        out[5] = r3;
    }
}

}  // namespace slp

using namespace slp;

// comm hooks (slp_comm.hip)
namespace slp {
bool comm_active();
void comm_allreduce_dev(double *buf, i64 count, int op);  // in place, on the library stream
}

struct slp_cp {
    slp_matrix *k = nullptr;
    bool owns_k = false;
    i64 n = 0, m = 0, m_eq = 0, m_ineq = 0;
    double alpha = 1, theta = 1;
    int order = SLP_ORDER_AUTO;
    int lanes_rows = 1, lanes_cols = 1;
    DevBuf<double> b, c, lb, ub, t, sigma, x, z, y, d, x4, pre, pre2, ymask, kz, rep, rowparts, colparts, out;
    // Equality AND inequality rows on strip copies (cp_split_setup): the reference adds a column's two partial sums apart,
    // d = (c + y_eq * a_eq) + y_ineq * a_ineq (ChambollePockPPD.py:206,216).  split = 1: kt[0] / kt[1] are copies of
    // A_e^T and A_i^T -- views over the chunks of a chunked matrix cut at m_eq, or copies of the two row ranges that the
    // solver owns -- multiplied by y + kt_off[]; split = 2: two products over the copy of the whole K^T with the other kind
    // of rows masked out of y (adding +-0.0 to a running sum leaves it as it is: the same two chains, at twice the bytes).
    int split = 0;
    StripJds view[2];
    slp_matrix *own[2] = {nullptr, nullptr};
    const StripJds *kt[2] = {nullptr, nullptr};
    i64 kt_off[2] = {0, 0};
    // ELL copies for short rows (0 = not used)
    int ell_w_rows = 0, ell_w_cols = 0;
    int ell_D = 0;                 // > 0: the ELL copies are packed (index | value id << 24), the values are k's dictionary
    DevBuf<i32> ell_idx_rows, ell_idx_cols;
    DevBuf<double> ell_val_rows, ell_val_cols;
    DevBuf<unsigned char> ell_len_rows, ell_len_cols;
    bool distributed = false;
    bool csr_bound = false;  // an iteration half walks the CSR arrays of `k` (counted in k->csr_bound for borrowed matrices)
    IterGraph graph;
    ~slp_cp() {
        delete own[0];
        delete own[1];
    }
};

namespace slp {

// Both kinds of rows, one GPU: the two partial sums of a column are formed apart (see slp_cp::split).
static bool cp_mixed(const slp_cp *s) { return !s->distributed && s->m_eq > 0 && s->m_ineq > 0; }

// The copy of K^T a single product of the primal half runs on (all rows of one kind, or the rows partitioned over several
// GPUs, where the all-reduce re-associates the column sums anyway), else NULL.
static const StripJds *cp_primal_strips(slp_cp *s) {
    if (cp_mixed(s)) return nullptr;
    return fast_format(s->k, true);
}

__global__ void k_cp_mask(i64 m, i64 lo, i64 hi, const double *__restrict__ y, double *__restrict__ out) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) out[i] = (i >= lo && i < hi) ? y[i] : 0.0;
}

// out = (rows of kind `which` of K)^T y -- which = 0: the equality rows, 1: the inequality rows; pw >= 0: of |K|^pw
static void cp_kt_product(slp_cp *s, int which, const double *y, double *out, double pw = -1.0) {
    hipStream_t st = ctx().stream;
    if (s->split == 1) {
        if (pw >= 0.0) strip_spmv_abs_pow(*s->kt[which], pw, y + s->kt_off[which], out);
        else strip_spmv(*s->kt[which], y + s->kt_off[which], out);
        return;
    }
    const StripJds *f = fast_format(s->k, true);
    SLP_REQUIRE(f, "Chambolle-Pock: no strip copy of K^T");
    if (s->ymask.n < (size_t)s->m) s->ymask.alloc((size_t)s->m);
    hipLaunchKernelGGL(k_cp_mask, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, which ? s->m_eq : (i64)0, which ? s->m : s->m_eq, y,
                       s->ymask.p);
    SLP_HIP(hipGetLastError());
    if (pw >= 0.0) strip_spmv_abs_pow(*f, pw, s->ymask.p, out);
    else strip_spmv(*f, s->ymask.p, out);
}

// Settles how the two partial column sums are formed when K has both kinds of rows and runs on strip copies (slp_cp::split).
// SLP_CP_SPLIT=masked forces the two masked products (tests).
static void cp_split_setup(slp_cp *s) {
    s->split = 0;
    if (!cp_mixed(s)) return;
    slp_matrix *k = s->k;
    const bool chunked = !k->chunks.empty();
    // an ordinary matrix whose copy of K^T would be the CSR walk keeps that walk: it forms the two sums in one pass (row_dot_split)
    if (!chunked && !k->csr_released && !fast_format(k, true)) return;
    if (!chunked && k->csr_released && !k->fat.ok) return;
    const char *e = getenv("SLP_CP_SPLIT");
    const bool want_masked = e && !strcmp(e, "masked");
    s->pre2.alloc((size_t)s->n);
    s->split = 2;
    if (want_masked) return;
    if (chunked) {
        // the chunks were cut at m_eq (problems.random_lp_on_device(m_eq=...), ChunkedDeviceMatrix.from_csr(cut_at=...)): the
        // equality chunks and the inequality chunks are two composites of their own -- two launches over tall cells, no CSR
        size_t ke = 0;
        while (ke < k->chunks.size() && k->chunk_row0[ke] < s->m_eq) ++ke;
        if (ke == 0 || ke >= k->chunks.size() || k->chunk_row0[ke] != s->m_eq) return;
        composite_of_chunks(k, true, 0, ke, s->view[0]);
        composite_of_chunks(k, true, ke, k->chunks.size(), s->view[1]);
        s->kt[0] = &s->view[0];
        s->kt[1] = &s->view[1];
        s->kt_off[0] = s->kt_off[1] = 0;   // (a composite's parts name their own slice of y)
        s->split = 1;
        return;
    }
    // an ordinary matrix with its CSR: copies of (A_e)^T and (A_i)^T of the solver's own, each built from its row range like a
    // chunk's (the strip kernels stage y + m_eq with 16-byte loads: m_eq even)
    if (k->csr_released || (s->m_eq & 1)) return;
    Phase ph("cp_split_setup: copies of A_eq^T and A_ineq^T");
    try {
        for (int w = 0; w < 2; ++w) {
            const i64 r0 = w ? s->m_eq : 0, r1 = w ? s->m : s->m_eq;
            s->own[w] = matrix_row_slice(k, r0, r1);
            s->own[w]->format_policy = k->format_policy;
            const StripJds *f = fast_format(s->own[w], true);
            if (!f) {   // a row range too small for a strip copy of its own: the masked products over the whole copy
                delete s->own[0]; delete s->own[1];
                s->own[0] = s->own[1] = nullptr;
                return;
            }
            matrix_drop_csr(s->own[w]);
            s->kt[w] = f;
            s->kt_off[w] = r0;
        }
    } catch (...) {
        delete s->own[0]; delete s->own[1];
        s->own[0] = s->own[1] = nullptr;
        throw;
    }
    s->split = 1;
}

static void cp_setup(slp_cp *s) {
    hipStream_t st = ctx().stream;
    // derived formats are settled here, never lazily inside a (possibly captured) iteration; the transposed CSR is formed only
    // when something walks it (no strip copy of K^T, or the (c + s_eq) + s_ineq order of cp_primal_strips)
    s->distributed = comm_active();
    ensure_transposed(s->k);
    cp_split_setup(s);
    if (!cp_primal_strips(s) && !s->split && s->k->chunks.empty()) build_transpose(s->k);
    if (fast_format(s->k, false)) s->kz.alloc((size_t)s->m);
    const CsrDev &a = s->k->a, &at = s->k->at;
    s->lanes_rows = lanes_for(a, s->order);
    s->lanes_cols = lanes_for(at, s->order);
    s->t.alloc((size_t)s->n);
    s->sigma.alloc((size_t)s->m);
    // short rows in both orientations, one lane per row, single GPU: ELL copies (SLP_CP_ELL=0 keeps the CSR walk)
    {
        const char *ee = getenv("SLP_CP_ELL");
        auto width = [](i64 maxlen) { return maxlen <= 4 ? 4 : (maxlen <= 8 ? 8 : (maxlen <= 16 ? 16 : 0)); };
        const bool small = a.nnz <= 50000000 && !(ee && ee[0] == '0') && !s->distributed && !fast_format(s->k, false) && !fast_format(s->k, true);
        // packed entries when the matrix has at most 256 distinct values and 24-bit indices (SLP_CP_PACKED=0: fp64 + int32 entries)
        const char *ep = getenv("SLP_CP_PACKED");
        const bool packed = small && !(ep && ep[0] == '0') && s->n < ((i64)1 << 24) && s->m < ((i64)1 << 24) && matrix_dictionary(s->k) &&
                            s->k->vdict.D <= 256;
        s->ell_D = packed ? s->k->vdict.D : 0;
        auto build = [&](const CsrDev &csr, int &W, DevBuf<i32> &ei, DevBuf<double> &ev, DevBuf<unsigned char> &el) {
            W = width(csr.max_row_len);
            if (!W || csr.nrow == 0) { W = 0; return; }
            ei.alloc((size_t)W * (size_t)csr.nrow); el.alloc((size_t)csr.nrow);
            if (s->ell_D) ev.copy_from(s->k->vdict.values);  // the table
            else ev.alloc((size_t)W * (size_t)csr.nrow);
            DevBuf<int> bad(1);
            bad.zero();
            hipLaunchKernelGGL(k_ell_fill, dim3(grid_for(csr.nrow, kBlock)), dim3(kBlock), 0, st, csr.nrow, W, csr.ptr.p, csr.idx.p,
                               csr.val.p, ei.p, ev.p, el.p, bad.p, s->ell_D, s->ell_D ? s->k->vdict.keys.p : (const unsigned long long *)nullptr);
            SLP_HIP(hipGetLastError());
            int hbad = 0;
            bad.download(&hbad, 1);
            if (hbad) {  // (max_row_len out of date): keep the CSR walk
                W = 0;
                ei.release(); ev.release(); el.release();
            }
        };
        if (small && s->lanes_rows == 1) build(a, s->ell_w_rows, s->ell_idx_rows, s->ell_val_rows, s->ell_len_rows);
        if (small && s->lanes_cols == 1) build(at, s->ell_w_cols, s->ell_idx_cols, s->ell_val_cols, s->ell_len_cols);
    }
    // Preconditioners (ChambollePockPPD.py:122-179): T_j = 1 / sum_i |K_ij|^(2-alpha), Sigma_i = 1 / sum_j |K_ij|^alpha (0 -> 1).
    // On value-dictionary copies the sums are products with a vector of ones over the copy whose value table is |v|^p --
    // the same chain of additions per column / row (storage order, from 0.0) as the CSR walks below, which read every entry
    // through one thread per row (17-19 x the matrix in HBM traffic at config 3, profiles/r02_c3_pmc_hbm.json) -- and they
    // need no CSR arrays.  Equality and inequality rows are summed apart ((0 + s_eq) + s_ineq, :134,144): two products.
    const StripJds *ft = fast_format(s->k, true), *fr = fast_format(s->k, false);
    if (s->n) {
        if (s->split) {
            // both kinds of rows on strip copies: (0 + s_eq) + s_ineq (:134,144) from the two copies / the two masked products
            DevBuf<double> ones((size_t)std::max<i64>(s->m, 1)), se((size_t)s->n), si((size_t)s->n);
            hipLaunchKernelGGL(k_cp_indicator, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, (i64)0, s->m, ones.p);
            cp_kt_product(s, 0, ones.p, se.p, 2.0 - s->alpha);
            cp_kt_product(s, 1, ones.p, si.p, 2.0 - s->alpha);
            hipLaunchKernelGGL(k_cp_join_sums, dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, se.p, si.p, s->t.p);
            hipLaunchKernelGGL(k_invert_or_one, dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, s->t.p);
            SLP_HIP(hipGetLastError());
            SLP_HIP(hipStreamSynchronize(st));
        } else if (ft && strip_abs_pow_supported(*ft)) {
            DevBuf<double> ones((size_t)std::max<i64>(s->m, 1));
            if (s->m_eq > 0 && s->m_ineq > 0) {
                DevBuf<double> se((size_t)s->n), si((size_t)s->n);
                hipLaunchKernelGGL(k_cp_indicator, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, (i64)0, s->m_eq, ones.p);
                strip_spmv_abs_pow(*ft, 2.0 - s->alpha, ones.p, se.p);
                hipLaunchKernelGGL(k_cp_indicator, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, s->m_eq, s->m, ones.p);
                strip_spmv_abs_pow(*ft, 2.0 - s->alpha, ones.p, si.p);
                hipLaunchKernelGGL(k_cp_join_sums, dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, se.p, si.p, s->t.p);
                SLP_HIP(hipStreamSynchronize(st));
            } else {
                hipLaunchKernelGGL(k_cp_indicator, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, (i64)0, s->m, ones.p);
                strip_spmv_abs_pow(*ft, 2.0 - s->alpha, ones.p, s->t.p);
                SLP_HIP(hipStreamSynchronize(st));
            }
            SLP_HIP(hipGetLastError());
            if (s->distributed) comm_allreduce_dev(s->t.p, s->n, 0);
            hipLaunchKernelGGL(k_invert_or_one, dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, s->t.p);
            SLP_HIP(hipGetLastError());
        } else {
            require_csr(s->k, "Chambolle-Pock preconditioner T (CSR walk)");
            build_transpose(s->k);
            hipLaunchKernelGGL(k_cp_colsum, dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, at.ptr.p, at.idx.p, at.val.p,
                               (i32)s->m_eq, s->m_ineq, 2.0 - s->alpha, s->t.p, s->distributed ? 0 : 1);
            SLP_HIP(hipGetLastError());
            if (s->distributed) {
                comm_allreduce_dev(s->t.p, s->n, 0);
                hipLaunchKernelGGL(k_invert_or_one, dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, s->t.p);
                SLP_HIP(hipGetLastError());
            }
        }
    }
    if (s->m) {
        if (fr && strip_abs_pow_supported(*fr)) {
            DevBuf<double> ones((size_t)std::max<i64>(s->n, 1));
            hipLaunchKernelGGL(k_cp_indicator, dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, (i64)0, s->n, ones.p);
            strip_spmv_abs_pow(*fr, s->alpha, ones.p, s->sigma.p);
            hipLaunchKernelGGL(k_invert_or_one, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, s->sigma.p);
            SLP_HIP(hipGetLastError());
            SLP_HIP(hipStreamSynchronize(st));
        } else {
            require_csr(s->k, "Chambolle-Pock preconditioner Sigma (CSR walk)");
            hipLaunchKernelGGL(k_cp_rowsum, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, st, s->m, a.ptr.p, a.val.p, s->alpha,
                               s->sigma.p);
            SLP_HIP(hipGetLastError());
        }
    }
    // ELL copies are the solver's own; everything else that is not a strip copy walks the matrix's CSR arrays
    s->csr_bound = (s->n > 0 && !cp_primal_strips(s) && !s->split && !s->ell_w_cols) || (s->m > 0 && !fast_format(s->k, false) && !s->ell_w_rows);
}

static void cp_primal(slp_cp *s, bool store_d) {
    hipStream_t st = ctx().stream;
    const CsrDev &at = s->k->at;
    if (s->n == 0) return;
    double *dout = store_d ? s->d.p : nullptr;
    const double opt = 1.0 + s->theta;
    if (s->distributed) {
        if (const StripJds *f = fast_format(s->k, true)) {
            strip_spmv(*f, s->y.p, s->pre.p);
        } else {
            require_csr(s->k, "Chambolle-Pock primal step (CSR walk)");
            const int lanes = s->lanes_cols;
            const int grid = grid_for(s->n * lanes, kBlock);
            SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cp_colsum_y<L>), dim3(grid), dim3(kBlock), 0, st, s->n, at.ptr.p,
                                                         at.idx.p, at.val.p, s->y.p, s->pre.p));
        }
        SLP_HIP(hipGetLastError());
        comm_allreduce_dev(s->pre.p, s->n, 0);
        hipLaunchKernelGGL((k_cp_primal<1, true>), dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, at.ptr.p, at.idx.p,
                           at.val.p, s->y.p, s->pre.p, (const double *)nullptr, s->c.p, s->t.p, s->lb.p, s->ub.p, s->x.p, s->z.p, dout,
                           (i32)s->m_eq, s->m_ineq, opt, s->theta);
    } else if (s->split) {
        // both kinds of rows, long columns: A_e^T y_e and A_i^T y_i as two products, d = (c + s_eq) + s_ineq in the elementwise
        // update -- the reference's order (:206,216) bit for bit, on strip / tall-cell / chunked copies alike
        cp_kt_product(s, 0, s->y.p, s->pre.p);
        cp_kt_product(s, 1, s->y.p, s->pre2.p);
        hipLaunchKernelGGL((k_cp_primal<1, true>), dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, at.ptr.p, at.idx.p,
                           at.val.p, s->y.p, s->pre.p, s->pre2.p, s->c.p, s->t.p, s->lb.p, s->ub.p, s->x.p, s->z.p, dout, (i32)s->m_eq,
                           s->m_ineq, opt, s->theta);
    } else if (const StripJds *f = cp_primal_strips(s)) {
        // long columns: LDS-tiled K^T y, then the elementwise update
        strip_spmv(*f, s->y.p, s->pre.p);
        hipLaunchKernelGGL((k_cp_primal<1, true>), dim3(grid_for(s->n, kBlock)), dim3(kBlock), 0, st, s->n, at.ptr.p, at.idx.p,
                           at.val.p, s->y.p, s->pre.p, (const double *)nullptr, s->c.p, s->t.p, s->lb.p, s->ub.p, s->x.p, s->z.p, dout,
                           (i32)s->m_eq, s->m_ineq, opt, s->theta);
    } else if (s->ell_w_cols) {
        const int grid = grid_for(s->n, kBlock);
#define SLP_ELL_PRIMAL(W, P)                                                                                                       \
    hipLaunchKernelGGL((k_cp_primal_ell<W, P>), dim3(grid), dim3(kBlock), 0, st, s->n, s->ell_len_cols.p, s->ell_idx_cols.p,        \
                       s->ell_val_cols.p, s->y.p, s->c.p, s->t.p, s->lb.p, s->ub.p, s->x.p, s->z.p, dout, (i32)s->m_eq, s->m_ineq, opt, \
                       s->theta, s->ell_D)
        if (s->ell_D) {
            if (s->ell_w_cols == 4) SLP_ELL_PRIMAL(4, true);
            else if (s->ell_w_cols == 8) SLP_ELL_PRIMAL(8, true);
            else SLP_ELL_PRIMAL(16, true);
        } else if (s->ell_w_cols == 4) SLP_ELL_PRIMAL(4, false);
        else if (s->ell_w_cols == 8) SLP_ELL_PRIMAL(8, false);
        else SLP_ELL_PRIMAL(16, false);
#undef SLP_ELL_PRIMAL
    } else {
        require_csr(s->k, "Chambolle-Pock primal step (CSR walk)");
        const int lanes = s->lanes_cols;
        const int grid = grid_for(s->n * lanes, kBlock);
        SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cp_primal<L, false>), dim3(grid), dim3(kBlock), 0, st, s->n, at.ptr.p,
                                                     at.idx.p, at.val.p, s->y.p, nullptr, nullptr, s->c.p, s->t.p, s->lb.p, s->ub.p,
                                                     s->x.p, s->z.p, dout, (i32)s->m_eq, s->m_ineq, opt, s->theta));
    }
    SLP_HIP(hipGetLastError());
}

static void cp_dual(slp_cp *s) {
    if (s->m == 0) return;
    const CsrDev &a = s->k->a;
    if (const StripJds *f = fast_format(s->k, false)) {
        if (s->kz.n < (size_t)s->m) s->kz.alloc((size_t)s->m);
        strip_spmv(*f, s->z.p, s->kz.p);
        hipLaunchKernelGGL(k_cp_dual_from, dim3(grid_for(s->m, kBlock)), dim3(kBlock), 0, ctx().stream, s->m, s->kz.p, s->b.p,
                           s->sigma.p, s->y.p, s->m_eq);
        SLP_HIP(hipGetLastError());
        return;
    }
    if (s->ell_w_rows) {
        const int grid = grid_for(s->m, kBlock);
#define SLP_ELL_DUAL(W, P)                                                                                                         \
    hipLaunchKernelGGL((k_cp_dual_ell<W, P>), dim3(grid), dim3(kBlock), 0, ctx().stream, s->m, s->ell_len_rows.p, s->ell_idx_rows.p, \
                       s->ell_val_rows.p, s->z.p, s->b.p, s->sigma.p, s->y.p, s->m_eq, s->ell_D)
        if (s->ell_D) {
            if (s->ell_w_rows == 4) SLP_ELL_DUAL(4, true);
            else if (s->ell_w_rows == 8) SLP_ELL_DUAL(8, true);
            else SLP_ELL_DUAL(16, true);
        } else if (s->ell_w_rows == 4) SLP_ELL_DUAL(4, false);
        else if (s->ell_w_rows == 8) SLP_ELL_DUAL(8, false);
        else SLP_ELL_DUAL(16, false);
#undef SLP_ELL_DUAL
        SLP_HIP(hipGetLastError());
        return;
    }
    require_csr(s->k, "Chambolle-Pock dual step (CSR walk)");
    const int lanes = s->lanes_rows;
    const int grid = grid_for(s->m * lanes, kBlock);
    SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cp_dual<L>), dim3(grid), dim3(kBlock), 0, ctx().stream, s->m, a.ptr.p,
                                                 a.idx.p, a.val.p, s->z.p, s->b.p, s->sigma.p, s->y.p, s->m_eq));
    SLP_HIP(hipGetLastError());
}

static slp_cp *cp_make(slp_matrix *k, bool owns, i64 m_eq, const double *b, const double *c, const double *lb,
                       const double *ub, const double *x0, double alpha, double theta, int order) {
    auto *s = new slp_cp();
    try {
        s->k = k;
        s->owns_k = owns;
        s->n = k->a.ncol;
        s->m = k->a.nrow;
        s->m_eq = m_eq;
        s->m_ineq = s->m - m_eq;
        SLP_REQUIRE(m_eq >= 0 && m_eq <= s->m, "slp_cp_create: m_eq out of range");
        s->alpha = alpha;
        s->theta = theta;
        s->order = order;
        s->b.upload(b, (size_t)s->m);
        s->c.upload(c, (size_t)s->n);
        s->lb.upload(lb, (size_t)s->n);
        s->ub.upload(ub, (size_t)s->n);
        s->x.alloc((size_t)s->n);
        if (x0) s->x.upload(x0, (size_t)s->n);
        else s->x.zero();
        s->z.copy_from(s->x);  // x3 = x (:190)
        s->y.alloc((size_t)s->m);
        s->y.zero();  // :166,177
        s->d.alloc((size_t)s->n);
        s->x4.alloc((size_t)s->n);
        s->pre.alloc((size_t)s->n);
        s->rowparts.alloc((size_t)kMaxPartials * 5);
        s->colparts.alloc((size_t)kMaxPartials * 2);
        s->out.alloc(8);
        cp_setup(s);
        SLP_HIP(hipStreamSynchronize(ctx().stream));
    } catch (...) {
        if (s->owns_k) delete s->k;
        delete s;
        throw;
    }
    return s;
}

}  // namespace slp

extern "C" {

slp_cp *slp_cp_create(int64_t n, int64_t m_eq, int64_t m_ineq, const int64_t *indptr, const int32_t *indices,
                      const double *data, const double *b, const double *c, const double *lb, const double *ub,
                      const double *x0, double alpha, double theta, int order) {
    SLP_API_PTR({
        SLP_REQUIRE(indptr && b && c && lb && ub, "slp_cp_create: NULL argument");
        slp_matrix *k = slp_matrix_create(m_eq + m_ineq, n, indptr, indices, data);
        if (!k) throw Error(slp_last_error());
        return cp_make(k, true, m_eq, b, c, lb, ub, x0, alpha, theta, order);
    })
}

slp_cp *slp_cp_create_on(slp_matrix *a, int64_t m_eq, const double *b, const double *c, const double *lb,
                         const double *ub, const double *x0, double alpha, double theta, int order) {
    SLP_API_PTR({
        SLP_REQUIRE(a && b && c && lb && ub, "slp_cp_create_on: NULL argument");
        slp_cp *s = cp_make(a, false, m_eq, b, c, lb, ub, x0, alpha, theta, order);
        ++a->borrowers;
        if (s->csr_bound) ++a->csr_bound;
        return s;
    })
}

void slp_cp_destroy(slp_cp *s) {
    if (!s) return;
    if (s->owns_k) delete s->k;
    else {
        --s->k->borrowers;
        if (s->csr_bound) --s->k->csr_bound;
    }
    delete s;
}

int slp_cp_iterate(slp_cp *s, int64_t k) {
    SLP_API_INT({
        SLP_REQUIRE(s && k >= 0, "slp_cp_iterate: bad arguments");
        auto one = [&]() { cp_primal(s, false); cp_dual(s); };
        // cache-resident problems are launch-latency bound: replay a captured graph of 16 iterations
        if (!s->distributed && s->k->a.nnz <= 20000000) s->graph.run(k, 16, one);
        else for (i64 it = 0; it < k; ++it) one();
    })
}

int slp_cp_split_form(const slp_cp *s) { return s ? s->split : -1; }

int slp_cp_primal_step(slp_cp *s) { SLP_API_INT({ SLP_REQUIRE(s, "NULL handle"); cp_primal(s, true); }) }

int slp_cp_dual_step(slp_cp *s) { SLP_API_INT({ SLP_REQUIRE(s, "NULL handle"); cp_dual(s); }) }

int slp_cp_report(slp_cp *s, double out[5]) {
    SLP_API_INT({
        SLP_REQUIRE(s && out, "slp_cp_report: NULL argument");
        hipStream_t st = ctx().stream;
        const CsrDev &a = s->k->a;
        int gc = grid_for(s->n, kBlock);
        if (gc > kMaxPartials) gc = kMaxPartials;
        hipLaunchKernelGGL(k_cp_report_cols, dim3(gc), dim3(kBlock), 0, st, s->n, s->c.p, s->x.p, s->d.p, s->lb.p, s->ub.p,
                           s->x4.p, s->colparts.p);
        SLP_HIP(hipGetLastError());
        const int lanes = s->lanes_rows;
        int gr = grid_for(s->m * lanes, kBlock);
        if (gr > kMaxPartials) gr = kMaxPartials;
        if (const StripJds *f = fast_format(s->k, false)) {
            // strip copies: the three products as vectors (the same values with or without the CSR arrays), then one
            // elementwise pass; the dual step's K z scratch is free between the two halves of an iteration
            if (s->rep.n < 2 * (size_t)s->m) s->rep.alloc(2 * (size_t)s->m);
            if (s->kz.n < (size_t)s->m) s->kz.alloc((size_t)s->m);
            strip_spmv2(*f, s->x.p, s->x4.p, s->rep.p, s->rep.p + s->m);
            strip_spmv(*f, s->z.p, s->kz.p);
            gr = grid_for(s->m, kBlock);
            if (gr > kMaxPartials) gr = kMaxPartials;
            hipLaunchKernelGGL(k_cp_report_rows_from, dim3(gr), dim3(kBlock), 0, st, s->m, s->rep.p, s->rep.p + s->m, s->kz.p, s->b.p,
                               s->y.p, s->m_eq, s->rowparts.p);
        } else {
            require_csr(s->k, "slp_cp_report");  // (unreachable: a release needs strip copies in both orientations)
            SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_cp_report_rows<L>), dim3(gr), dim3(kBlock), 0, st, s->m, a.ptr.p,
                                                         a.idx.p, a.val.p, s->x.p, s->x4.p, s->z.p, s->b.p, s->y.p, s->m_eq,
                                                         s->rowparts.p));
        }
        SLP_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_cp_report_final, dim3(1), dim3(kBlock), 0, st, gr, s->rowparts.p, gc, s->colparts.p, s->out.p);
        SLP_HIP(hipGetLastError());
        double h[7];
        s->out.download(h, 7);
        if (s->distributed) {
            // c.x and c.x4 are replicated; the row terms are summed / maxed over the ranks
            double sums[2] = {h[2], h[3]}, maxs[3] = {h[4], h[5], h[6]};
            SLP_REQUIRE(slp_comm_allreduce_host(sums, 2, 0) == 0, slp_last_error());
            SLP_REQUIRE(slp_comm_allreduce_host(maxs, 3, 1) == 0, slp_last_error());
            h[2] = sums[0]; h[3] = sums[1]; h[4] = maxs[0]; h[5] = maxs[1]; h[6] = maxs[2];
        }
        out[0] = h[0] + h[2];
        out[1] = h[1] + h[3];
        out[2] = (s->m_eq > 0 || s->distributed) ? (h[4] == -__builtin_inf() ? 0.0 : h[4]) : 0.0;
        out[3] = h[5];
        out[4] = (h[6] == -__builtin_inf()) ? 0.0 : h[6];
    })
}

int slp_cp_get_x(slp_cp *s, double *x) { SLP_API_INT({ SLP_REQUIRE(s && x, "NULL argument"); s->x.download(x, (size_t)s->n); }) }

int slp_cp_get_y(slp_cp *s, double *y) { SLP_API_INT({ SLP_REQUIRE(s && y, "NULL argument"); s->y.download(y, (size_t)s->m); }) }

int slp_cp_get_preconditioners(slp_cp *s, double *t, double *sigma) {
    SLP_API_INT({
        SLP_REQUIRE(s, "NULL handle");
        if (t) s->t.download(t, (size_t)s->n);
        if (sigma) s->sigma.download(sigma, (size_t)s->m);
    })
}

int slp_cp_bench(slp_cp *s, int64_t k, double ms[3]) {
    SLP_API_INT({
        SLP_REQUIRE(s && k > 0 && ms, "slp_cp_bench: bad arguments");
        Context &c = ctx();
        float f = 0.f;
        // whole iterations
        SLP_HIP(hipEventRecord(c.ev0, c.stream));
        for (i64 it = 0; it < k; ++it) { cp_primal(s, false); cp_dual(s); }
        SLP_HIP(hipEventRecord(c.ev1, c.stream));
        SLP_HIP(hipEventSynchronize(c.ev1));
        SLP_HIP(hipEventElapsedTime(&f, c.ev0, c.ev1));
        ms[0] = (double)f / (double)k;
        // each kernel alone, back to back (state keeps evolving: the kernels are idempotent in cost)
        SLP_HIP(hipEventRecord(c.ev0, c.stream));
        for (i64 it = 0; it < k; ++it) cp_primal(s, false);
        SLP_HIP(hipEventRecord(c.ev1, c.stream));
        SLP_HIP(hipEventSynchronize(c.ev1));
        SLP_HIP(hipEventElapsedTime(&f, c.ev0, c.ev1));
        ms[1] = (double)f / (double)k;
        SLP_HIP(hipEventRecord(c.ev0, c.stream));
        for (i64 it = 0; it < k; ++it) cp_dual(s);
        SLP_HIP(hipEventRecord(c.ev1, c.stream));
        SLP_HIP(hipEventSynchronize(c.ev1));
        SLP_HIP(hipEventElapsedTime(&f, c.ev0, c.ev1));
        ms[2] = (double)f / (double)k;
    })
}

}  // extern "C"
