"""ADMM with the reference's conjugate-gradient x-step, matrix-free on the GPU.

The reference ships ``lp_admm`` with a projected Gauss-Seidel x-step on the
explicit ``M = gamma_eq A^T A + gamma_ineq I``; its other x-steps are selected
by editing hard-coded flags (ADMM.py:66-71).  The conjugate-gradient branch
(ADMM.py:182-201, ``conjgrad`` with one step) only needs products ``M v``,
which the device evaluates as ``gamma_eq A^T (A v) + gamma_ineq v``.  It is
the ADMM that exists on the BASELINE-sized problems, where ``M`` would be
dense (hard part 2).  Kernels: csrc/slp_admm_cg.hip.
"""
import time

import numpy as np

from . import _lib
from ._lib import ORDER_AUTO
from .tools import convert_to_standard_form_with_bounds, precondition_constraints


class _CGBase:
    def close(self):
        if getattr(self, "_h", None):
            self._l.slp_admm_cg_destroy(self._h)
            self._h = None

    __del__ = close

    def set_reuse(self, reuse):
        """How many matrix products an iteration spends (same mathematics, fp64 rounding differences only):
        0 -- ten, as the reference writes the iteration;
        1 -- eight: the CG residual ``y - M(x + t d)`` is formed from ``M x`` and ``M d`` of the line search;
        2 -- six: additionally ``A^T (g_eq A x + lambda_eq)`` is one product instead of two;
        3 -- five: additionally ``A d`` of the new direction ``d = t d_old + a r`` comes from ``A d_old`` and ``A r``
             (taken as a product again every 64 iterations);
        4 -- four: additionally ``M d = step M d_old + a M r`` (``M r`` is computed for the CG step anyway), so the
             ``A^T`` pass of the line search carries one vector too (refreshed with ``A d``)."""
        reuse = int(reuse)
        _lib.check(self._l.slp_admm_cg_set_reuse(self._h, reuse))
        self.reuse = reuse

    def iterate(self, k):
        _lib.check(self._l.slp_admm_cg_iterate(self._h, int(k)))

    def xstep(self):
        _lib.check(self._l.slp_admm_cg_xstep(self._h))

    def multiplier_step(self):
        _lib.check(self._l.slp_admm_cg_multiplier_step(self._h))

    def report(self):
        out = np.zeros(4)
        _lib.check(self._l.slp_admm_cg_report(self._h, _lib.ptr(out)))
        return out

    def x(self, count=None):
        count = self.N if count is None else int(count)
        out = np.empty(count)
        _lib.check(self._l.slp_admm_cg_get_x(self._h, _lib.ptr(out), count))
        return out


class ADMMCGState(_CGBase):
    """Standard-form problem handed over from the host (any mix of equalities / inequalities)."""

    def __init__(self, a, b, c, lb, ub, x0, gamma_eq, gamma_ineq, order=ORDER_AUTO):
        self._l = _lib.lib()
        self.N, self.m = a.shape[1], a.shape[0]
        b, c, lb, ub, x0 = (_lib.f64(v) for v in (b, c, lb, ub, x0))
        self._h = _lib.check_handle(self._l.slp_admm_cg_create(
            self.N, self.m, _lib.ptr(a.indptr), _lib.ptr(a.indices), _lib.ptr(a.data), _lib.ptr(b), _lib.ptr(c),
            _lib.ptr(lb), _lib.ptr(ub), _lib.ptr(x0), float(gamma_eq), float(gamma_ineq), int(order)))


class DeviceADMM(_CGBase):
    """LP over a DeviceMatrix: the first ``m_eq`` rows are equalities ``a_i x = b_i``, the others inequalities
    ``a_i x <= b_i``.  Setup transforms run on the device and scale the matrix IN PLACE (the DeviceMatrix then
    holds the row-normalised values)."""

    def __init__(self, a, b_upper, c, lb, ub, gamma_eq=2.0, gamma_ineq=3.0, order=ORDER_AUTO, reuse=4, m_eq=0, b_lower=None):
        self._l = _lib.lib()
        self.a = a
        self.n = a.shape[1]
        self.N = a.shape[1] + a.shape[0]
        self.c = _lib.f64(c)
        b_upper, lb, ub = _lib.f64(b_upper), _lib.f64(lb), _lib.f64(ub)
        b_lower = None if b_lower is None else _lib.f64(b_lower)  # two-sided rows b_lower <= a_i x <= b_upper
        assert b_lower is None or b_lower.size == b_upper.size
        self._h = _lib.check_handle(self._l.slp_admm_cg_create_on_two_sided(
            a._h, int(m_eq), None if b_lower is None else _lib.ptr(b_lower), _lib.ptr(b_upper), _lib.ptr(self.c), _lib.ptr(lb),
            _lib.ptr(ub), float(gamma_eq), float(gamma_ineq), int(order)))
        self.set_reuse(reuse)

    def objective(self):
        return float(self.c.dot(self.x(self.n)))

    def two_vector_passes(self):
        """A [x, dir] and the two A^T products of the line search share one sweep over the matrix each
        when both orientations run on the strip kernel."""
        l = self._l
        return bool(self.reuse and l.slp_matrix_spmv_kernel(self.a._h, 0) >= 1 and l.slp_matrix_spmv_kernel(self.a._h, 1) >= 1)

    def matrix_products_per_iteration(self):
        return {0: 10, 1: 8, 2: 6, 3: 5, 4: 4}[self.reuse]

    def matrix_passes_per_iteration(self):
        if not self.two_vector_passes():
            return self.matrix_products_per_iteration()
        return {1: 5, 2: 4, 3: 4, 4: 4}[self.reuse]  # levels 3 / 4: one / both two-vector passes carry a single vector

    def describe(self):
        return ("ADMM, matrix-free conjugate-gradient x-step of the reference (ADMM.py:182-201), gamma_eq=2, gamma_ineq=3, "
                f"{self.matrix_products_per_iteration()} matrix products in {self.matrix_passes_per_iteration()} passes per iteration "
                f"(reuse level {self.reuse})")


def lp_admm_cg(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=None, gamma_eq=2, gamma_ineq=3, nb_iter=100,
               callback_func=None, max_time=None, use_preconditioning=True, nb_iter_plot=10, order=ORDER_AUTO,
               reuse=False):
    """``lp_admm`` of the reference with its ``use_cg`` flags; same signature, callback and return value."""
    c = _lib.f64(c)
    n = c.size
    if x0 is None:
        x0 = np.zeros(n)
    if a_eq is not None:
        a_eq, beq = precondition_constraints(a_eq, beq, alpha=2)
    if a_ineq is not None:
        a_ineq, b_lower, b_upper = precondition_constraints(a_ineq, b_lower, b_upper, alpha=2)
    c2, a, b, lb2, ub2, x_init = convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0)
    if use_preconditioning:
        a, b = precondition_constraints(a, b, alpha=2)
    # under a communicator (parallel.init_comm_from_env): every rank ran the (cheap, deterministic) transforms above on the
    # whole LP and hands over only its block of the standard-form rows; the N unknowns are replicated
    from .parallel import collective_elapsed, local_rows

    r0, r1, _ = local_rows(a.indptr)
    if (r0, r1) != (0, a.shape[0]):
        from .tools import CsrArrays

        k0, k1 = int(a.indptr[r0]), int(a.indptr[r1])
        a = CsrArrays(np.asarray(a.indptr[r0:r1 + 1]) - k0, a.indices[k0:k1], a.data[k0:k1], (r1 - r0, a.shape[1]))
        b = np.ascontiguousarray(b[r0:r1])
    state = ADMMCGState(a, b, c2, lb2, ub2, x_init, gamma_eq, gamma_ineq, order)
    state.set_reuse(reuse)
    try:
        start = time.perf_counter()
        i = 0
        while i <= nb_iter:
            if i % nb_iter_plot == 0:
                state.xstep()
                elapsed = time.perf_counter() - start
                # (under a communicator the decision is the same on every rank: the max over the ranks' clocks)
                if max_time is not None and collective_elapsed(elapsed) > max_time:
                    break
                energy1, max_violated_equality, max_violated_inequality = state.report()[:3]
                if callback_func is not None:
                    callback_func(i, state.x(n), energy1, energy1, elapsed, max_violated_equality, max_violated_inequality)
                state.multiplier_step()
                i += 1
            else:
                k = min(nb_iter_plot - i % nb_iter_plot, nb_iter + 1 - i)
                state.iterate(k)
                i += k
        return state.x(n)
    finally:
        state.close()
