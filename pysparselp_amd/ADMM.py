"""ADMM LP solver: host driver over the HIP kernels.

Drop-in for ``pysparselp.ADMM.lp_admm`` (reference ADMM.py:47-269) as shipped,
i.e. with the x-step solved by ONE sweep of box-projected Gauss-Seidel on the
explicit ``M = gamma_eq A^T A + gamma_ineq I`` (flags at ADMM.py:66-71).  Same
signature, callback contract and return value (the first ``n`` entries of the
standard-form iterate).

The two constraint blocks are uploaded once, as the caller holds them; the whole
setup chain of ADMM.py:73-101 -- row normalisation of each block, slack standard
form, row normalisation of the stacked system, ``M = gamma_eq A^T A + gamma_ineq I``
and ``A^T b`` -- runs on the device (``slp_admm_create_lp``, csrc/slp_spgemm.hip)
with scipy's entry orders and accumulation orders, so the state is bit-identical
to the reference's.  (``SLP_HOST_SETUP=1`` prepares the same arrays with the numpy
restatement in tools.py instead and uploads them; the tests compare the two.)
Only the level schedule of the Gauss-Seidel sweep is planned on the host, from
one download of ``M``.  The loop -- right-hand side with ``A^T lambda``, the
level-scheduled Gauss-Seidel sweep, the multiplier update with ``A x`` and the
report reductions -- runs on the GPU (csrc/slp_admm.hip).
"""
import os
import time

import numpy as np

from . import _lib
from ._lib import ORDER_AUTO
from .tools import convert_to_standard_form_with_bounds, normal_matrix, precondition_constraints


class ADMMState:
    """Device-resident ADMM state (thin RAII wrapper of ``slp_admm``)."""

    def __init__(self, a, b, c, lb, ub, x0, m, gamma_eq, gamma_ineq, order=ORDER_AUTO):
        """``m``: the explicit ``M`` (CSR arrays) or ``None`` to have it formed on the device from ``a``."""
        self._l = _lib.lib()
        self.N = a.shape[1]
        self.m = a.shape[0]
        b, c, lb, ub, x0 = (_lib.f64(v) for v in (b, c, lb, ub, x0))
        assert b.size == self.m and c.size == self.N and lb.size == self.N and ub.size == self.N and x0.size == self.N
        self._h = _lib.check_handle(self._l.slp_admm_create(
            self.N, self.m, _lib.ptr(a.indptr), _lib.ptr(a.indices), _lib.ptr(a.data), _lib.ptr(b), _lib.ptr(c),
            _lib.ptr(lb), _lib.ptr(ub), _lib.ptr(x0), *((None, None, None) if m is None else (_lib.ptr(m.indptr), _lib.ptr(m.indices), _lib.ptr(m.data))),
            float(gamma_eq), float(gamma_ineq), int(order)))

    @classmethod
    def from_lp(cls, c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0, gamma_eq, gamma_ineq, use_preconditioning=True,
                order=ORDER_AUTO):
        """The solver state from the LP as ``lp_admm`` receives it; all setup transforms run on the device."""
        from .tools import CsrArrays

        self = cls.__new__(cls)
        self._l = _lib.lib()
        a_eq, a_ineq = CsrArrays.from_any(a_eq), CsrArrays.from_any(a_ineq)
        if a_ineq is None:  # what the reference does on this input (tools.py:92-127)
            raise UnboundLocalError("local variable 'a_eq2' referenced before assignment (no inequality constraints)")
        n = a_ineq.shape[1]
        c, lb, ub = _lib.f64(c), _lib.f64(lb), _lib.f64(ub)
        m_eq = a_eq.shape[0] if a_eq is not None else 0
        m_ineq = a_ineq.shape[0]
        self.N, self.m = n + m_ineq, m_eq + m_ineq
        opt = lambda v: None if v is None else _lib.f64(v)  # noqa: E731
        beq, b_lower, b_upper, x0 = opt(beq), opt(b_lower), opt(b_upper), opt(x0)
        eq = (None, None, None) if a_eq is None else (_lib.ptr(a_eq.indptr), _lib.ptr(a_eq.indices), _lib.ptr(a_eq.data))
        self._h = _lib.check_handle(self._l.slp_admm_create_lp(
            n, m_eq, *eq, _lib.ptr(beq), m_ineq, _lib.ptr(a_ineq.indptr), _lib.ptr(a_ineq.indices), _lib.ptr(a_ineq.data),
            _lib.ptr(b_lower), _lib.ptr(b_upper), _lib.ptr(c), _lib.ptr(lb), _lib.ptr(ub), _lib.ptr(x0), float(gamma_eq),
            float(gamma_ineq), int(bool(use_preconditioning)), int(order)))
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._l.slp_admm_destroy(self._h)
            self._h = None

    __del__ = close

    def set_xstep(self, mode):
        _lib.check(self._l.slp_admm_set_xstep(self._h, int(mode)))

    def iterate(self, k):
        _lib.check(self._l.slp_admm_iterate(self._h, int(k)))

    def sweep_step(self):
        _lib.check(self._l.slp_admm_sweep_step(self._h))

    def multiplier_step(self):
        _lib.check(self._l.slp_admm_multiplier_step(self._h))

    def report(self):
        out = np.zeros(4)
        _lib.check(self._l.slp_admm_report(self._h, _lib.ptr(out)))
        return out

    def x(self, count=None):
        count = self.N if count is None else int(count)
        out = np.empty(count)
        _lib.check(self._l.slp_admm_get_x(self._h, _lib.ptr(out), count))
        return out

    def lam(self):
        out = np.empty(self.m)
        _lib.check(self._l.slp_admm_get_lambda(self._h, _lib.ptr(out)))
        return out

    def num_levels(self):
        return int(self._l.slp_admm_num_levels(self._h))

    def num_bands(self):
        """Workgroups sharing the runs of narrow levels of M's Gauss-Seidel plan (0: one workgroup per run)."""
        return int(self._l.slp_admm_num_bands(self._h))

    def bench(self, k):
        ms = np.zeros(1)
        _lib.check(self._l.slp_admm_bench(self._h, int(k), _lib.ptr(ms)))
        return float(ms[0])


def lp_admm(
    c,
    a_eq,
    beq,
    a_ineq,
    b_lower,
    b_upper,
    lb,
    ub,
    x0=None,
    gamma_eq=2,
    gamma_ineq=3,
    nb_iter=100,
    callback_func=None,
    max_time=None,
    use_preconditioning=True,
    nb_iter_plot=10,
    order=ORDER_AUTO,
    xstep="gauss_seidel",
):
    """minimise c.x  s.t.  a_eq x = beq,  b_lower <= a_ineq x <= b_upper,  lb <= x <= ub.

    ``xstep`` (extension): ``"gauss_seidel"`` is the reference as shipped;
    ``"cg"`` is the reference's conjugate-gradient branch (flags at
    ADMM.py:66-71), run matrix-free -- see ``admm_cg.py``;
    ``"gauss_seidel_unbounded"`` is its plain Gauss-Seidel + over-relaxation
    branch (ADMM.py:164-181).
    """
    if xstep == "cg":
        from .admm_cg import lp_admm_cg

        return lp_admm_cg(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=x0, gamma_eq=gamma_eq, gamma_ineq=gamma_ineq,
                          nb_iter=nb_iter, callback_func=callback_func, max_time=max_time,
                          use_preconditioning=use_preconditioning, nb_iter_plot=nb_iter_plot, order=order)
    if xstep not in ("gauss_seidel", "gauss_seidel_unbounded"):
        raise ValueError(f"unknown xstep {xstep!r}")
    c = _lib.f64(c)
    n = c.size
    if os.environ.get("SLP_HOST_SETUP") == "1":
        # the numpy restatement of the setup chain (tools.py), then one upload of the finished arrays
        if x0 is None:
            x0 = np.zeros(n)
        if a_eq is not None:
            a_eq, beq = precondition_constraints(a_eq, beq, alpha=2)
        if a_ineq is not None:
            a_ineq, b_lower, b_upper = precondition_constraints(a_ineq, b_lower, b_upper, alpha=2)
        c2, a, b, lb2, ub2, x_init = convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0)
        if use_preconditioning:
            a, b = precondition_constraints(a, b, alpha=2)
        m_mat = normal_matrix(a, gamma_eq, gamma_ineq) if os.environ.get("SLP_HOST_SPGEMM") == "1" else None
        state = ADMMState(a, b, c2, lb2, ub2, x_init, m_mat, gamma_eq, gamma_ineq, order)
    else:
        # ADMM.py:76-101 on the device: blocks uploaded once, nothing else crosses PCIe but M's download for the level plan
        state = ADMMState.from_lp(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0, gamma_eq, gamma_ineq, use_preconditioning, order)
    if xstep == "gauss_seidel_unbounded":  # the reference's use_unbounded_gauss_siedel flags (ADMM.py:164-181)
        state.set_xstep(1)
    try:
        start = time.perf_counter()
        i = 0
        while i <= nb_iter:  # ADMM.py:143: nb_iter + 1 sweeps
            if i % nb_iter_plot == 0:
                state.sweep_step()
                elapsed = time.perf_counter() - start
                if max_time is not None and elapsed > max_time:
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
