"""Solvers over a device-resident (possibly row-partitioned) constraint matrix.

Used where the LP never exists on the host: the synthetic BASELINE problems
(1e6 x 2e6 and beyond) are generated in HBM (``problems.random_lp_on_device``)
and solved in place.  All rows are inequalities ``A x <= b_upper``.
"""
import numpy as np

from . import _lib
from ._lib import ORDER_AUTO


class DeviceCP:
    """Chambolle-Pock (reference ChambollePockPPD.py:195-343) on a DeviceMatrix."""

    def __init__(self, a, b_upper, c, lb, ub, alpha=1.0, theta=1.0, order=ORDER_AUTO, m_eq=0):
        self._l = _lib.lib()
        self.a = a
        self.n = a.shape[1]
        self.c = _lib.f64(c)
        b_upper, lb, ub = _lib.f64(b_upper), _lib.f64(lb), _lib.f64(ub)
        self._h = _lib.check_handle(self._l.slp_cp_create_on(a._h, int(m_eq), _lib.ptr(b_upper), _lib.ptr(self.c), _lib.ptr(lb),
                                                             _lib.ptr(ub), None, float(alpha), float(theta), int(order)))

    def close(self):
        if getattr(self, "_h", None):
            self._l.slp_cp_destroy(self._h)
            self._h = None

    __del__ = close

    def iterate(self, k):
        _lib.check(self._l.slp_cp_iterate(self._h, int(k)))

    def x(self):
        out = np.empty(self.n)
        _lib.check(self._l.slp_cp_get_x(self._h, _lib.ptr(out)))
        return out

    def objective(self):
        return float(self.c.dot(self.x()))

    def bench(self, k):
        ms = np.zeros(3)
        _lib.check(self._l.slp_cp_bench(self._h, int(k), _lib.ptr(ms)))
        return ms

    def matrix_passes_per_iteration(self):
        return 2  # one A^T y, one A z

    def describe(self):
        return "diagonally preconditioned Chambolle-Pock, alpha=1, theta=1"


def make_solver(method, a, b_upper, c, lb, ub, m_eq=0):
    """``b_upper[:m_eq]`` are equality right-hand sides (the first ``m_eq`` rows), the rest upper bounds."""
    if method == "chambolle_pock_ppd":
        return DeviceCP(a, b_upper, c, lb, ub, m_eq=m_eq)
    if method == "admm":
        from .admm_cg import DeviceADMM

        return DeviceADMM(a, b_upper, c, lb, ub, m_eq=m_eq)
    raise ValueError(method)
