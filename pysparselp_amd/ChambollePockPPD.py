"""Chambolle-Pock LP solver: host driver over the HIP kernels.

Drop-in for ``pysparselp.ChambollePockPPD.chambolle_pock_ppd``
(reference ChambollePockPPD.py:36-346): same signature, same callback contract
``callback_func(niter, x, energy1, energy2, elapsed, max_violated_equality,
max_violated_inequality)`` every ``nb_iter_plot`` iterations (including
iteration 0), same return value ``(x[:n], best_integer_solution)``.

The host keeps only the control flow; the preconditioners, every SpMV / SpMV^T,
the projections and the report reductions run on the GPU
(pysparselp_amd/csrc/slp_cp.hip) through the C ABI of include/slp_hip.h.
"""
import time

import numpy as np

from . import _lib
from ._lib import ORDER_AUTO
from .parallel import collective_elapsed


def _take_rows(a, rows, sign):
    """CSR rows ``rows`` of ``a`` (entry order kept), values times ``sign``."""
    indptr, indices, data = _lib.csr_arrays(a)
    cnt = (indptr[1:] - indptr[:-1])[rows]
    ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(cnt, out=ptr[1:])
    src = np.repeat(indptr[rows] - ptr[:-1], cnt) + np.arange(ptr[-1])
    return ptr, indices[src], sign * data[src]


def one_sided_system(a_ineq, b_lower, b_upper):
    """``b_lower <= A x <= b_upper``  ->  ``K x <= b`` (reference :74-88).

    Returns raw CSR arrays ``(indptr, indices, data, nrows)`` and ``b``: rows
    with a finite upper bound first, then the negated rows with a finite lower
    bound; when no lower bound is finite the matrix is returned untouched.
    """
    indptr, indices, data = _lib.csr_arrays(a_ineq)
    if b_lower is None:
        return (indptr, indices, data, a_ineq.shape[0]), _lib.f64(b_upper)
    b_lower, b_upper = _lib.f64(b_lower), _lib.f64(b_upper)
    up = np.nonzero(b_upper != np.inf)[0]
    lo = np.nonzero(b_lower != -np.inf)[0]
    b = np.hstack((b_upper[up], -b_lower[lo]))
    if len(lo) > 0 and len(up) > 0:
        p1, j1, v1 = _take_rows(a_ineq, up, 1.0)
        p2, j2, v2 = _take_rows(a_ineq, lo, -1.0)
        mat = (np.concatenate((p1, p1[-1] + p2[1:])), np.concatenate((j1, j2)), np.concatenate((v1, v2)), len(up) + len(lo))
    elif len(lo) > 0:
        mat = (indptr, indices, -data, a_ineq.shape[0])
    else:
        mat = (indptr, indices, data, a_ineq.shape[0])
    return mat, b


class CPState:
    """Device-resident Chambolle-Pock state (thin RAII wrapper of ``slp_cp``)."""

    def __init__(self, c, a_eq, beq, ineq, b_ineq, lb, ub, x0, alpha, theta, order=ORDER_AUTO):
        self._l = _lib.lib()
        c, lb, ub = _lib.f64(c), _lib.f64(lb), _lib.f64(ub)
        self.n = c.size
        parts_ptr, parts_idx, parts_val, parts_b = [np.zeros(1, dtype=np.int64)], [], [], []
        self.m_eq = 0
        if a_eq is not None:
            p, j, v = _lib.csr_arrays(a_eq)
            self.m_eq = a_eq.shape[0]
            parts_ptr.append(p[1:])
            parts_idx.append(j)
            parts_val.append(v)
            parts_b.append(_lib.f64(beq))
        self.m_ineq = 0
        if ineq is not None:
            p, j, v, rows = ineq
            self.m_ineq = rows
            off = parts_ptr[-1][-1] if len(parts_ptr) > 1 else 0
            parts_ptr.append(off + p[1:])
            parts_idx.append(j)
            parts_val.append(v)
            parts_b.append(_lib.f64(b_ineq))
        indptr = np.ascontiguousarray(np.concatenate(parts_ptr), dtype=np.int64)
        indices = np.ascontiguousarray(np.concatenate(parts_idx) if parts_idx else np.zeros(0), dtype=np.int32)
        data = _lib.f64(np.concatenate(parts_val) if parts_val else np.zeros(0))
        b = _lib.f64(np.concatenate(parts_b) if parts_b else np.zeros(0))
        if indices.size and (indices.min() < 0 or indices.max() >= self.n):
            raise ValueError("constraint matrix has a column index outside [0, n)")
        # under a communicator (parallel.init_comm_from_env): every rank holds the LP and hands over only its block of the
        # stacked rows K = [A_eq; A_ineq]; x, c, T, lb, ub are replicated, one all-reduce of n doubles per iteration
        from .parallel import local_rows

        r0, r1, m_eq_local = local_rows(indptr, self.m_eq)
        if (r0, r1) != (0, indptr.size - 1):
            k0, k1 = int(indptr[r0]), int(indptr[r1])
            indptr = np.ascontiguousarray(indptr[r0:r1 + 1] - k0)
            indices, data, b = np.ascontiguousarray(indices[k0:k1]), np.ascontiguousarray(data[k0:k1]), np.ascontiguousarray(b[r0:r1])
            self.m_eq, self.m_ineq = m_eq_local, (r1 - r0) - m_eq_local
        x0 = _lib.f64(x0) if x0 is not None else None
        self._h = _lib.check_handle(self._l.slp_cp_create(
            self.n, self.m_eq, self.m_ineq, _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(data), _lib.ptr(b),
            _lib.ptr(c), _lib.ptr(lb), _lib.ptr(ub), _lib.ptr(x0), float(alpha), float(theta), int(order)))

    def close(self):
        if getattr(self, "_h", None):
            self._l.slp_cp_destroy(self._h)
            self._h = None

    __del__ = close

    def iterate(self, k):
        _lib.check(self._l.slp_cp_iterate(self._h, int(k)))

    def primal_step(self):
        _lib.check(self._l.slp_cp_primal_step(self._h))

    def dual_step(self):
        _lib.check(self._l.slp_cp_dual_step(self._h))

    def report(self):
        out = np.zeros(8)
        _lib.check(self._l.slp_cp_report(self._h, _lib.ptr(out)))
        return out

    def x(self):
        out = np.empty(self.n)
        _lib.check(self._l.slp_cp_get_x(self._h, _lib.ptr(out)))
        return out

    def y(self):
        out = np.empty(self.m_eq + self.m_ineq)
        _lib.check(self._l.slp_cp_get_y(self._h, _lib.ptr(out)))
        return out

    def preconditioners(self):
        t, s = np.empty(self.n), np.empty(self.m_eq + self.m_ineq)
        _lib.check(self._l.slp_cp_get_preconditioners(self._h, _lib.ptr(t), _lib.ptr(s)))
        return t, s

    def bench(self, k):
        ms = np.zeros(3)
        _lib.check(self._l.slp_cp_bench(self._h, int(k), _lib.ptr(ms)))
        return ms


def chambolle_pock_ppd(
    c,
    a_eq,
    beq,
    a_ineq,
    b_lower,
    b_upper,
    lb,
    ub,
    x0=None,
    alpha=1,
    theta=1,
    nb_max_iter=100,
    callback_func=None,
    max_time=None,
    save_problem=False,
    force_integer=False,
    nb_iter_plot=10,
    order=ORDER_AUTO,
):
    """minimise c.x  s.t.  a_eq x = beq,  b_lower <= a_ineq x <= b_upper,  lb <= x <= ub.

    ``order`` (extension) selects the dot-product summation order, see
    include/slp_hip.h; the default reproduces the reference's iterates bit for
    bit while rows are short (mean <= 16 stored entries) and switches to
    wavefront-parallel sums for long rows.
    """
    if save_problem or force_integer:
        # debugging pickle / rounding heuristic of the reference: outside the accelerated path
        raise NotImplementedError("save_problem / force_integer are not supported by pysparselp_amd")
    start = time.perf_counter()
    c = _lib.f64(c)
    lb, ub = _lib.f64(lb), _lib.f64(ub)
    n = c.size
    assert lb.size == n and ub.size == n
    if a_eq is not None and a_eq.shape[0] == 0:  # reference :70-72
        a_eq, beq = None, None
    if a_ineq is not None and a_ineq.shape[0] == 0:
        a_ineq = None
    if a_eq is None and a_ineq is None:  # reference :147-151: no constraints, a vertex of the box
        x = np.zeros_like(lb)
        x[c > 0] = lb[c > 0]
        x[c < 0] = ub[c < 0]
        return x
    for a in (a_eq, a_ineq):
        if a is not None:
            assert a.shape[1] == n
    ineq, b_ineq = (None, None)
    if a_ineq is not None:
        ineq, b_ineq = one_sided_system(a_ineq, b_lower, b_upper)
        assert b_ineq.size == ineq[3]

    state = CPState(c, a_eq, beq, ineq, b_ineq, lb, ub, x0, alpha, theta, order)
    best_integer_solution_energy = np.inf
    best_integer_solution = None
    try:
        niter = 0
        while niter < nb_max_iter:
            if niter % nb_iter_plot == 0:
                state.primal_step()
                elapsed = time.perf_counter() - start
                if (max_time is not None) and collective_elapsed(elapsed) > max_time:  # the same decision on every rank
                    break
                energy1, energy2, max_violated_equality, max_violated_inequality, max_eq_at_x = state.report()[:5]
                if a_ineq is None:
                    max_violated_inequality = 0  # the reference dereferences a_ineq here (:283) and fails
                x = None
                if max_eq_at_x == 0 and max_violated_inequality <= 0:  # :284-291 with force_integer=False
                    x = state.x()
                    energy_rounded = c.dot(x)
                    if energy_rounded < best_integer_solution_energy:
                        best_integer_solution_energy = energy_rounded
                        best_integer_solution = x
                if callback_func is not None:
                    if x is None:
                        x = state.x()
                    callback_func(niter, x, energy1, energy2, elapsed, max_violated_equality, max_violated_inequality)
                state.dual_step()
                niter += 1
            else:
                k = min(nb_iter_plot - niter % nb_iter_plot, nb_max_iter - niter)
                state.iterate(k)
                niter += k
        x = state.x()
    finally:
        state.close()
    if best_integer_solution is not None:
        best_integer_solution = best_integer_solution[:n]
    return x[:n], best_integer_solution
