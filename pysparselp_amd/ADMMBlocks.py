"""Block-splitting ADMM: host driver over the HIP kernels of ``csrc/slp_blocks.hip``.

Drop-in for ``pysparselp.ADMMBlocks.lp_admm_block_decomposition`` (reference
ADMMBlocks.py:45-352): same signature, callback contract (called AFTER the multiplier
update of every ``nb_iter_plot``-th iteration with the clamped consensus variable,
:309-350) and return value.  One copy of the variables per block of constraints
(``a.blocks``, recorded by the modelling layer: SparseLP.py:93-95), consensus
average + clamp, scaled multipliers, over-relaxation 1.95.

Where the reference factorises one KKT matrix per block with a sparse LU (:178-243),
this build solves all per-block projections together, matrix-free, with conjugate
gradients on the device (see slp_blocks.hip): results agree with the LU form to the
CG tolerance (``cg_tol``, relative residual, default 1e-13), not bit for bit.
``use_lu`` / ``use_preconditioning`` are accepted and ignored like in the reference
(its body never reads them).  ``max_time=None`` means no limit (the reference
compares ``elapsed > None``, which raises in Python 3).
"""
import time

import numpy as np

from . import _lib
from .tools import CsrArrays, convert_to_standard_form_with_bounds


def split_by_blocks(a, blocks):
    """The copies layout of ADMMBlocks.py:178-193.

    Returns ``(a_split, owner, copy_ptr, copy_idx)``: ``a_split`` is ``a`` with one column per
    (block, variable used by the block) pair -- block g's columns are the sorted variables with a
    nonzero absolute column sum inside its rows (:183-185) --, ``owner[p]`` the variable of copy p,
    and the copies of every variable in block order as a CSR list.
    """
    m, n = a.shape
    indptr, indices, data = a.indptr, a.indices, a.data
    out_cols = np.zeros(indices.size, dtype=np.int64)
    keep = np.zeros(indices.size, dtype=bool)
    owner, offset = [], 0
    covered = np.zeros(m, dtype=bool)
    for lo, hi in blocks:
        if covered[lo:hi + 1].any():
            raise ValueError("constraint blocks overlap")
        covered[lo:hi + 1] = True
        s, e = indptr[lo], indptr[hi + 1]
        cols = indices[s:e]
        colsum = np.bincount(cols, weights=np.abs(data[s:e]), minlength=n)
        ids = np.nonzero(colsum)[0]
        local = np.full(n, -1, dtype=np.int64)
        local[ids] = np.arange(ids.size)
        out_cols[s:e] = offset + local[cols]
        keep[s:e] = local[cols] >= 0   # entries of all-zero columns are dropped with their column
        owner.append(ids)
        offset += ids.size
    if not covered.all():
        raise ValueError("every constraint row must belong to a block")
    owner = np.concatenate(owner).astype(np.int32) if owner else np.zeros(0, dtype=np.int32)
    rows = a.row_of_entry()[keep]
    new_ptr = np.concatenate(([0], np.cumsum(np.bincount(rows, minlength=m)))).astype(np.int64)
    a_split = CsrArrays(new_ptr, out_cols[keep].astype(np.int32), data[keep], (m, int(offset)))
    order = np.argsort(owner, kind="stable")  # copies of a variable in increasing block order
    copy_ptr = np.concatenate(([0], np.cumsum(np.bincount(owner, minlength=n)))).astype(np.int64)
    return a_split, owner, copy_ptr, order.astype(np.int32)


class BlocksState:
    """Device-resident state (thin RAII wrapper of ``slp_blocks``)."""

    def __init__(self, a_split, b, c, lb, ub, xp0, owner, copy_ptr, copy_idx, gamma, cg_tol=1e-13, cg_max_steps=500):
        self._l = _lib.lib()
        self.N = c.size
        b, c, lb, ub, xp0 = (_lib.f64(v) for v in (b, c, lb, ub, xp0))
        self._h = _lib.check_handle(self._l.slp_blocks_create(
            a_split.shape[1], a_split.shape[0], self.N, _lib.ptr(a_split.indptr), _lib.ptr(a_split.indices), _lib.ptr(a_split.data),
            _lib.ptr(b), _lib.ptr(c), _lib.ptr(lb), _lib.ptr(ub), _lib.ptr(xp0), _lib.ptr(owner), _lib.ptr(copy_ptr),
            _lib.ptr(copy_idx), float(gamma)))
        _lib.check(self._l.slp_blocks_set_cg(self._h, float(cg_tol), int(cg_max_steps)))

    def close(self):
        if getattr(self, "_h", None):
            self._l.slp_blocks_destroy(self._h)
            self._h = None

    __del__ = close

    def iterate(self, k):
        _lib.check(self._l.slp_blocks_iterate(self._h, int(k)))

    def report(self):
        out = np.zeros(2)
        _lib.check(self._l.slp_blocks_report(self._h, _lib.ptr(out)))
        return out

    def xp(self, count=None):
        count = self.N if count is None else int(count)
        out = np.empty(count)
        _lib.check(self._l.slp_blocks_get_xp(self._h, _lib.ptr(out), count))
        return out


def lp_admm_block_decomposition(
    c,
    a_eq,
    beq,
    a_ineq,
    b_lower,
    b_upper,
    lb,
    ub,
    x0=None,
    gamma_ineq=0.7,
    nb_iter=100,
    callback_func=None,
    max_time=None,
    use_preconditioning=True,
    use_lu=True,
    nb_iter_plot=10,
    cg_tol=1e-13,
    cg_max_steps=500,
):
    """minimise c.x  s.t.  a_eq x = beq,  b_lower <= a_ineq x <= b_upper,  lb <= x <= ub  (block-splitting ADMM)."""
    c = _lib.f64(c)
    n = c.size
    if x0 is None:
        x0 = np.zeros(n)
    c2, a, b, lb2, ub2, x_init = convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0)  # :79-81
    blocks = list(a.blocks)
    if not blocks:
        raise ValueError("the constraint matrices carry no `blocks` attribute (SparseLP.py:93-95 records it)")
    xp0 = np.minimum(np.maximum(x_init, lb2), ub2)  # :84-86
    a_split, owner, copy_ptr, copy_idx = split_by_blocks(a, blocks)
    state = BlocksState(a_split, b, c2, lb2, ub2, xp0, owner, copy_ptr, copy_idx, gamma_ineq, cg_tol, cg_max_steps)
    try:
        start = time.perf_counter()
        i = 0
        while i <= nb_iter:  # :264: nb_iter + 1 iterations, report after the iterations with i % nb_iter_plot == 0
            k = 1 if i % nb_iter_plot == 0 else min(nb_iter_plot - i % nb_iter_plot, nb_iter + 1 - i)
            reports = (i + k - 1) % nb_iter_plot == 0
            state.iterate(k)
            i += k
            if reports:
                elapsed = time.perf_counter() - start
                if max_time is not None and elapsed > max_time:
                    break
                energy = state.report()[0]
                if callback_func is not None:
                    callback_func(i - 1, state.xp(n), energy, energy, elapsed, 0, 0)
        return state.xp(n)
    finally:
        state.close()
