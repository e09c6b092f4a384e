"""Solvers over a device-resident (possibly row-partitioned) constraint matrix.

Used where the LP never exists on the host: the synthetic BASELINE problems
(1e6 x 2e6 and beyond) are generated in HBM (``problems.random_lp_on_device``)
and solved in place.  All rows are inequalities ``A x <= b_upper``.
"""
import numpy as np

from . import _lib
from ._lib import ORDER_AUTO


def one_sided_rows(m, m_eq, b_lower, b_upper):
    """Row plan of the one-sided stacking  b_lower <= A_ineq x <= b_upper  ->  K x <= b  (ChambollePockPPD.py:74-88):
    ``(rows, scale, b)`` with K's row r = ``scale[r] *`` row ``rows[r]`` -- the equality rows (the first ``m_eq``), the rows
    with a finite upper bound, then the negated rows with a finite lower bound -- or ``None`` when no lower bound is
    finite (the matrix is used as it is).  With no finite upper bound every inequality row is negated (:82-83)."""
    ineq = np.arange(m_eq, m)
    up = ineq[b_upper[m_eq:] != np.inf]
    lo = ineq[b_lower[m_eq:] != -np.inf]
    if len(lo) == 0:
        return None
    eq = np.arange(m_eq)
    if len(up) > 0:
        rows = np.concatenate((eq, up, lo))
        scale = np.concatenate((np.ones(m_eq + len(up)), -np.ones(len(lo))))
        b = np.concatenate((b_upper[:m_eq], b_upper[up], -b_lower[lo]))
    else:
        if len(lo) != m - m_eq:
            raise ValueError("rows without any finite bound next to lower-bounded rows: the reference's stacking is "
                             "inconsistent here (ChambollePockPPD.py:80-86); drop the unbounded rows first")
        rows = np.arange(m)
        scale = np.concatenate((np.ones(m_eq), -np.ones(m - m_eq)))
        b = np.concatenate((b_upper[:m_eq], -b_lower[lo]))
    return rows.astype(np.int64), scale, np.ascontiguousarray(b)


class DeviceCP:
    """Chambolle-Pock (reference ChambollePockPPD.py:195-343) on a DeviceMatrix."""

    def __init__(self, a, b_upper, c, lb, ub, alpha=1.0, theta=1.0, order=ORDER_AUTO, m_eq=0, b_lower=None, remove_fixed=False):
        """``remove_fixed=False`` is the reference's ``chambolle_pock_ppd`` function on the LP as given;
        ``remove_fixed=True`` adds what ``SparseLP.solve`` does before calling it (SparseLP.py:1244-1248): variables with
        ``ub == lb`` are dropped -- here on the device (column compaction + ``b - A shift``)."""
        self._l = _lib.lib()
        self.n_full = a.shape[1]
        c, b_upper, lb, ub = _lib.f64(c), _lib.f64(b_upper), _lib.f64(lb), _lib.f64(ub)
        b_lower = None if b_lower is None else _lib.f64(b_lower)
        self._stacked = self._reduced = None
        # remove_fixed_variables (SparseLP.py:632-674) on the device: column compaction + b - A shift
        self.free = (ub > lb) if remove_fixed else np.ones(lb.size, dtype=bool)
        self.shift = np.where(self.free, 0.0, lb)
        if not self.free.all():
            self._reduced, a_shift = a.remove_columns(self.free, self.shift)
            a = self._reduced
            b_upper = b_upper - a_shift          # equality rows: b_eq - A_eq shift ; inequality rows: b_upper - A_ineq shift
            if b_lower is not None:
                b_lower = b_lower - a_shift
            c, lb, ub = c[self.free], lb[self.free], ub[self.free]
        self.n = a.shape[1]
        self.c = np.ascontiguousarray(c)
        if b_lower is not None:
            a, b_upper = self._one_sided(a, int(m_eq), b_lower, b_upper)
        self.a = a
        self._h = _lib.check_handle(self._l.slp_cp_create_on(a._h, int(m_eq), _lib.ptr(b_upper), _lib.ptr(self.c), _lib.ptr(lb),
                                                             _lib.ptr(ub), None, float(alpha), float(theta), int(order)))

    def _one_sided(self, a, m_eq, b_lower, b_upper):
        plan = one_sided_rows(a.shape[0], m_eq, b_lower, b_upper)
        if plan is None:
            return a, b_upper
        rows, scale, b = plan
        self._stacked = a.gather_rows(rows, scale)
        return self._stacked, b

    def close(self):
        if getattr(self, "_h", None):
            self._l.slp_cp_destroy(self._h)
            self._h = None
        for name in ("_stacked", "_reduced"):
            if getattr(self, name, None) is not None:
                getattr(self, name).close()
                setattr(self, name, None)

    __del__ = close

    def iterate(self, k):
        _lib.check(self._l.slp_cp_iterate(self._h, int(k)))

    def primal_step(self):
        """First half of an iteration (ChambollePockPPD.py:198-240), keeping ``d`` for the report."""
        _lib.check(self._l.slp_cp_primal_step(self._h))

    def dual_step(self):
        _lib.check(self._l.slp_cp_dual_step(self._h))

    def report(self):
        """``(energy1, energy2, max_violated_equality [max |A_e z - b_e|, :235,275], max_violated_inequality [max (A_i x - b_i),
        :283], max |A_e x - b_e|)`` between the two halves of an iteration, as the reference's periodic report computes them
        (ChambollePockPPD.py:242-329).  Runs on the strip copies when the matrix has them -- also after ``release_csr``."""
        out = np.zeros(5)
        _lib.check(self._l.slp_cp_report(self._h, _lib.ptr(out)))
        return out

    def split_form(self):
        """How ``d = (c + y_eq * a_eq) + y_ineq * a_ineq`` (ChambollePockPPD.py:206,216) is formed when the LP has both kinds of
        rows and runs on strip copies: 1 -- two products over copies of ``A_e^T`` and ``A_i^T`` (the chunks of a chunked matrix cut at
        ``m_eq``, or copies of the two row ranges built for this solver); 2 -- two products over the copy of the whole ``K^T`` with
        the other kind of rows masked out of ``y``; 0 -- one kind of rows only, or the CSR / ELL walk that forms both sums in one pass."""
        return int(self._l.slp_cp_split_form(self._h))

    def x_reduced(self):
        """The iterate over the free variables (what the solver works on)."""
        out = np.empty(self.n)
        _lib.check(self._l.slp_cp_get_x(self._h, _lib.ptr(out)))
        return out

    def x(self):
        """The iterate over ALL variables: free ones from the solver, fixed ones at their value (``m_change x + shift``)."""
        if self.free.all():
            return self.x_reduced()
        out = self.shift.copy()
        out[self.free] = self.x_reduced()
        return out

    def objective(self):
        """``c . x`` over the free variables (the reduced problem's objective, as the reference's callback reports it)."""
        return float(self.c.dot(self.x_reduced()))

    def bench(self, k):
        ms = np.zeros(3)
        _lib.check(self._l.slp_cp_bench(self._h, int(k), _lib.ptr(ms)))
        return ms

    def matrix_passes_per_iteration(self):
        return 2  # one A^T y, one A z

    def describe(self):
        return "diagonally preconditioned Chambolle-Pock, alpha=1, theta=1"


def make_solver(method, a, b_upper, c, lb, ub, m_eq=0, blocks_per_rank=1, jacobi=False):
    """``b_upper[:m_eq]`` are equality right-hand sides (the first ``m_eq`` rows), the rest upper bounds."""
    if method == "chambolle_pock_ppd":
        return DeviceCP(a, b_upper, c, lb, ub, m_eq=m_eq)
    if method == "admm":
        from .admm_cg import DeviceADMM

        return DeviceADMM(a, b_upper, c, lb, ub, m_eq=m_eq)
    if method == "admm_blocks":
        if blocks_per_rank > 1:
            rows = a.shape[0]
            cuts = [rows * g // blocks_per_rank for g in range(blocks_per_rank + 1)]
            return DeviceBlocksGroup(a, cuts, b_upper, c, lb, ub, m_eq=m_eq)
        return DeviceBlocks(a, b_upper, c, lb, ub, m_eq=m_eq, jacobi=jacobi)
    raise ValueError(method)


class DeviceBlocks:
    """Block-splitting ADMM (reference ADMMBlocks.py:45-352) with ONE block per rank: the rows of the DeviceMatrix ``a``
    (this rank's row block when ``slp_comm_init`` is active).  The first ``m_eq`` rows are equalities, the others
    ``b_lower <= a_i x <= b_upper``.  Per-block projections by conjugate gradients, no exchange inside them; one
    all-reduce of n doubles per iteration for the consensus."""

    def __init__(self, a, b_upper, c, lb, ub, gamma=0.7, m_eq=0, b_lower=None, cg_tol=1e-13, cg_max_steps=500, jacobi=False):
        self._l = _lib.lib()
        self.a = a
        self.n = a.shape[1]
        self.c = _lib.f64(c)
        b_upper, lb, ub = _lib.f64(b_upper), _lib.f64(lb), _lib.f64(ub)
        b_lower = None if b_lower is None else _lib.f64(b_lower)
        self._h = _lib.check_handle(self._l.slp_blocks_create_on(
            a._h, int(m_eq), None if b_lower is None else _lib.ptr(b_lower), _lib.ptr(b_upper), _lib.ptr(self.c), _lib.ptr(lb),
            _lib.ptr(ub), float(gamma)))
        _lib.check(self._l.slp_blocks_set_cg(self._h, float(cg_tol), int(cg_max_steps)))
        if jacobi:
            _lib.check(self._l.slp_blocks_set_precond(self._h, 1))

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

    def x(self):
        out = np.empty(self.n)
        _lib.check(self._l.slp_blocks_get_xp(self._h, _lib.ptr(out), self.n))
        return out

    def objective(self):
        return float(self.c.dot(self.x()))

    def cg_steps(self):
        """CG steps taken by this rank so far (local: safe to call on one rank only)."""
        return int(self._l.slp_blocks_cg_steps(self._h))

    def projection_residual(self):
        """``(|| rhs - S sol ||_2, || rhs ||_2)`` of the last block update's projection system, operator applied afresh."""
        out = np.zeros(2)
        _lib.check(self._l.slp_blocks_projection_residual(self._h, _lib.ptr(out)))
        return float(out[0]), float(out[1])

    def matrix_passes_per_iteration(self):
        return None  # 3 + 2 per conjugate-gradient step; see cg_steps()

    def describe(self):
        return "block-splitting ADMM (ADMMBlocks.py), one block per rank, matrix-free per-block projections (CG), gamma=0.7, alpha=1.95"


class DeviceBlocksGroup:
    """Block-splitting ADMM with SEVERAL row blocks on this rank (the ``blocks`` metadata of the reference,
    ADMMBlocks.py:93-95,178-243, at scale): block g = rows ``cuts[g] .. cuts[g + 1]`` of the DeviceMatrix ``a`` (cut on the
    device, ``slp_matrix_gather_rows``), each with its own copy of the variables it uses, its own multipliers and its own
    matrix-free projection; one all-reduce of n doubles per iteration for the whole group (``slp_blocks_group_iterate``).
    All rows are inequalities ``b_lower <= a_i x <= b_upper`` (equality rows: give ``m_eq`` leading rows of block 0)."""

    def __init__(self, a, cuts, b_upper, c, lb, ub, gamma=0.7, b_lower=None, m_eq=0, cg_tol=1e-13, cg_max_steps=500):
        cuts = [int(v) for v in cuts]
        assert cuts[0] == 0 and cuts[-1] == a.shape[0] and all(x <= y for x, y in zip(cuts, cuts[1:]))
        mats = [a.gather_rows(np.arange(r0, r1, dtype=np.int64)) for r0, r1 in zip(cuts, cuts[1:])]
        self._link(mats, cuts, b_upper, c, lb, ub, gamma, b_lower, m_eq, cg_tol, cg_max_steps)

    def _link(self, mats, cuts, b_upper, c, lb, ub, gamma, b_lower, m_eq, cg_tol, cg_max_steps):
        import ctypes

        self._l = _lib.lib()
        self.n = mats[0].shape[1]
        self.c = _lib.f64(c)
        self._link_args = (cuts, b_upper, c, lb, ub, gamma, b_lower, m_eq, cg_tol, cg_max_steps)
        assert 0 <= m_eq <= cuts[1]
        b_upper = _lib.f64(b_upper)
        b_lower = None if b_lower is None else _lib.f64(b_lower)
        self.blocks, self._mats = [], list(mats)
        for g, (r0, r1) in enumerate(zip(cuts, cuts[1:])):
            self.blocks.append(DeviceBlocks(mats[g], b_upper[r0:r1], c, lb, ub, gamma=gamma, m_eq=m_eq if g == 0 else 0,
                                            b_lower=None if b_lower is None else b_lower[r0:r1], cg_tol=cg_tol, cg_max_steps=cg_max_steps))
        self._handles = (ctypes.c_void_p * len(self.blocks))(*[blk._h for blk in self.blocks])
        _lib.check(self._l.slp_blocks_group_link(self._handles, len(self.blocks)))

    @classmethod
    def from_generator(cls, n, m, density, seed, cuts, row_offset=0, chunks_per_block=1, gamma=0.7, cg_tol=1e-13, cg_max_steps=500):
        """The group over rows ``row_offset + cuts[0] .. row_offset + cuts[-1]`` of the synthetic ``m x n`` LP of
        ``problems.random_lp_on_device`` WITHOUT a resident matrix: every block is generated from its own row range, converted
        into its product copies for both orientations and its CSR released before the next block is generated
        (``ChunkedDeviceMatrix`` of ``chunks_per_block`` chunks) -- BASELINE config 5 (5e7 variables, eight blocks of 5e5 rows at
        1e-4: 2e10 stored entries, 240 GB of CSR per orientation) is resident on ONE GPU as 8 x 26 GB of tall cells.  The solvers'
        vectors are allocated only after ALL blocks stand, so that a conversion's temporaries never sit beside them.
        Returns ``(group, feasible_x, c, lb, ub, b_upper)`` (``b_upper`` over this rank's rows, block after block)."""
        from .problems import random_lp_on_device

        cuts = [int(v) for v in cuts]
        assert cuts[0] == 0 and all(x < y for x, y in zip(cuts, cuts[1:])) and row_offset + cuts[-1] <= m
        b = np.empty(cuts[-1])
        xf = c = lb = ub = None
        import time

        self = cls.__new__(cls)
        self.blocks, self._mats = [], []
        t0 = time.perf_counter()
        try:
            for g, (r0, r1) in enumerate(zip(cuts, cuts[1:])):
                got = random_lp_on_device(n, m, density, seed=seed, row_offset=row_offset + r0, rows=r1 - r0, chunks=chunks_per_block,
                                          chunked=True, columns=(g == 0))
                self._mats.append(got[0])
                if g == 0:
                    xf, c, lb, ub = got[1:5]
                b[r0:r1] = got[5]
            _lib.check(_lib.lib().slp_trim())   # the conversions' cached temporaries go back before the solvers' vectors are taken
            _lib.check(_lib.lib().slp_synchronize())
            self.seconds_generating = time.perf_counter() - t0   # generator + conversions (bench.py: setup_breakdown)
            self._link(self._mats, cuts, b, c, lb, ub, gamma, None, 0, cg_tol, cg_max_steps)
        except BaseException:
            self.close()   # (blocks first: they borrow the matrices)
            raise
        return self, xf, c, lb, ub, b

    @property
    def nnz(self):
        return sum(mat.nnz for mat in self._mats)

    def restart(self):
        """Fresh solver state (x0 = 0, zero multipliers) over the SAME block matrices: the copies of a 2e10-entry LP are built
        once, a second run (determinism checks, another tolerance) starts from here."""
        for blk in self.blocks:
            blk.close()
        self.blocks = []
        self._link(self._mats, *self._link_args)

    def projection_residuals(self):
        """``(|| rhs - S sol ||, || rhs ||)`` of every block's last projection, operator applied afresh."""
        return [blk.projection_residual() for blk in self.blocks]

    def close(self):
        for blk in getattr(self, "blocks", []):
            blk.close()
        for mat in getattr(self, "_mats", []):
            mat.close()
        self.blocks, self._mats = [], []

    __del__ = close

    def iterate(self, k):
        _lib.check(self._l.slp_blocks_group_iterate(self._handles, len(self.blocks), int(k)))

    def x(self):
        return self.blocks[0].x()

    def objective(self):
        return float(self.c.dot(self.x()))

    def cg_steps(self):
        return sum(blk.cg_steps() for blk in self.blocks)

    def matrix_passes_per_iteration(self):
        return None

    def describe(self):
        return f"block-splitting ADMM (ADMMBlocks.py), {len(self.blocks)} row blocks on this rank, matrix-free per-block projections (CG)"
