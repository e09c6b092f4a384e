"""``SparseLP``: the modelling front end and the ``solve(method=...)`` dispatch.

Keeps the public surface of ``pysparselp.SparseLP.SparseLP`` that the
first-order solvers need (reference SparseLP.py:162-1383): the LP is held as

    minimise costsvector . x
    s.t.     a_equalities x = b_equalities
             b_lower <= a_inequalities x <= b_upper
             lower_bounds <= x <= upper_bounds

with scipy CSR matrices, and ``solve`` (reference :990-1002, :1064-1093,
:1193-1208, :1243-1288, :1378-1383) runs one of the GPU solvers and fills the
same convergence-curve attributes.  The hot-path methods ``"admm"`` and
``"chambolle_pock_ppd"`` and the block-splitting ``"admm_blocks"`` exist here; the other solvers of the reference
(interior point, dual ascent, external solver bridges, rounding heuristics,
MPS export) are out of scope.
"""
import copy
import time

import numpy as np

import scipy.sparse

from .ADMM import lp_admm
from .ChambollePockPPD import chambolle_pock_ppd
from ._lib import ORDER_AUTO

solving_methods = ("chambolle_pock_ppd", "admm", "admm_blocks")

_SCALARS = (int, float, np.integer, np.floating)


def empty_csr_matrix(ncols=0):
    a = scipy.sparse.csr_matrix((0, ncols), dtype=np.float64)
    a.__dict__["blocks"] = []
    return a


def _append_rows(a, b):
    """Rows of ``b`` appended under ``a`` (entry order kept); records the row block like the reference."""
    b = scipy.sparse.csr_matrix(b)
    ncols = max(a.shape[1], b.shape[1])
    blocks = a.__dict__.get("blocks", [])
    blocks.append((a.shape[0], a.shape[0] + b.shape[0] - 1))
    out = scipy.sparse.csr_matrix(
        (np.concatenate((a.data, b.data.astype(np.float64))),
         np.concatenate((a.indices, b.indices)).astype(np.int32),
         np.concatenate((a.indptr[:-1], a.indptr[-1] + b.indptr)).astype(np.int32)),
        shape=(a.shape[0] + b.shape[0], ncols))
    out.__dict__["blocks"] = blocks
    return out


def _with_ncols(a, ncols):
    out = scipy.sparse.csr_matrix((a.data, a.indices, a.indptr), shape=(a.shape[0], ncols))
    out.__dict__["blocks"] = a.__dict__.get("blocks", [])
    return out


def crd_matrix(cols, vals):
    """CSR matrix with the same number of candidate entries in every row:
    ``m[i, cols[i, j]] = vals[i, j]``; zero values are not stored; a variable may
    appear only once per row."""
    cols, vals = np.broadcast_arrays(np.asarray(cols), np.asarray(vals, dtype=np.float64))
    assert cols.ndim == 2
    srt = np.sort(cols, axis=1)
    if srt.shape[1] > 1 and np.any(srt[:, 1:] == srt[:, :-1]):
        bad = np.nonzero(np.any(srt[:, 1:] == srt[:, :-1], axis=1))[0]
        raise ValueError(f"the same variable appears twice in {len(bad)} constraint(s): {bad}")
    keep = vals != 0
    indptr = np.concatenate(([0], np.cumsum(keep.sum(axis=1))))
    ncols = int(cols.max()) + 1 if cols.size else 0
    return scipy.sparse.csr_matrix((vals[keep], cols[keep], indptr), shape=(cols.shape[0], ncols))


ADMM_AUTO_M_ENTRIES = 2.0e8  # solve(method="admm", xstep="auto"): above this estimate of nnz(M) the matrix-free x-step is taken


class SparseLP:
    """Sparse linear program + first-order GPU solvers."""

    def __init__(self):
        self.nb_variables = 0
        self.variables_dict = dict()
        self.upper_bounds = np.empty(0, dtype=np.float64)
        self.lower_bounds = np.empty(0, dtype=np.float64)
        self.costsvector = np.empty(0, dtype=np.float64)
        self.is_integer = np.empty(0, dtype=bool)
        self.a_inequalities = empty_csr_matrix()
        self.b_lower = np.empty(0, dtype=np.float64)
        self.b_upper = np.empty(0, dtype=np.float64)
        self.a_equalities = empty_csr_matrix()
        self.b_equalities = np.empty(0, dtype=np.float64)
        self.solution = None

    # ------------------------------------------------------------------ variables
    def _bounds_as_arrays(self, shape, lower_bounds, upper_bounds):
        def expand(v, default):
            if v is None:
                return np.full(shape, default, dtype=np.float64)
            if isinstance(v, _SCALARS):
                return np.full(shape, float(v), dtype=np.float64)
            v = np.asarray(v, dtype=np.float64)
            assert v.shape == tuple(shape)
            return v

        return expand(lower_bounds, -np.inf), expand(upper_bounds, np.inf)

    def add_variables_array(self, shape, lower_bounds, upper_bounds, costs=0, name=None, is_integer=False):
        """Append ``prod(shape)`` variables; returns their indices as an array of that shape."""
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        shape = tuple(int(s) for s in shape)
        count = int(np.prod(shape))
        indices = np.arange(count).reshape(shape) + self.nb_variables
        self.nb_variables += count
        self.a_inequalities = _with_ncols(self.a_inequalities, self.nb_variables)
        self.a_equalities = _with_ncols(self.a_equalities, self.nb_variables)
        if isinstance(costs, _SCALARS):
            costs = np.full(shape, float(costs))
        costs = np.asarray(costs, dtype=np.float64)
        assert costs.shape == shape
        lo, hi = self._bounds_as_arrays(shape, lower_bounds, upper_bounds)
        self.lower_bounds = np.append(self.lower_bounds, lo.ravel())
        self.upper_bounds = np.append(self.upper_bounds, hi.ravel())
        self.costsvector = np.append(self.costsvector, costs.ravel())
        self.is_integer = np.append(self.is_integer, np.full(count, is_integer, dtype=bool))
        if name:
            self.variables_dict[name] = indices
        return indices

    def get_variables_indices(self, name):
        return self.variables_dict[name]

    def get_variables_bounds(self):
        return None, self.lower_bounds, self.upper_bounds

    def set_costs_variables(self, indices, costs):
        assert np.shape(costs) == np.shape(indices)
        self.costsvector[np.ravel(indices)] = np.ravel(costs)

    def set_bounds_on_variables(self, indices, lower_bounds, upper_bounds):
        lo, hi = self._bounds_as_arrays(np.shape(indices), lower_bounds, upper_bounds)
        self.lower_bounds[np.ravel(indices)] = lo.ravel()
        self.upper_bounds[np.ravel(indices)] = hi.ravel()

    # ---------------------------------------------------------------- constraints
    def nb_equality_constraints(self):
        return self.a_equalities.shape[0]

    def nb_inequality_constraints(self):
        return self.a_inequalities.shape[0]

    def add_equality_constraints_sparse(self, a, b):
        self.a_equalities = _append_rows(self.a_equalities, a)
        self.b_equalities = np.append(self.b_equalities, b)

    def add_inequality_constraints_sparse(self, a, lower_bounds=None, upper_bounds=None):
        """Add ``lower_bounds <= a x <= upper_bounds``; a scalar ``lower == upper`` makes equalities."""
        rows = a.shape[0]
        if isinstance(lower_bounds, (int, float)) and not isinstance(lower_bounds, bool) and lower_bounds == upper_bounds:
            self.a_equalities = _append_rows(self.a_equalities, a)
            self.b_equalities = np.append(self.b_equalities, np.full(rows, float(lower_bounds)))
            return
        lo, hi = self._bounds_as_arrays((rows,), lower_bounds, upper_bounds)
        self.a_inequalities = _append_rows(self.a_inequalities, a)
        self.b_lower = np.append(self.b_lower, lo)
        self.b_upper = np.append(self.b_upper, hi)

    def add_inequality_constraints(self, cols, vals, lower_bounds=None, upper_bounds=None):
        """``lower[i] <= sum_j vals[i, j] x[cols[i, j]] <= upper[i]`` for every row i."""
        self.add_inequality_constraints_sparse(crd_matrix(cols, vals), lower_bounds=lower_bounds, upper_bounds=upper_bounds)

    def add_equality_constraints(self, cols, vals, b):
        self.add_inequality_constraints(cols, vals, lower_bounds=b, upper_bounds=b)

    def add_soft_inequality_constraints(self, cols, vals, coef_penalization, lower_bounds=None, upper_bounds=None):
        """Penalised version: adds ``sum_i coef[i] * max(0, lower[i] - y_i, y_i - upper[i])`` to the objective,
        ``y_i = sum_j vals[i, j] x[cols[i, j]]``, through one auxiliary variable per row (reference :575-613).
        With an infinite penalty this is the hard constraint.  Returns the auxiliary variable indices (or None)."""
        coef = np.asarray(coef_penalization, dtype=np.float64)
        if np.all(coef == np.inf):
            self.add_inequality_constraints(cols, vals, lower_bounds=lower_bounds, upper_bounds=upper_bounds)
            return None
        if np.any(coef == np.inf):
            raise ValueError("mixing finite and infinite penalisation is not supported")
        cols, vals = np.broadcast_arrays(np.asarray(cols), np.asarray(vals, dtype=np.float64))
        assert upper_bounds is not None or lower_bounds is not None
        costs = coef if coef.ndim else float(coef)
        aux = self.add_variables_array((cols.shape[0],), lower_bounds=0, upper_bounds=None, costs=costs)
        cols2 = np.column_stack((cols, aux))
        if upper_bounds is not None:   # y - aux <= upper
            self.add_inequality_constraints(cols2, np.column_stack((vals, -np.ones(vals.shape[0]))), None, upper_bounds)
        if lower_bounds is not None:   # y + aux >= lower
            self.add_inequality_constraints(cols2, np.column_stack((vals, np.ones(vals.shape[0]))), lower_bounds, None)
        return aux

    def add_soft_equality_constraints(self, cols, vals, b, coef_penalization):
        """Adds ``sum_i coef[i] * |y_i - b_i|`` to the objective (reference :546-558)."""
        return self.add_soft_inequality_constraints(cols, vals, coef_penalization, lower_bounds=b, upper_bounds=b)

    # ------------------------------------------------------------------ transforms
    def convert_to_one_sided_inequality_system(self):
        """``b_lower <= A x <= b_upper`` -> ``A' x <= b_upper'`` (finite upper rows, then negated finite lower rows)."""
        if self.a_inequalities is None or self.b_lower is None:
            return
        up = np.nonzero(self.b_upper != np.inf)[0]
        lo = np.nonzero(self.b_lower != -np.inf)[0]
        a = self.a_inequalities
        if len(lo) > 0 and len(up) > 0:
            a = scipy.sparse.vstack((a[up, :], -a[lo, :])).tocsr()
        elif len(lo) > 0:
            a = -a
        a.__dict__["blocks"] = [(0, a.shape[0] - 1)]
        self.a_inequalities = a
        self.b_upper = np.hstack((self.b_upper[up], -self.b_lower[lo]))
        self.b_lower = None

    def remove_fixed_variables(self):
        """Drop the variables with ``upper == lower`` (reference :632-674).

        Returns ``(free, shift)``: the boolean mask of kept variables and the
        vector holding the fixed values; right-hand sides are moved accordingly.
        """
        free = self.upper_bounds > self.lower_bounds
        shift = np.zeros(self.nb_variables)
        shift[~free] = self.lower_bounds[~free]
        self.b_equalities = self.b_equalities - self.a_equalities * shift
        if self.b_lower is not None:
            self.b_lower = self.b_lower - self.a_inequalities * shift
        if self.b_upper is not None:
            self.b_upper = self.b_upper - self.a_inequalities * shift
        for name in ("a_inequalities", "a_equalities"):
            a = getattr(self, name)
            blocks = a.__dict__.get("blocks", [])
            a = a[:, free]
            a.__dict__["blocks"] = blocks
            setattr(self, name, a)
        self.costsvector = self.costsvector[free]
        self.lower_bounds = self.lower_bounds[free]
        self.upper_bounds = self.upper_bounds[free]
        self.nb_variables = int(np.sum(free))
        return free, shift

    # ---------------------------------------------------------------------- checks
    def max_constraint_violation(self, solution):
        """Largest violation of any bound or constraint of the (unscaled) LP (reference :186-204)."""
        worst = 0
        worst = max(worst, np.max(self.lower_bounds - solution))
        worst = max(worst, np.max(solution - self.upper_bounds))
        if self.a_equalities.shape[0] > 0:
            worst = max(worst, np.max(np.abs(self.a_equalities * solution - self.b_equalities)))
        if self.a_inequalities.shape[0] > 0:
            ax = self.a_inequalities * solution
            if self.b_upper is not None:
                worst = max(worst, np.max(ax - self.b_upper))
            if self.b_lower is not None:
                worst = max(worst, np.max(self.b_lower - ax))
        return worst

    def check_solution(self, solution, tol=1e-6):
        return bool(self.max_constraint_violation(solution) < tol)

    # ------------------------------------------------------------------------- I/O
    def save_mps(self, filename, exact=True):
        """Write the LP as an MPS file (reference SparseLP.py:280-366: objective row ``OBJ``, equality rows ``E<i>``,
        inequality rows ``I<i>``, variables ``X<j>``, right-hand-side set ``RHS0``, bound set ``bound``).

        Differences from the reference, which cannot run as written (``a_eq.ruse_preconditioning``, :310) and prints six
        decimals: numbers are written with 17 significant digits, so that reading the file back
        (``MPSparser.mps_parser``) returns the arrays bit for bit; two-sided and lower-bounded inequality rows are written
        (``G`` rows, ``RANGES`` -- the reference asserts ``b_lower is None``); bounds are written only where they differ
        from the MPS default ``[0, +inf)``.  Integer variables are refused (the reader refuses them too).  A two-sided row
        whose bounds neither ``rhs - range`` nor ``rhs + range`` reproduces in fp64 raises ``ValueError`` (``exact=False``:
        written in the L form, its lower bound then comes back within one ulp)."""
        if np.any(self.is_integer):
            raise NotImplementedError("save_mps: integer variables are not supported")
        n = self.nb_variables
        a_eq, a_in = scipy.sparse.csc_matrix(self.a_equalities), scipy.sparse.csc_matrix(self.a_inequalities)
        m_in = a_in.shape[0]
        up = np.full(m_in, np.inf) if self.b_upper is None else np.asarray(self.b_upper, dtype=np.float64)
        lo = np.full(m_in, -np.inf) if self.b_lower is None else np.asarray(self.b_lower, dtype=np.float64)
        if np.any(np.isneginf(lo) & np.isposinf(up)):
            raise ValueError("save_mps: an inequality row without any finite bound cannot be written")

        def num(v):
            return "%.17g" % v

        with open(filename, "w") as f:
            f.write("NAME          exportedFromPython\nROWS\n N  OBJ\n")
            for i in range(a_eq.shape[0]):
                f.write(" E  E%d\n" % i)
            # a two-sided row is an L row with a range (b_lower = rhs - range) unless only the G form (b_upper = rhs + range)
            # reproduces both bounds exactly
            g_row = ~np.isfinite(up)
            both = np.isfinite(up) & np.isfinite(lo)
            with np.errstate(invalid="ignore"):
                l_exact, g_exact = up - (up - lo) == lo, lo + (up - lo) == up
            g_row |= both & ~l_exact & g_exact
            if exact and np.any(both & ~l_exact & ~g_exact):
                # neither rhs - range nor rhs + range gives the other bound back in fp64: the file could not return the arrays bit
                # for bit (the promise above), so refuse instead of writing a bound that is off by an ulp
                bad = int(np.nonzero(both & ~l_exact & ~g_exact)[0][0])
                raise ValueError("save_mps: the two bounds of inequality row %d cannot be written exactly as a right-hand side and a "
                                 "range; split it into an upper-bounded and a lower-bounded row, or pass exact=False" % bad)
            for i in range(m_in):
                f.write(" %s  I%d\n" % ("G" if g_row[i] else "L", i))
            f.write("COLUMNS\n")
            for j in range(n):
                f.write("    X%-9d OBJ        %s\n" % (j, num(self.costsvector[j])))
                for k in range(a_eq.indptr[j], a_eq.indptr[j + 1]):
                    f.write("    X%-9d E%-9d %s\n" % (j, a_eq.indices[k], num(a_eq.data[k])))
                for k in range(a_in.indptr[j], a_in.indptr[j + 1]):
                    f.write("    X%-9d I%-9d %s\n" % (j, a_in.indices[k], num(a_in.data[k])))
            f.write("RHS\n")
            for i in range(a_eq.shape[0]):
                f.write("    RHS0       E%-9d %s\n" % (i, num(self.b_equalities[i])))
            for i in range(m_in):
                f.write("    RHS0       I%-9d %s\n" % (i, num(lo[i] if g_row[i] else up[i])))
            f.write("RANGES\n")
            for i in range(m_in):
                if np.isfinite(up[i]) and np.isfinite(lo[i]):
                    f.write("    RNG0       I%-9d %s\n" % (i, num(up[i] - lo[i])))
            f.write("BOUNDS\n")
            for j in range(n):
                l, u = self.lower_bounds[j], self.upper_bounds[j]
                if np.isneginf(l) and np.isposinf(u):
                    f.write(" FR bound      X%d\n" % j)
                    continue
                if l == u:
                    f.write(" FX bound      X%-9d %s\n" % (j, num(l)))
                    continue
                if np.isneginf(l):
                    f.write(" MI bound      X%d\n" % j)
                elif l != 0:
                    f.write(" LO bound      X%-9d %s\n" % (j, num(l)))
                if not np.isposinf(u):
                    f.write(" UP bound      X%-9d %s\n" % (j, num(u)))
            f.write("ENDATA\n")

    # ----------------------------------------------------------------------- solve
    def solve(
        self,
        method="admm",
        get_timing=True,
        x0=None,
        nb_iter=10000,
        max_time=None,
        callback_func=None,
        nb_iter_plot=10,
        plot_solution=None,
        ground_truth=None,
        ground_truth_indices=None,
        order=ORDER_AUTO,
        xstep="gauss_seidel",
    ):
        """Run a first-order solver on the GPU; returns ``(x, elapsed)`` or ``x``.

        Extensions (keyword-only in spirit; the reference's positional arguments are unchanged): ``order`` -- summation
        order of the dot products (include/slp_hip.h); ``xstep`` for ``method="admm"`` -- ``"gauss_seidel"`` is the
        reference as shipped (one projected Gauss-Seidel sweep on the explicit ``M = 2 A^T A + 3 I``, ADMM.py:162),
        ``"cg"`` its conjugate-gradient branch (ADMM.py:182-201) run matrix-free -- the ADMM that exists where ``M`` cannot
        (1000 entries per row: ``M`` is dense) and the one that shards over several GPUs -- and ``"auto"`` picks ``"cg"``
        when the standard-form rows promise more than ``ADMM_AUTO_M_ENTRIES`` (2e8) entries in ``M`` (sum of squared row
        lengths) or a communicator is active (the sequential sweep does not partition), else ``"gauss_seidel"``.

        Under a communicator (``parallel.init_comm_from_env`` in every rank of a ``torch.distributed.run`` / ``mpirun``
        launch) every rank calls ``solve`` on the same LP: ``"chambolle_pock_ppd"`` and ``"admm"`` with ``xstep="cg"`` hand
        over only this rank's block of constraint rows (equal stored entries per rank), exchange one / two all-reduces of
        the variable vector per iteration and return the same ``x`` on every rank; the exact Gauss-Seidel ADMM and
        ``admm_blocks`` run as replicas.

        Fills, at every report (every ``nb_iter_plot`` iterations): ``itrn_curve,
        opttime_curve, dopttime_curve, pobj_curve, dobj_curve,
        max_violated_constraint, max_violated_equality, max_violated_inequality,
        distance_to_ground_truth, distanceToGroundTruthAfterRounding``.
        As in the reference, ``callback_func`` is accepted and unused (it is
        shadowed by the internal bookkeeping callback, reference :997,:1064).
        """
        if method not in solving_methods:
            raise ValueError(f"method {method!r} not valid; available methods: {solving_methods}")
        a_ineq = self.a_inequalities if (self.a_inequalities is not None and self.a_inequalities.shape[0] > 0) else None
        a_eq, b_eq = (self.a_equalities, self.b_equalities) if self.a_equalities.shape[0] > 0 else (None, None)
        if a_ineq is not None:
            assert a_ineq.indices.size == 0 or a_ineq.indices.max() < a_ineq.shape[1]
        start = time.perf_counter()
        for name in ("distance_to_ground_truth", "distanceToGroundTruthAfterRounding", "opttime_curve", "dopttime_curve",
                     "pobj_curve", "dobj_curve", "pobjbound", "max_violated_inequality", "max_violated_equality",
                     "max_violated_constraint", "itrn_curve"):
            setattr(self, name, [])

        def record(niter, solution, energy1, energy2, duration, max_violated_equality, max_violated_inequality):
            if ground_truth is not None:
                picked = solution[ground_truth_indices]
                self.distance_to_ground_truth.append(np.mean(np.abs(ground_truth - picked)))
                self.distanceToGroundTruthAfterRounding.append(np.mean(np.abs(ground_truth - np.round(picked))))
            self.itrn_curve.append(niter)
            self.opttime_curve.append(duration)
            self.dopttime_curve.append(duration)
            self.dobj_curve.append(energy2)
            self.pobj_curve.append(energy1)
            self.max_violated_constraint.append(self.max_constraint_violation(solution))
            self.max_violated_equality.append(max_violated_equality)
            self.max_violated_inequality.append(max_violated_inequality)
            if plot_solution is not None:
                plot_solution(niter, solution, is_active_variable=None)

        if method == "admm":
            if xstep == "auto":
                from .parallel import comm_world

                rows_sq = 0.0
                for blk, slack in ((a_eq, 0), (a_ineq, 1)):
                    if blk is not None:
                        rows_sq += float(np.sum((np.diff(blk.indptr).astype(np.float64) + slack) ** 2))
                xstep = "cg" if (rows_sq > ADMM_AUTO_M_ENTRIES or comm_world()[0] > 1) else "gauss_seidel"
            x = lp_admm(self.costsvector, a_eq, b_eq, a_ineq, self.b_lower, self.b_upper, self.lower_bounds,
                        self.upper_bounds, nb_iter=nb_iter, x0=x0, callback_func=record, max_time=max_time,
                        nb_iter_plot=nb_iter_plot, order=order, xstep=xstep)
        elif method == "admm_blocks":  # reference :1210-1225
            from .ADMMBlocks import lp_admm_block_decomposition

            x = lp_admm_block_decomposition(self.costsvector, a_eq, b_eq, a_ineq, self.b_lower, self.b_upper, self.lower_bounds,
                                            self.upper_bounds, nb_iter=nb_iter, nb_iter_plot=nb_iter_plot, x0=x0,
                                            callback_func=record, max_time=max_time)
        else:  # chambolle_pock_ppd: fixed variables are eliminated first (reference :1244-1248)
            reduced = copy.deepcopy(self)
            free, shift = reduced.remove_fixed_variables()
            free_ids = np.nonzero(free)[0]

            def expand(sol):
                # reference :1259,:1288: x = m_change * sol - shift (note the sign it applies to the fixed values)
                full = np.zeros(free.size)
                full[free_ids] = sol
                return full - shift

            def record_reduced(niter, solution, *rest):
                record(niter, expand(solution), *rest)

            x, _ = chambolle_pock_ppd(reduced.costsvector, reduced.a_equalities, reduced.b_equalities,
                                      reduced.a_inequalities, reduced.b_lower, reduced.b_upper, reduced.lower_bounds,
                                      reduced.upper_bounds, x0=None, alpha=1, theta=1, nb_max_iter=nb_iter,
                                      callback_func=record_reduced, max_time=max_time, nb_iter_plot=nb_iter_plot,
                                      order=order)
            x = expand(x)
        elapsed = time.perf_counter() - start
        return (x, elapsed) if get_timing else x
