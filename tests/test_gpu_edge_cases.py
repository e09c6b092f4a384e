"""Edge cases of the solver entry points, each against the oracle: empty / ragged inputs, equality-only
and two-sided systems, warm start, non-default alpha / theta, degenerate calls and error behaviour.  -m gpu."""
import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu
BIG = 10 ** 9


def _lp(seed, n=35, me=6, mi=50, density=0.25, empty_row=None, empty_col=None):
    rng = np.random.RandomState(seed)
    ae = scipy.sparse.random(me, n, density=density, random_state=rng, format="lil")
    ai = scipy.sparse.random(mi, n, density=density, random_state=rng, format="lil")
    if empty_row is not None:
        ai[empty_row, :] = 0
    if empty_col is not None:
        ai[:, empty_col] = 0
        ae[:, empty_col] = 0
    ae, ai = ae.tocsr(), ai.tocsr()
    ae.eliminate_zeros()
    ai.eliminate_zeros()
    ae.data = np.round(rng.randn(ae.nnz) * 100) / 100 + 0.005
    ai.data = np.round(rng.randn(ai.nnz) * 100) / 100 + 0.005
    xf = np.round(rng.randn(n) * 100) / 100
    be = ae @ xf
    bu = ai @ xf + rng.rand(mi)
    bl = ai @ xf - rng.rand(mi)
    c = np.round(rng.randn(n) * 100) / 100
    t = np.abs(rng.randn(n)) + 0.1
    return c, ae, be, ai, bl, bu, xf - t, xf + t


def _cp():
    from pysparselp_amd.ChambollePockPPD import chambolle_pock_ppd

    return chambolle_pock_ppd


def test_cp_two_sided_inequalities_and_equalities():
    c, ae, be, ai, bl, bu, lb, ub = _lp(0)
    bl[::3] = -np.inf  # mixed: some rows one-sided
    bu[1::4] = np.inf
    x, _ = _cp()(c, ae, be, ai, bl, bu, lb, ub, nb_max_iter=300, nb_iter_plot=BIG, order=1)
    xo, _ = oracle.chambolle_pock_ppd(c, ae, be, ai, bl, bu, lb, ub, nb_max_iter=300, nb_iter_plot=BIG)
    assert np.array_equal(x, xo)


def test_cp_only_lower_bounded_rows():
    c, ae, be, ai, bl, bu, lb, ub = _lp(1)
    bu[:] = np.inf  # every row  a x >= bl  ->  the matrix is negated (ChambollePockPPD.py:82-83)
    x, _ = _cp()(c, ae, be, ai, bl, bu, lb, ub, nb_max_iter=200, nb_iter_plot=BIG, order=1)
    xo, _ = oracle.chambolle_pock_ppd(c, ae, be, ai, bl, bu, lb, ub, nb_max_iter=200, nb_iter_plot=BIG)
    assert np.array_equal(x, xo)


def test_cp_equalities_only_runs_and_reports():
    """The reference dereferences a_ineq in its report (ChambollePockPPD.py:283) and crashes without
    inequalities; here the iteration is the same and the report gives 0 for the missing block."""
    c, ae, be, ai, bl, bu, lb, ub = _lp(2)
    empty = scipy.sparse.csr_matrix((0, c.size))
    seen = []
    x, _ = _cp()(c, ae, be, empty, None, np.zeros(0), lb, ub, nb_max_iter=120, nb_iter_plot=40, order=1,
                 callback_func=lambda it, sol, e1, e2, dt, veq, vin: seen.append((it, veq, vin)))
    # oracle: an all-zero inequality row with a positive bound is arithmetically inert (its dual stays 0)
    dummy = scipy.sparse.csr_matrix((1, c.size))
    xo, _ = oracle.chambolle_pock_ppd(c, ae, be, dummy, None, np.ones(1), lb, ub, nb_max_iter=120, nb_iter_plot=BIG)
    assert np.array_equal(x, xo)
    assert [s[0] for s in seen] == [0, 40, 80] and all(s[2] == 0 for s in seen) and seen[-1][1] > 0


def test_cp_warm_start_empty_row_and_column():
    c, ae, be, ai, bl, bu, lb, ub = _lp(3, empty_row=4, empty_col=7)
    x0 = 0.5 * (lb + ub)
    x, _ = _cp()(c, ae, be, ai, None, bu, lb, ub, x0=x0, nb_max_iter=150, nb_iter_plot=BIG, order=1)
    xo, _ = oracle.chambolle_pock_ppd(c, ae, be, ai, None, bu, lb, ub, x0=x0, nb_max_iter=150, nb_iter_plot=BIG)
    assert np.array_equal(x, xo)
    assert not np.array_equal(x0, x)


def test_cp_general_alpha_theta():
    """alpha != 1 goes through pow() on both sides (libm vs device pow: last-bit differences)."""
    c, ae, be, ai, bl, bu, lb, ub = _lp(4)
    x, _ = _cp()(c, ae, be, ai, None, bu, lb, ub, alpha=1.5, theta=0.7, nb_max_iter=100, nb_iter_plot=BIG, order=1)
    xo, _ = oracle.chambolle_pock_ppd(c, ae, be, ai, None, bu, lb, ub, alpha=1.5, theta=0.7, nb_max_iter=100, nb_iter_plot=BIG)
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-11


def test_cp_without_constraints_returns_box_vertex():
    c, ae, be, ai, bl, bu, lb, ub = _lp(5)
    empty = scipy.sparse.csr_matrix((0, c.size))
    x = _cp()(c, empty, np.zeros(0), None, None, None, lb, ub)
    assert np.array_equal(x, np.where(c > 0, lb, np.where(c < 0, ub, 0.0)))


def test_cp_rejects_bad_input():
    c, ae, be, ai, bl, bu, lb, ub = _lp(6)
    bad = ai.copy()
    bad.indices = bad.indices.copy()
    bad.indices[0] = c.size + 3
    with pytest.raises(ValueError):
        _cp()(c, ae, be, bad, None, bu, lb, ub, nb_max_iter=2)
    with pytest.raises(NotImplementedError):
        _cp()(c, ae, be, ai, None, bu, lb, ub, force_integer=True)


def test_admm_requires_inequalities_like_the_reference():
    from pysparselp_amd.ADMM import lp_admm

    c, ae, be, ai, bl, bu, lb, ub = _lp(7)
    with pytest.raises(UnboundLocalError):  # tools.py:92,127 in the reference
        lp_admm(c, ae, be, None, None, None, lb, ub, nb_iter=3)
    with pytest.raises(ValueError):
        lp_admm(c, ae, be, ai, bl, bu, lb, ub, nb_iter=3, xstep="newton")


def test_admm_two_sided_rows_warm_start_and_no_equalities():
    from pysparselp_amd.ADMM import lp_admm

    c, ae, be, ai, bl, bu, lb, ub = _lp(8, empty_row=2)
    x0 = 0.25 * (lb + ub)
    for a_eq, b_eq in ((ae, be), (None, None)):
        x = lp_admm(c, a_eq, b_eq, ai, bl, bu, lb, ub, x0=x0, nb_iter=150, nb_iter_plot=BIG, order=1)
        xo = oracle.lp_admm(c, a_eq, b_eq, ai, bl, bu, lb, ub, x0=x0, nb_iter=150, nb_iter_plot=BIG)
        assert np.array_equal(x, xo)


def test_admm_gamma_and_no_preconditioning():
    from pysparselp_amd.ADMM import lp_admm

    c, ae, be, ai, bl, bu, lb, ub = _lp(9)
    x = lp_admm(c, ae, be, ai, bl, bu, lb, ub, gamma_eq=1.5, gamma_ineq=0.75, use_preconditioning=False, nb_iter=80,
                nb_iter_plot=BIG, order=1)
    xo = oracle.lp_admm(c, ae, be, ai, bl, bu, lb, ub, gamma_eq=1.5, gamma_ineq=0.75, use_preconditioning=False, nb_iter=80,
                        nb_iter_plot=BIG)
    assert np.array_equal(x, xo)


def test_gauss_seidel_degenerate_sizes():
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    m = scipy.sparse.csr_matrix(np.array([[4.0]]))
    x = np.array([0.3])
    boundedGaussSeidelClass(m).solve(np.array([2.0]), np.array([-np.inf]), np.array([0.4]), x, maxiter=1)
    assert x[0] == 0.4
    d = scipy.sparse.diags(np.arange(1.0, 301.0)).tocsr()  # diagonal: a single level, all rows independent
    bs = boundedGaussSeidelClass(d)
    assert bs.num_levels == 1
    x = np.zeros(300)
    bs.solve(np.ones(300), np.full(300, -1.0), np.full(300, 1.0), x, maxiter=1, w=1)
    assert np.array_equal(x, 1.0 * (1.0 - 0.0) * (1 / np.arange(1.0, 301.0)) + 0.0)
    with pytest.raises(TypeError):
        bs.solve(np.ones(300), -1.0, 1.0, [0.0] * 300)


def test_matrix_entry_points_on_empty_matrices():
    from pysparselp_amd.device import DeviceMatrix

    for shape in ((0, 5), (4, 0 + 1)):
        a = scipy.sparse.csr_matrix(shape)
        dm = DeviceMatrix.from_csr(a)
        assert dm.nnz == 0
        assert np.array_equal(dm.matvec(np.ones(shape[1])), np.zeros(shape[0]))
        assert np.array_equal(dm.rmatvec(np.ones(shape[0])), np.zeros(shape[1]))


def test_randomised_lps_all_host_api_solvers():
    """tools/fuzz_solvers.py: 40 random small LPs (empty rows / columns, one- and two-sided rows, infinite bounds, with and
    without equalities, warm starts, odd reporting cadences) through lp_admm (both Gauss-Seidel forms), chambolle_pock_ppd --
    bit for bit -- and the matrix-free and block-splitting ADMM within their tolerances."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_solvers

    assert fuzz_solvers.run(40, seed=11) == 40
