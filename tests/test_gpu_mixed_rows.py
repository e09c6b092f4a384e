"""Equality rows on the at-scale paths (the 10 %-equality variant of the synthetic LP, randomLP.py:62-68):
device-side standard form [A_eq 0; A_ineq -I] for the matrix-free ADMM, and Chambolle-Pock with both kinds of
rows through the strip kernels.  Against the oracle on the downloaded matrix.  -m gpu."""
import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("min_nnz", ["1", "100000000000"])  # strip kernels / CSR kernels
def test_mixed_rows_match_oracle(monkeypatch, min_nnz):
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", min_nnz)
    n, m, p, m_eq = 30000, 40000, 0.001, 4000
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=9)
    s = a.download()
    b = b.copy()
    b[:m_eq] = a.matvec(xf)[:m_eq]
    ae, ai = s[:m_eq], s[m_eq:]
    # Chambolle-Pock
    cp = DeviceCP(a, b, c, lb, ub, m_eq=m_eq)
    cp.iterate(40)
    x = cp.x()
    cp.close()
    xo, _ = oracle.chambolle_pock_ppd(c, ae, b[:m_eq], ai, None, b[m_eq:], lb, ub, nb_max_iter=40, nb_iter_plot=10 ** 9)
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-10
    # matrix-free ADMM (scales the matrix in place: run it last)
    for level in (2, 0):
        a2, _, _, _, _, _ = random_lp_on_device(n, m, p, seed=9)
        admm = DeviceADMM(a2, b, c, lb, ub, m_eq=m_eq, reuse=level)
        admm.iterate(25)
        x = admm.x(n)
        rep = admm.report()
        admm.close()
        a2.close()
        xo = oracle.lp_admm_cg(c, ae, b[:m_eq], ai, None, b[m_eq:], lb, ub, nb_iter=24, nb_iter_plot=10 ** 9)
        assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-9
        assert abs(c.dot(x) - c.dot(xo)) <= 1e-6 * abs(c.dot(xo))
        assert np.all(np.isfinite(rep[:3]))
    a.close()


@pytest.mark.parametrize("min_nnz", ["1", "100000000000"])  # (value-dictionary) strip kernels with the row-scale vector / CSR kernels, scaled in place
def test_two_sided_rows_admm_match_oracle(monkeypatch, min_nnz):
    """b_lower <= A x <= b_upper on the at-scale ADMM: the lower bounds are scaled with their rows and bound the slack variables
    from below (tools.py:117-121,286-288); a third of the rows keep b_lower = -inf, some are equalities."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", min_nnz)
    n, m, p, m_eq = 30000, 40000, 0.001, 3000
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=11)
    s = a.download()
    ax = a.matvec(xf)
    rng = np.random.RandomState(4)
    b = b.copy()
    b[:m_eq] = ax[:m_eq]
    bl = np.where(rng.rand(m) < 0.33, -np.inf, ax - rng.rand(m))  # feasible: b_lower <= A xf <= b_upper
    ae, ai = s[:m_eq], s[m_eq:]
    admm = DeviceADMM(a, b, c, lb, ub, m_eq=m_eq, b_lower=bl)
    admm.iterate(25)
    x = admm.x(n)
    slack = admm.x(n + m)[n + m_eq:]
    admm.close()
    a.close()
    xo = oracle.lp_admm_cg(c, ae, b[:m_eq], ai, bl[m_eq:], b[m_eq:], lb, ub, nb_iter=24, nb_iter_plot=10 ** 9)
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-9
    assert abs(c.dot(x) - c.dot(xo)) <= 1e-6 * abs(c.dot(xo))
    assert np.sum(np.isfinite(bl[m_eq:])) > 20000 and np.all(np.isfinite(slack))


@pytest.mark.parametrize("min_nnz", ["1", "100000000000"])
def test_two_sided_rows_cp_match_oracle(monkeypatch, min_nnz):
    """b_lower <= A x <= b_upper on the at-scale Chambolle-Pock: the one-sided stacking [A[up]; -A[lo]] of
    ChambollePockPPD.py:74-88 is built on the device (slp_matrix_gather_rows); equality rows stay in front."""
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", min_nnz)
    n, m, p, m_eq = 30000, 40000, 0.001, 3000
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=12)
    s = a.download()
    ax = a.matvec(xf)
    rng = np.random.RandomState(6)
    b = b.copy()
    b[:m_eq] = ax[:m_eq]
    bl = np.where(rng.rand(m) < 0.4, -np.inf, ax - rng.rand(m))
    bu = np.where(rng.rand(m) < 0.2, np.inf, b)   # some rows only bounded from below
    bu[:m_eq] = b[:m_eq]
    bl[(bl == -np.inf) & (bu == np.inf)] = -5.0   # every inequality row keeps at least one finite side
    ae, ai = s[:m_eq], s[m_eq:]
    cp = DeviceCP(a, bu, c, lb, ub, m_eq=m_eq, b_lower=bl, order=1)
    assert cp.a.shape[0] > m  # stacked
    cp.iterate(40)
    x = cp.x()
    cp.close()
    a.close()
    xo, _ = oracle.chambolle_pock_ppd(c, ae, b[:m_eq], ai, bl[m_eq:], bu[m_eq:], lb, ub, nb_max_iter=40, nb_iter_plot=10 ** 9)
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-10


def test_gather_rows_on_device():
    from pysparselp_amd.problems import random_lp_on_device

    a = random_lp_on_device(500, 700, 0.02, seed=1)[0]
    s = a.download()
    rows = np.array([5, 5, 699, 0, 12], dtype=np.int64)
    scale = np.array([1.0, -1.0, 2.5, 1.0, -1.0])
    g = a.gather_rows(rows, scale)
    got = g.download()
    ref = (scipy.sparse.diags(scale) @ s[rows]).tocsr()
    ref.sort_indices()  # scipy's product leaves the rows reversed
    assert got.shape == ref.shape and np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
    assert np.array_equal(got.data, ref.data)
    assert a.gather_rows(np.zeros(0, dtype=np.int64)).shape == (0, 500)
    g.close()
    a.close()


def test_cp_on_generated_matrix_with_short_rows_regression():
    """Regression: the generator left max_row_len unset, so the ELL copy of Chambolle-Pock's short-row path was built 4 wide
    and truncated rows of ~7-20 entries.  Device-generated 25000 x 9000 LP, CSR kernels, one lane per row: bit-exact."""
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    a, xf, c, lb, ub, b = random_lp_on_device(9000, 25000, 0.0008, seed=684)
    s = a.download()
    assert np.diff(s.indptr).max() > 16  # longer than the widest ELL copy: the CSR walk must be used
    for order in (0, 1):
        cp = DeviceCP(a, b, c, lb, ub, order=order)
        cp.iterate(10)
        x = cp.x()
        cp.close()
        xo, _ = oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=10, nb_iter_plot=10 ** 9)
        assert np.max(np.abs(x - xo)) < 1e-14
    a.close()


def test_randomised_at_scale_solvers():
    """tools/fuzz_scale.py: DeviceCP and DeviceADMM (reuse levels 0 ... 4) on device-generated LPs of mixed shapes, every
    strip variant and the CSR kernels, equality and two-sided rows -- against the oracle."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_scale

    assert fuzz_scale.run(12, seed=2) == 12
