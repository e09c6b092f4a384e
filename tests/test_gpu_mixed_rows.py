"""Equality rows on the at-scale paths (the 10 %-equality variant of the synthetic LP, randomLP.py:62-68):
device-side standard form [A_eq 0; A_ineq -I] for the matrix-free ADMM, and Chambolle-Pock with both kinds of
rows through the strip kernels.  Against the oracle on the downloaded matrix.  -m gpu."""
import numpy as np
import pytest

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
