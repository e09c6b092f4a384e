"""The at-scale ADMM (matrix-free CG x-step, csrc/slp_admm_cg.hip) against the ORACLE across the refresh of its
recurrences.  Reuse levels 3 / 4 carry ``A dir`` and ``M dir`` by recurrences and take them as products again every 64
iterations (slp_admm_cg.hip: cg_multipliers); the bench default is level 4.  Here levels 0, 2 and 4 run 200 iterations on a
20 000 x 40 000 random LP of the benchmark distribution (density 5e-3) and are compared with ``oracle.lp_admm_cg``
(ADMM.py:182-201 + conjugateGradientLinearSolver.py:30-52 restated, pinned by the flag-flipped reference's iterates in
tests/golden) after 50, 130 (two refreshes passed) and 200 iterations -- on the value-dictionary strips with the deferred
row scaling (``rs``), on the fp64 strips with the in-place scaling, on the CSR kernels, and with equality rows (m_eq > 0).

Tolerances (fp64, summation-order differences only; the iteration is not contractive):
|x_gpu - x_oracle| <= 1e-9 (1 + |x|) per entry and |c.x_gpu - c.x_oracle| <= 1e-9 |c.x|.  -m gpu."""
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

N, M, P = 20_000, 40_000, 5e-3
CHECK = (50, 130, 200)


def _oracle_iterates(c, s_eq, beq, s_ineq, b, lb, ub):
    snaps = {}

    def hook(i, x, *_):
        if i + 1 in CHECK:  # hook i sees the iterate after i + 1 x-steps = what `iterate(i + 1)` leaves on the device
            snaps[i + 1] = np.array(x, copy=True)

    oracle.lp_admm_cg(c, s_eq, beq, s_ineq, None, b, lb, ub, nb_iter=max(CHECK) - 1, nb_iter_plot=10 ** 9, iterate_hook=hook)
    return snaps


def _compare(solver, c, ref, n):
    done = 0
    for k in CHECK:
        solver.iterate(k - done)
        done = k
        x = solver.x(n)
        err = float(np.max(np.abs(x - ref[k]) / (1 + np.abs(ref[k]))))
        assert err < 1e-9, (k, err)
        assert abs(float(c.dot(x)) - float(c.dot(ref[k]))) <= 1e-9 * abs(float(c.dot(ref[k]))), k


@pytest.fixture(scope="module")
def lp():
    from pysparselp_amd.problems import random_lp_on_device

    a, xf, c, lb, ub, b = random_lp_on_device(N, M, P, seed=1)
    s = a.download()
    a.close()
    return s, xf, c, lb, ub, b, _oracle_iterates(c, None, None, s, b, lb, ub)


@pytest.mark.parametrize("policy,level", [(0, 0), (0, 2), (0, 3), (0, 4), (1, 4), (2, 4)])
def test_reuse_levels_follow_the_oracle_across_refreshes(lp, policy, level):
    """policy 0: value-dictionary strips + deferred row scaling; 1: fp64 strips, rows scaled in place; 2: CSR kernels."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device

    s, xf, c, lb, ub, b, ref = lp
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        a = random_lp_on_device(N, M, P, seed=1)[0]
        a.set_format(policy)
        solver = DeviceADMM(a, b, c, lb, ub, reuse=level)
        kernels = {0: (2, 3), 1: (1,), 2: (0,)}[policy]
        assert a.spmv_kernel(False) in kernels and a.spmv_kernel(True) in kernels
        _compare(solver, c, ref, N)
        solver.close()
        a.close()
    finally:
        del os.environ["SLP_STRIP_MIN_NNZ"]


@pytest.mark.parametrize("policy", [0, 1])
def test_level4_with_equality_rows_follows_the_oracle(lp, policy):
    """The first 10 % of the rows as equalities a_i x = a_i x_feasible (randomLP.py:62-68): no slack entry on those rows,
    right-hand side scaled in both passes (tools.py:96-107)."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device

    s, xf, c, lb, ub, b, _ = lp
    m_eq = M // 10
    beq = oracle.matvec(oracle.as_csr(s[:m_eq]), xf)
    b2 = b.copy()
    b2[:m_eq] = beq
    ref = _oracle_iterates(c, s[:m_eq], beq, s[m_eq:], b[m_eq:], lb, ub)
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        a = random_lp_on_device(N, M, P, seed=1)[0]
        a.set_format(policy)
        solver = DeviceADMM(a, b2, c, lb, ub, reuse=4, m_eq=m_eq)
        _compare(solver, c, ref, N)
        solver.close()
        a.close()
    finally:
        del os.environ["SLP_STRIP_MIN_NNZ"]
