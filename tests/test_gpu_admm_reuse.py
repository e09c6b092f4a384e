"""8-pass form of the matrix-free ADMM (the CG residual reuses the line search's
products M x and M dir): same reference iterates, same tolerance.  -m gpu."""
import numpy as np
import pytest

from conftest import Recorder, load_golden, solver_args

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["sc50a", "sc105", "potts8", "random1"])
def test_admm_cg_reuse_matches_reference_iterates(case):
    from pysparselp_amd.admm_cg import lp_admm_cg

    d = load_golden("lp_" + case)
    keep = [it for it in d["admmcg_it"] if it <= 200]
    rec = Recorder(keep)
    lp_admm_cg(*solver_args(d), nb_iter=200, callback_func=rec, nb_iter_plot=1, reuse=True)
    assert rec.it == keep
    for got, ref in zip(rec.x, d["admmcg_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < 1e-9


@pytest.mark.parametrize("case", ["sc50a", "sc105", "potts8", "random1"])
def test_admm_cg_fused_matches_reference_iterates(case):
    """reuse level 2: A^T (g_eq A x + lambda_eq) as one product (six products per iteration)."""
    from pysparselp_amd.admm_cg import lp_admm_cg

    d = load_golden("lp_" + case)
    keep = [it for it in d["admmcg_it"] if it <= 200]
    rec = Recorder(keep)
    lp_admm_cg(*solver_args(d), nb_iter=200, callback_func=rec, nb_iter_plot=1, reuse=2)
    assert rec.it == keep
    for got, ref in zip(rec.x, d["admmcg_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < 1e-9
    for got, ref in zip(rec.e1, d["admmcg_e1"]):
        assert abs(got - ref) <= 1e-9 * (1 + abs(ref))


def test_reuse_levels_agree_on_device_generated_lp():
    """Strip kernels + two-vector passes: levels 0 ... 4 give the same iterates to rounding."""
    import os
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device

    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        xs = []
        for level in (0, 1, 2, 3, 4):
            a, xf, c, lb, ub, b = random_lp_on_device(30000, 40000, 0.001, seed=3)
            s = DeviceADMM(a, b, c, lb, ub, reuse=level)
            assert s.matrix_passes_per_iteration() == {0: 10, 1: 5, 2: 4, 3: 4, 4: 4}[level]
            assert s.matrix_products_per_iteration() == {0: 10, 1: 8, 2: 6, 3: 5, 4: 4}[level]
            s.iterate(25)
            xs.append(s.x(30000))
            s.close()
            a.close()
        for x in xs[1:]:
            assert np.max(np.abs(x - xs[0]) / (1 + np.abs(xs[0]))) < 1e-10
    finally:
        del os.environ["SLP_STRIP_MIN_NNZ"]


def test_level3_recurrence_does_not_drift_over_a_refresh_period():
    """Levels 3 / 4 carry A dir (and M dir) by recurrences and refresh them every 64 iterations: 150 iterations stay with level 2."""
    import os
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device

    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        xs = []
        for level in (2, 3, 4):
            a, xf, c, lb, ub, b = random_lp_on_device(30000, 40000, 0.001, seed=4)
            s = DeviceADMM(a, b, c, lb, ub, reuse=level)
            s.iterate(150)
            xs.append(s.x(30000))
            s.close()
            a.close()
        for x in xs[1:]:
            assert np.max(np.abs(x - xs[0]) / (1 + np.abs(xs[0]))) < 1e-9
            assert abs(c.dot(x) - c.dot(xs[0])) <= 1e-9 * abs(c.dot(xs[0]))
    finally:
        del os.environ["SLP_STRIP_MIN_NNZ"]
