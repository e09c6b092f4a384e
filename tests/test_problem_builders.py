"""The modelling layer rebuilds the reference's example LPs array for array (CPU), and -- on the GPU --
the reference's own example-level tests run through it: tests/test_kmedians.py:11-14 and
tests/test_l1_svm.py:12-26."""
import numpy as np
import pytest

from conftest import csr_of, load_golden
from pysparselp_amd.problems import kmedians_lp, l1svm_lp


def _same_lp(lp, d):
    assert np.array_equal(lp.costsvector, d["c"])
    assert np.array_equal(lp.lower_bounds, d["lb"]) and np.array_equal(lp.upper_bounds, d["ub"])
    assert np.array_equal(lp.b_equalities, d["be"])
    assert np.array_equal(lp.b_upper, d["bu"]) and np.array_equal(lp.b_lower, d["bl"])
    for tag, m in (("Ae", lp.a_equalities), ("Ai", lp.a_inequalities)):
        r = csr_of(d, tag)
        assert m.shape == r.shape
        assert np.array_equal(m.indptr, r.indptr) and np.array_equal(m.indices, r.indices) and np.array_equal(m.data, r.data)


def _l1svm_data():
    np.random.seed(1)
    x = np.random.rand(1000, 2)
    xh = np.hstack((x, np.ones((1000, 1))))
    w = np.random.randn(3, 2)
    w = w / np.sum(w ** 2, axis=1)[:, None]
    w = np.hstack((w, -0.5 * np.sum(w, axis=1)[:, None]))
    return x, xh, np.argmax((w.dot(xh.T)).T, axis=1)


def test_kmedians_lp_matches_reference_arrays():
    d = load_golden("ka_kmedians")
    lp, labeling, _ = kmedians_lp(d["points"], 5, 50)
    _same_lp(lp, d)
    assert labeling.shape == (500, 50) and lp.nb_equality_constraints() == 500


def test_l1svm_lp_matches_reference_arrays():
    d = load_golden("ka_l1svm")
    x, _, classes = _l1svm_data()
    lp, w, eps = l1svm_lp(x, classes)
    _same_lp(lp, d)
    assert np.isneginf(lp.lower_bounds[w]).all() and np.isposinf(lp.upper_bounds[eps]).all()


def test_soft_constraints_add_one_auxiliary_per_row():
    from pysparselp_amd.SparseLP import SparseLP

    lp = SparseLP()
    v = lp.add_variables_array(3, 0, 1, costs=0)
    cols = np.array([[0, 1], [1, 2]])
    aux = lp.add_soft_equality_constraints(cols, np.array([[1.0, -1.0]]), b=np.array([0.3, -0.2]), coef_penalization=2.5)
    assert lp.nb_variables == 5 and np.array_equal(aux, [3, 4]) and np.array_equal(lp.costsvector[aux], [2.5, 2.5])
    a = lp.a_inequalities.toarray()
    assert a.shape == (4, 5)
    assert np.array_equal(a[0], [1, -1, 0, -1, 0]) and np.array_equal(a[2], [1, -1, 0, 1, 0])  # y - aux <= b ; y + aux >= b
    assert np.array_equal(lp.b_upper[:2], [0.3, -0.2]) and np.isneginf(lp.b_lower[:2]).all()
    assert np.array_equal(lp.b_lower[2:], [0.3, -0.2]) and np.isposinf(lp.b_upper[2:]).all()
    assert lp.add_soft_inequality_constraints(cols, np.array([[1.0, 1.0]]), np.inf, upper_bounds=1.5) is None
    assert lp.nb_inequality_constraints() == 6 and lp.nb_variables == 5
    assert v.size == 3


@pytest.mark.gpu
def test_kmedians_like_the_reference_test():
    """cost == 238.9849948936172 (tests/test_kmedians.py:14), LP built by this package's modelling layer."""
    d = load_golden("ka_kmedians")
    lp, labeling, pairdistances = kmedians_lp(d["points"], 5, 50)
    s = lp.solve(method="admm", nb_iter=1000, max_time=np.inf, nb_iter_plot=500)[0]
    label = np.argmax(s[labeling], axis=1)
    cost = 0
    for l in range(50):
        group = np.nonzero(label == l)
        center_id = np.argmin(np.sum(pairdistances[group, :], axis=1))
        cost += np.sum(pairdistances[group, center_id])
    assert cost == 238.9849948936172
    assert lp.max_constraint_violation(s) < 0.2 and len(lp.itrn_curve) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("method,expected", [("admm", 99.5), ("chambolle_pock_ppd", 99.4)])
def test_l1svm_like_the_reference_test(method, expected):
    """tests/test_l1_svm_results.json through SparseLP.solve(nb_iter=2000)."""
    x, xh, classes = _l1svm_data()
    lp, w, eps = l1svm_lp(x, classes)
    sol, elapsed = lp.solve(method=method, get_timing=True, nb_iter=2000, max_time=np.inf, plot_solution=None)
    weights = sol[w]
    assert 100 * np.mean(classes == np.argmax(xh.dot(weights.T), axis=1)) == expected
