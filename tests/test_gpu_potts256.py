"""BASELINE config 2: the Potts image-segmentation LP on a 256 x 256 grid (196 096 variables,
261 120 inequality rows, 3 entries per row), Chambolle-Pock on one MI355X -- iterates against the
oracle bit for bit, and the ADMM (Gauss-Seidel) path on the same LP.  -m gpu."""
import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def potts256():
    from pysparselp_amd.problems import potts_lp

    lp, gt, gt_idx, _ = potts_lp(256)
    assert lp.nb_variables == 196096 and lp.a_inequalities.shape == (261120, 196096) and lp.a_inequalities.nnz == 783360
    return lp, gt, gt_idx


def test_cp_potts256_bit_exact(potts256):
    from pysparselp_amd.ChambollePockPPD import chambolle_pock_ppd

    lp, gt, gt_idx = potts256
    args = (lp.costsvector, scipy.sparse.csr_matrix((0, lp.nb_variables)), np.zeros(0), lp.a_inequalities, lp.b_lower,
            lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    seen = {}
    x, _ = chambolle_pock_ppd(*args, nb_max_iter=150, nb_iter_plot=50,
                              callback_func=lambda it, sol, *r: seen.__setitem__(it, (sol.copy(),) + r))
    ref = {}
    xo, _ = oracle.chambolle_pock_ppd(*args, nb_max_iter=150, nb_iter_plot=50,
                                      callback_func=lambda it, sol, *r: ref.__setitem__(it, (sol.copy(),) + r))
    assert np.array_equal(x, xo)
    assert sorted(seen) == sorted(ref) == [0, 50, 100]
    for it in ref:
        assert np.array_equal(seen[it][0], ref[it][0])
        np.testing.assert_allclose(seen[it][1:3], ref[it][1:3], rtol=1e-9)  # energies (BLAS vs tree sums)
        assert seen[it][5] == ref[it][5]                                     # max violated inequality


def test_solve_potts256_converges_towards_graph_cut(potts256):
    """SparseLP.solve end to end: the distance to the exact min-cut labelling decreases."""
    lp, gt, gt_idx = potts256
    lp.solve(method="chambolle_pock_ppd", nb_iter=3000, nb_iter_plot=500, ground_truth=gt, ground_truth_indices=gt_idx)
    d = lp.distance_to_ground_truth
    assert len(d) == 6 and d[-1] < 0.1 * d[0]
    assert lp.max_violated_constraint[-1] < 1e-2


def test_admm_potts256_bit_exact(potts256):
    from pysparselp_amd.ADMM import lp_admm

    lp, gt, gt_idx = potts256
    args = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    x = lp_admm(*args, nb_iter=12, nb_iter_plot=5)
    xo = oracle.lp_admm(*args, nb_iter=12, nb_iter_plot=5)
    assert np.array_equal(x, xo)
