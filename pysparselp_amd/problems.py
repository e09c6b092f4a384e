"""Workload builders for tests and benchmarks.

* :func:`potts_lp` -- the Potts image-segmentation LP of the reference's
  ``examples/example_pott_segmentation.py:12-92`` (BASELINE config 2 at 256x256),
  with its exact min-cut ground truth.
* :func:`random_lp_on_device` -- the synthetic random LP of ``randomLP.py:29-75``
  (BASELINE configs 3/4), generated on the GPU so that it scales to 1e6 x 2e6.
"""
import numpy as np

import scipy.sparse
from scipy.sparse.csgraph import breadth_first_order, maximum_flow

from .SparseLP import SparseLP
from .device import DeviceMatrix


def _min_cut_labels(unary, pairwise):
    """Exact minimiser of sum_p unary_p l_p + pairwise * #{neighbours with different labels},
    l in {0,1}, 4-neighbourhood (integer weights): label 1 = source side of a minimum cut."""
    h, w = unary.shape
    ids = np.arange(h * w).reshape(h, w)
    s, t = h * w, h * w + 1
    src, dst, cap = [], [], []
    for a, b in ((ids[:, 1:], ids[:, :-1]), (ids[1:, :], ids[:-1, :])):
        a, b = a.ravel(), b.ravel()
        src += [a, b]
        dst += [b, a]
        cap += [np.full(a.size, pairwise), np.full(a.size, pairwise)]
    u = unary.ravel().astype(np.int64)
    wants_one = np.nonzero(u < 0)[0]
    wants_zero = np.nonzero(u > 0)[0]
    src += [np.full(wants_one.size, s), wants_zero]
    dst += [wants_one, np.full(wants_zero.size, t)]
    cap += [-u[wants_one], u[wants_zero]]
    g = scipy.sparse.csr_matrix((np.concatenate(cap).astype(np.int32), (np.concatenate(src), np.concatenate(dst))),
                                shape=(h * w + 2, h * w + 2))
    flow = maximum_flow(g, s, t).flow
    residual = (g - flow).tocsr()
    residual.data = (residual.data > 0).astype(np.int32)
    residual.eliminate_zeros()
    reach = breadth_first_order(residual, s, directed=True, return_predecessors=False)
    label = np.zeros(h * w + 2, dtype=np.int64)
    label[reach] = 1
    return label[: h * w].reshape(h, w)


def potts_lp(image_size, coef_potts=0.5, coef_mul=500, seed=1):
    """LP relaxation of a binary Potts model on an ``image_size`` x ``image_size`` grid.

    Variables: one label in [0, 1] per pixel (cost = unary / coef_mul), one
    auxiliary |difference| variable per neighbouring pair (cost = Potts weight).
    Constraints (all inequalities, 3 non-zeros per row, values +-1):
    ``x_p - x_q - a_pq <= 0`` and ``-x_p + x_q - a_pq <= 0``.
    Returns ``(lp, ground_truth, ground_truth_indices, unary_terms)``.
    """
    rng = np.random.RandomState(seed)
    shape = (image_size, image_size, 1)
    unary = np.round(coef_mul * (rng.rand(*shape) * 2 - 1))
    potts_w = round(coef_potts * coef_mul)
    ground_truth = _min_cut_labels(unary[:, :, 0], potts_w)[:, :, None]

    lp = SparseLP()
    pix = lp.add_variables_array(shape, lower_bounds=0, upper_bounds=1, costs=unary / coef_mul)

    def penalise_differences(p, q, weight):
        span = np.maximum(lp.upper_bounds[p] - lp.lower_bounds[q], lp.upper_bounds[q] - lp.lower_bounds[p])
        aux = lp.add_variables_array(p.shape, lower_bounds=0, upper_bounds=span, costs=weight)
        cols = np.column_stack((p.ravel(), q.ravel(), aux.ravel()))
        lp.add_inequality_constraints(cols, np.tile([1.0, -1.0, -1.0], (p.size, 1)), lower_bounds=None, upper_bounds=0)
        lp.add_inequality_constraints(cols, np.tile([-1.0, 1.0, -1.0], (p.size, 1)), lower_bounds=None, upper_bounds=0)

    weight = potts_w / coef_mul
    penalise_differences(pix[:, 1:], pix[:, :-1], weight)   # horizontal neighbours
    penalise_differences(pix[1:, :], pix[:-1, :], weight)   # vertical neighbours
    return lp, ground_truth, pix, unary


def random_lp_on_device(n, m, density, seed=0, row_offset=0, rows=None):
    """Synthetic random LP ``min c.x  s.t.  A x <= b_upper, lb <= x <= ub`` (all inequalities).

    Generates rows ``row_offset .. row_offset + rows`` (default: all ``m``) of the
    m x n matrix directly in HBM.  Returns ``(DeviceMatrix, feasible_x, c, lb, ub, b_upper)``;
    ``feasible_x`` satisfies every constraint by construction (randomLP.py:33,43-46,53-55).
    """
    rows = m if rows is None else rows
    assert 0 <= row_offset and row_offset + rows <= m
    a = DeviceMatrix.random(rows, n, density, seed, row_offset)
    xf, c, lb, ub, b = a.random_lp_vectors(density, seed, row_offset)
    return a, xf, c, lb, ub, b
