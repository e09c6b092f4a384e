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
from .device import ChunkedDeviceMatrix, DeviceMatrix


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


def random_lp_on_device(n, m, density, seed=0, row_offset=0, rows=None, chunks=1, chunked=False, columns=True, m_eq=0):
    """Synthetic random LP ``min c.x  s.t.  A x <= b_upper, lb <= x <= ub`` (all inequalities; ``m_eq`` > 0: the 10 %-equality
    variant of randomLP.py:62-68 -- the first ``m_eq`` of the generated rows are equalities ``a_i x = b_i``, ``b[:m_eq] = A_e
    feasible_x``; a chunked matrix is then cut at ``m_eq`` (even), so that the two kinds of rows live in chunks of their own and
    Chambolle-Pock forms ``(c + y_eq * a_eq) + y_ineq * a_ineq`` from two products over the chunks' copies).

    Generates rows ``row_offset .. row_offset + rows`` (default: all ``m``) of the
    m x n matrix directly in HBM.  Returns ``(DeviceMatrix, feasible_x, c, lb, ub, b_upper)``;
    ``feasible_x`` satisfies every constraint by construction (randomLP.py:33,43-46,53-55).

    ``chunks > 1``: the same LP as a ``ChunkedDeviceMatrix`` -- the rows are generated, converted and released ``chunks``
    row ranges at a time, so the CSR of the whole matrix never exists (the generator is keyed by the global row: every
    chunking draws the same matrix; ``b_upper`` is taken once all chunks stand, from the chunked matrix's own product).
    ``chunked=True``: a ``ChunkedDeviceMatrix`` even for one chunk -- the rows are converted into their product copies and the
    CSR is released at once (a row block of the block-splitting ADMM at BASELINE config 5: only ever multiplied).
    ``columns=False``: only ``b_upper`` is formed (``feasible_x, c, lb, ub`` are ``None``): the column vectors do not depend on
    the row range, a caller building several row blocks of one LP asks for them once.
    """
    rows = m if rows is None else rows
    assert 0 <= row_offset and row_offset + rows <= m
    assert 0 <= m_eq <= rows
    if chunks <= 1 and not chunked:
        a = DeviceMatrix.random(rows, n, density, seed, row_offset)
        xf, c, lb, ub, b = a.random_lp_vectors(density, seed, row_offset, columns=columns, m_eq=m_eq)
        return a, xf, c, lb, ub, b
    cuts = ChunkedDeviceMatrix.cuts(rows, max(1, chunks), cut_at=m_eq)
    a = ChunkedDeviceMatrix(n, expect_chunks=len(cuts) - 1, expect_rows=rows)
    for r0, r1 in zip(cuts, cuts[1:]):
        a.append(DeviceMatrix.random(r1 - r0, n, density, seed, row_offset + r0))
    # b_upper = ceil((A x_f + ...) 1000) / 1000 of all rows at once, through the product copies (one launch of the fused product)
    xf, c, lb, ub, b = a.random_lp_vectors(density, seed, row_offset, columns=columns, m_eq=m_eq)
    return a, xf, c, lb, ub, b


def kmedians_lp(points, k, n_center_candidates, seed=0):
    """LP relaxation of k-medians (reference examples/example_kmedians.py:15-44).

    Variables: ``labeling[i, j]`` in [0, 1] (point i served by candidate j, cost = distance) and
    ``used[j]`` in [0, 1].  Constraints: ``0 <= sum_j used[j] <= k``; ``sum_j labeling[i, j] = 1`` for every
    point; ``labeling[i, j] - used[j] <= 0``.  Returns ``(lp, labeling_indices, pairdistances)``.
    """
    n = points.shape[0]
    rng = np.random.RandomState(seed)
    centers = points[rng.choice(n, n_center_candidates), :]
    pairdistances = np.sqrt(np.sum((points[:, None, :] - centers[None, :, :]) ** 2, axis=2))
    lp = SparseLP()
    labeling = lp.add_variables_array(pairdistances.shape, 0, 1, pairdistances)
    used = lp.add_variables_array(n_center_candidates, 0, 1, 0)
    lp.add_inequality_constraints(used[None, :], np.ones((1, n_center_candidates)), lower_bounds=0, upper_bounds=k)
    lp.add_inequality_constraints(labeling, np.ones((n, n_center_candidates)), lower_bounds=1, upper_bounds=1)
    cols = np.column_stack((labeling.reshape(-1, 1), np.tile(used, n).reshape(-1, 1)))
    vals = np.column_stack((np.ones(n * n_center_candidates), -np.ones(n * n_center_candidates)))
    lp.add_inequality_constraints(cols, vals, lower_bounds=None, upper_bounds=0)
    return lp, labeling, pairdistances


def l1svm_lp(x, classes, nb_classes=None):
    """L1-regularised multi-class SVM as an LP (reference examples/example_l1_svm.py:10-68).

    Variables: weights (classes x (features + 1), free), one |w| auxiliary per weight (>= 0, cost 1), one slack
    per example (>= 0, cost 1).  Constraints: ``+-w - aux <= 0`` and, for every class k and every example not
    of class k, ``(w_class(e) - w_k) . [x_e, 1] + eps_e >= 1``.  Returns ``(lp, weight_indices, eps_indices)``.
    """
    nb_examples, nb_features = x.shape
    xh = np.hstack((x, np.ones((nb_examples, 1))))
    if nb_classes is None:
        nb_classes = int(np.max(classes)) + 1
    lp = SparseLP()
    w = lp.add_variables_array((nb_classes, nb_features + 1), None, None)
    aux = lp.add_variables_array(w.size, upper_bounds=None, lower_bounds=0)
    lp.set_costs_variables(aux, np.ones(aux.shape))
    cols = np.column_stack((w.ravel(), aux.ravel()))
    lp.add_inequality_constraints(cols, np.tile([1.0, -1.0], (w.size, 1)), lower_bounds=None, upper_bounds=0)
    lp.add_inequality_constraints(cols, np.tile([-1.0, -1.0], (w.size, 1)), lower_bounds=None, upper_bounds=0)
    eps = lp.add_variables_array((nb_examples, 1), upper_bounds=None, lower_bounds=0, costs=1)
    margin = np.ones((nb_examples, nb_classes))
    margin[np.arange(nb_examples), classes] = 0
    own = w[classes, :]
    for k in range(nb_classes):
        keep = classes != k
        other = np.tile(w[[k], :], (nb_examples, 1))
        vals = np.column_stack((xh, -xh, np.ones(eps.shape)))
        cols = np.column_stack((own, other, eps))
        lp.add_inequality_constraints(cols[keep, :], vals[keep, :], lower_bounds=margin[keep, k], upper_bounds=None)
    return lp, w, eps
