"""Parity of the HIP path (through the C ABI) with the oracle and the golden
fixtures.  Needs a real MI355X: run with ``-m gpu``.

Bars (DESIGN.md "parity"):
* SEQUENTIAL / AUTO order on short rows: BIT-EXACT iterates (np.array_equal).
* TREE order (wavefront-parallel row sums): same terms in another association;
  |x_gpu - x_oracle| <= 1e-9 * (1 + |x|) per iterate over the tested horizons,
  objective within 1e-6 relative (the tolerance BASELINE.json's north_star states).
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse

from conftest import GOLDEN, Recorder, csr_of, lp_from_golden, load_golden, solver_args
from oracle import oracle
from test_oracle_golden import _reduced, kmedians_cost, l1svm_percent

pytestmark = pytest.mark.gpu

CASES = ["sc50a", "sc105", "potts8", "potts50", "random0", "random1", "random2"]
TREE_RTOL = 1e-9


def _mods():
    from pysparselp_amd import ORDER_AUTO, ORDER_SEQUENTIAL, ORDER_TREE
    from pysparselp_amd.ADMM import lp_admm
    from pysparselp_amd.ChambollePockPPD import chambolle_pock_ppd

    return lp_admm, chambolle_pock_ppd, ORDER_AUTO, ORDER_SEQUENTIAL, ORDER_TREE


# ------------------------------------------------------------------ kernels
def test_native_library_loaded_and_device_bound():
    from pysparselp_amd import _lib

    lib = _lib.lib()
    assert lib.slp_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libslp_hip.so" in f.read()


@pytest.mark.parametrize("order", [1, 2, 0])
def test_spmv_kat(order):
    from pysparselp_amd.device import DeviceMatrix

    d = load_golden("kernel_kats")
    a = csr_of(d, "spmv_A")
    dm = DeviceMatrix.from_csr(a)
    ax = dm.matvec(d["spmv_x"], order)
    ya = dm.rmatvec(d["spmv_y"], order)
    if order in (0, 1):  # short rows: sequential order -> the reference's bits
        assert np.array_equal(ax, d["spmv_Ax"]) and np.array_equal(ya, d["spmv_yA"])
    else:
        np.testing.assert_allclose(ax, d["spmv_Ax"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(ya, d["spmv_yA"], rtol=1e-13, atol=1e-13)
    # the device-built transpose lists every column's rows increasingly (scipy csr_tocsc order)
    t = dm.download(transposed=True)
    ref = scipy.sparse.csr_matrix(a.T.tocsr())
    cptr, crow, cdata = oracle.to_csc(oracle.as_csr(a))
    assert np.array_equal(t.indptr, cptr) and np.array_equal(t.indices, crow) and np.array_equal(t.data, cdata)
    assert t.shape == ref.shape


@pytest.mark.parametrize("shape,density", [((1, 1), 1.0), ((5, 7), 0.0), ((300, 200), 0.3), ((64, 5000), 0.2), ((3000, 70), 0.5)])
@pytest.mark.parametrize("order", [1, 2])
def test_spmv_shapes_against_oracle(shape, density, order):
    """Empty matrices, single entries, long rows (every lanes-per-row width), ragged rows."""
    from pysparselp_amd.device import DeviceMatrix

    rng = np.random.RandomState(3)
    a = scipy.sparse.random(shape[0], shape[1], density=density, random_state=rng, format="csr")
    a.data = rng.randn(a.nnz)
    if shape[0] > 10:  # ragged: empty some rows
        a = a.tolil()
        a[1::7, :] = 0
        a = a.tocsr()
        a.eliminate_zeros()
    x, y = rng.randn(shape[1]), rng.randn(shape[0])
    oa = oracle.as_csr(a)
    dm = DeviceMatrix.from_csr(a)
    ax, ya = dm.matvec(x, order), dm.rmatvec(y, order)
    if order == 1:
        assert np.array_equal(ax, oracle.matvec(oa, x)) and np.array_equal(ya, oracle.rmatvec(oa, y))
    else:
        scale_x = np.abs(a).dot(np.abs(x)) + 1e-300
        scale_y = np.abs(a).T.dot(np.abs(y)) + 1e-300
        assert np.max(np.abs(ax - oracle.matvec(oa, x)) / scale_x) < 1e-14 if a.nnz else True
        assert np.max(np.abs(ya - oracle.rmatvec(oa, y)) / scale_y) < 1e-14 if a.nnz else True


@pytest.mark.parametrize("tag", ["a", "b"])
def test_gauss_seidel_kat(tag):
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    d = load_golden("kernel_kats")
    m = csr_of(d, "gs_M")
    x = d[f"gs_{tag}_x0"].copy()
    bs = boundedGaussSeidelClass(m)
    ret = bs.solve(d["gs_b"], d["gs_lo"], d["gs_hi"], x, maxiter=int(d[f"gs_{tag}_iters"]), w=float(d[f"gs_{tag}_w"]))
    assert ret is x
    assert np.array_equal(x, d[f"gs_{tag}_x"])
    assert 1 < bs.num_levels <= m.shape[0]


@pytest.mark.parametrize("n,density", [(1, 1.0), (700, 0.004), (5000, 0.0008), (40000, 0.00005)])
def test_gauss_seidel_random_against_oracle(n, density):
    """Single-workgroup and one-launch-per-level paths, unsymmetric pattern, +-inf bounds."""
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    rng = np.random.RandomState(n)
    k = max(1, int(density * n * n))  # (scipy.sparse.random permutes all n * n cells: a minute for n = 40000)
    b0 = scipy.sparse.coo_matrix((np.ones(k), (rng.randint(0, n, size=k), rng.randint(0, n, size=k))), shape=(n, n)).tocsr()
    b0.sum_duplicates()
    b0.sort_indices()
    b0.data = rng.randn(b0.nnz)
    m = (b0 + scipy.sparse.diags(np.abs(b0).sum(axis=1).A1 + 1.0)).tocsr()  # unsymmetric, diagonally dominant
    rhs = rng.randn(n)
    lo = np.where(rng.rand(n) < 0.3, -np.inf, -rng.rand(n))
    hi = np.where(rng.rand(n) < 0.3, np.inf, rng.rand(n))
    x0 = rng.randn(n)
    xo = x0.copy()
    oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=2, w=1.1)
    xg = x0.copy()
    boundedGaussSeidelClass(m).solve(rhs, lo, hi, xg, maxiter=2, w=1.1)
    assert np.array_equal(xg, xo)


# ----------------------------------------------------------- solver iterates
@pytest.mark.parametrize("case", CASES)
def test_admm_iterates_bit_exact(case):
    lp_admm = _mods()[0]
    d = load_golden("lp_" + case)
    rec = Recorder(d["admm_it"])
    x = lp_admm(*solver_args(d), nb_iter=int(d["admm_it"][-1]), callback_func=rec, nb_iter_plot=1, order=_mods()[3])
    assert rec.it == list(d["admm_it"])
    assert np.array_equal(np.array(rec.x), d["admm_x"])
    assert np.array_equal(x, d["admm_x"][-1])
    assert np.array_equal(rec.veq, d["admm_veq"]) and np.array_equal(rec.vineq, d["admm_vineq"])
    np.testing.assert_allclose(rec.e1, d["admm_e1"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("case", CASES)
def test_cp_iterates_bit_exact(case):
    cp = _mods()[1]
    d = load_golden("lp_" + case)
    rec = Recorder(d["cp_it"])
    x, _ = cp(*_reduced(d), nb_max_iter=int(d["cp_it"][-1]) + 1, callback_func=rec, nb_iter_plot=1, order=_mods()[3])
    assert rec.it == list(d["cp_it"])
    assert np.array_equal(np.array(rec.x), d["cp_x"])
    assert np.array_equal(x, d["cp_x"][-1])
    assert np.array_equal(np.asarray(rec.veq, dtype=np.float64), d["cp_veq"]) and np.array_equal(rec.vineq, d["cp_vineq"])
    np.testing.assert_allclose(rec.e1, d["cp_e1"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rec.e2, d["cp_e2"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("case", ["sc105", "potts50", "random2"])
def test_reporting_cadence_does_not_change_iterates(case):
    """k iterations enqueued back to back (nb_iter_plot=500) == stepping one by one."""
    lp_admm, cp = _mods()[:2]
    d = load_golden("lp_" + case)
    last = int(d["admm_it"][-1])
    x = lp_admm(*solver_args(d), nb_iter=last, nb_iter_plot=500, order=_mods()[3])
    assert np.array_equal(x, d["admm_x"][-1])
    last = int(d["cp_it"][-1])
    x, _ = cp(*_reduced(d), nb_max_iter=last + 1, nb_iter_plot=500, order=_mods()[3])
    assert np.array_equal(x, d["cp_x"][-1])


@pytest.mark.parametrize("case", ["sc105", "potts50", "random0"])
def test_tree_order_within_tolerance(case):
    lp_admm, cp, _, _, tree = _mods()
    d = load_golden("lp_" + case)
    rec = Recorder(d["cp_it"])
    cp(*_reduced(d), nb_max_iter=int(d["cp_it"][-1]) + 1, callback_func=rec, nb_iter_plot=1, order=tree)
    for got, ref in zip(rec.x, d["cp_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < TREE_RTOL
    rec = Recorder(d["admm_it"])
    lp_admm(*solver_args(d), nb_iter=int(d["admm_it"][-1]), callback_func=rec, nb_iter_plot=1, order=tree)
    for got, ref in zip(rec.x, d["admm_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < TREE_RTOL


def test_cp_preconditioners_match_oracle():
    from pysparselp_amd.ChambollePockPPD import CPState, one_sided_system

    d = load_golden("lp_sc105")
    c, a_eq, beq, a_ineq, bl, bu, lb, ub = _reduced(d)
    ineq, b_ineq = one_sided_system(a_ineq, bl, bu)
    st = CPState(c, a_eq, beq, ineq, b_ineq, lb, ub, None, 1, 1)
    t, sigma = st.preconditioners()
    oa, ob = oracle.one_sided(a_ineq, bl, bu)
    rt, se, si = oracle.cp_setup(oracle.as_csr(a_eq), oa, 1)
    assert np.array_equal(t, rt) and np.array_equal(sigma, np.concatenate((se, si)))
    st.close()


# ------------------------------------------- the reference's own test suite, on the GPU
def _solve_curve(case, method, nb_iter):
    from pysparselp_amd.SparseLP import SparseLP

    d = load_golden("lp_" + case)
    lp = lp_from_golden(d, SparseLP)
    sol, elapsed = lp.solve(method=method, get_timing=True, nb_iter=nb_iter, max_time=None, ground_truth=d["gt"],
                            ground_truth_indices=d["gt_idx"], plot_solution=None, nb_iter_plot=500)
    return d, lp, sol


@pytest.mark.parametrize("case,ref_json,nb_iter", [("sc105", "ref_netlib_curves_SC105.json", 20000),
                                                   ("potts50", "ref_test_pott_segmentation_curves.json", 10000)])
@pytest.mark.parametrize("method", ["admm", "chambolle_pock_ppd"])
def test_reference_golden_curves(case, ref_json, nb_iter, method):
    """tests/test_netlib.py:96-125 and tests/test_pott_segmentation.py:20-37 through SparseLP.solve."""
    d, lp, sol = _solve_curve(case, method, nb_iter)
    ref = np.array(json.load(open(os.path.join(GOLDEN, ref_json)))[method])
    got = np.array(lp.distance_to_ground_truth)
    k = min(len(ref), len(got))
    assert k >= 20
    np.testing.assert_almost_equal(got[:k], ref[:k])
    tag = "admm" if method == "admm" else "cp"
    # everything SparseLP.solve records, against what the reference recorded for the same call
    assert np.array_equal(got, d[f"{tag}_curve_dist"])
    assert np.array_equal(lp.itrn_curve, d[f"{tag}_curve_itrn"])
    assert np.array_equal(lp.max_violated_constraint, d[f"{tag}_curve_maxviol"])
    np.testing.assert_allclose(lp.pobj_curve, d[f"{tag}_curve_pobj"], rtol=1e-9, atol=1e-9)
    assert np.array_equal(sol, d[f"{tag}_solution"])
    assert len(lp.opttime_curve) == len(got) == len(lp.dobj_curve) == len(lp.max_violated_equality)


def test_known_answer_kmedians():
    """tests/test_kmedians.py:11-14."""
    from pysparselp_amd.SparseLP import SparseLP

    d = load_golden("ka_kmedians")
    lp = lp_from_golden(d, SparseLP)
    s = lp.solve(method="admm", nb_iter=1000, max_time=np.inf, nb_iter_plot=500)[0]
    assert np.array_equal(s, d["solution"])
    assert kmedians_cost(s, d["points"]) == 238.9849948936172


def test_known_answer_l1svm():
    """tests/test_l1_svm.py:12-26 (free variables, lower-bound-only rows)."""
    from pysparselp_amd.SparseLP import SparseLP

    d = load_golden("ka_l1svm")
    lp = lp_from_golden(d, SparseLP)
    s, _ = lp.solve(method="admm", get_timing=True, nb_iter=2000, max_time=np.inf, plot_solution=None)
    assert np.array_equal(s, d["admm_solution"]) and l1svm_percent(s) == 99.5
    s, _ = lp.solve(method="chambolle_pock_ppd", get_timing=True, nb_iter=2000, max_time=np.inf, plot_solution=None)
    assert np.array_equal(s, d["cp_solution"]) and l1svm_percent(s) == 99.4


def test_max_time_stops_at_a_report():
    from pysparselp_amd.SparseLP import SparseLP

    d = load_golden("lp_sc50a")
    lp = lp_from_golden(d, SparseLP)
    lp.solve(method="admm", nb_iter=10 ** 7, max_time=0.2, nb_iter_plot=500)
    assert 1 <= len(lp.itrn_curve) < 10 ** 7 // 500 and lp.itrn_curve[-1] % 500 == 0


# ------------------------------------------------- synthetic benchmark problem
def test_random_lp_generator_properties():
    """Distribution of randomLP.rand_sparse; feasibility by construction; regenerable row blocks."""
    from pysparselp_amd.problems import random_lp_on_device

    n, m, p = 4000, 6000, 0.01
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=5)
    s = a.download()
    assert s.shape == (m, n) and s.has_sorted_indices
    assert abs(s.nnz / (m * n * p) - (1 - 0.004)) < 0.02          # Bernoulli(p) minus the ~0.4% exact zeros
    assert np.all(s.data != 0) and np.array_equal(np.round(s.data * 100) / 100, s.data)
    assert abs(np.std(s.data) - 1.0) < 0.02 and abs(np.mean(s.data)) < 0.01
    assert np.all(lb <= xf) and np.all(xf <= ub) and np.all(s @ xf <= b + 1e-9)   # feasible_x is feasible
    assert np.mean((b - s @ xf) < 1.001e-3) > 0.95                                 # most rows are tight to 1e-3
    cols = np.diff(s.tocsc().indptr)
    assert abs(cols.mean() - m * p * 0.996) < 1.0
    # a row block regenerated elsewhere (another GPU) is identical
    blk, _, c2, _, _, b2 = random_lp_on_device(n, m, p, seed=5, row_offset=1000, rows=500)
    sb = blk.download()
    assert (sb != s[1000:1500]).nnz == 0 and np.array_equal(b2, b[1000:1500]) and np.array_equal(c2, c)
    other = random_lp_on_device(n, m, p, seed=6, rows=10)[0].download()
    assert (other != s[:10]).nnz > 0


@pytest.mark.parametrize("n,m,p", [(4000, 900, 0.01), (300000, 700, 0.002), (70, 500, 0.5), (100000, 64, 0.00001)])
def test_random_lp_generator_wave_per_row_draws_the_same_matrix(monkeypatch, n, m, p):
    """The generator's two forms -- a thread per row, a wave per row (rows of many entries: coalesced writes) -- draw every
    number from (row, event) and must give the same arrays, whatever the row length (empty rows, several 64-event chunks,
    a chunk that ends exactly at the last column)."""
    from pysparselp_amd.device import DeviceMatrix

    out = []
    for wave in ("0", "1"):
        monkeypatch.setenv("SLP_RANDOM_WAVE", wave)
        a = DeviceMatrix.random(m, n, p, 9, 3)
        s = a.download()
        a.close()
        out.append(s)
    assert np.array_equal(out[0].indptr, out[1].indptr) and np.array_equal(out[0].indices, out[1].indices)
    assert np.array_equal(out[0].data, out[1].data) and out[0].nnz > 0


def test_cp_on_device_generated_lp_matches_oracle():
    """Long rows (TREE order, 64 lanes per row): iterates vs the oracle on the downloaded matrix,
    objective within 1e-6 relative (north_star's tolerance)."""
    from pysparselp_amd import _lib
    from pysparselp_amd.problems import random_lp_on_device

    n, m, p = 20000, 40000, 0.01
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=1)
    s = a.download()
    l = _lib.lib()
    h = _lib.check_handle(l.slp_cp_create_on(a._h, 0, _lib.ptr(b), _lib.ptr(c), _lib.ptr(lb), _lib.ptr(ub), None, 1.0, 1.0, 0))
    iters = 60
    _lib.check(l.slp_cp_iterate(h, iters))
    x = np.empty(n)
    _lib.check(l.slp_cp_get_x(h, _lib.ptr(x)))
    l.slp_cp_destroy(h)
    xo, _ = oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=iters, nb_iter_plot=10 ** 9)
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < TREE_RTOL
    assert abs(c.dot(x) - c.dot(xo)) <= 1e-6 * abs(c.dot(xo))


def test_spmv_linearity_at_scale():
    """Size-independent property at a size the oracle is not run on: A(ax + by) = a Ax + b Ay,
    and <A x, y> = <x, A^T y> (both orientations agree)."""
    from pysparselp_amd.problems import random_lp_on_device

    n, m, p = 200000, 300000, 0.001
    a = random_lp_on_device(n, m, p, seed=2)[0]
    rng = np.random.RandomState(0)
    x1, x2, y = rng.randn(n), rng.randn(n), rng.randn(m)
    lhs = a.matvec(2.0 * x1 - 3.0 * x2)
    rhs = 2.0 * a.matvec(x1) - 3.0 * a.matvec(x2)
    assert np.max(np.abs(lhs - rhs)) < 1e-10 * np.max(np.abs(rhs))
    assert abs(a.matvec(x1).dot(y) - x1.dot(a.rmatvec(y))) < 1e-9 * abs(a.matvec(x1).dot(y))


# ------------------------------------------------ ADMM with the matrix-free CG x-step
@pytest.mark.parametrize("case", ["sc50a", "sc105", "potts8", "potts50", "random0", "random1", "random2"])
@pytest.mark.parametrize("order", [1, 2])
def test_admm_cg_iterates(case, order):
    """Reference iterates of the conjugate-gradient branch (flag-flipped ADMM.py).  Tolerance, not bits:
    dot products are reduced in another order than BLAS does, and this variant amplifies rounding
    differences on some LPs (make_golden.py) -- 1e-9 relative over the first 200 iterations."""
    lp_admm = _mods()[0]
    d = load_golden("lp_" + case)
    keep = [it for it in d["admmcg_it"] if it <= 200]
    rec = Recorder(keep)
    lp_admm(*solver_args(d), nb_iter=200, callback_func=rec, nb_iter_plot=1, xstep="cg", order=order)
    assert rec.it == keep
    for got, ref in zip(rec.x, d["admmcg_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < TREE_RTOL
    for got, ref in zip(rec.e1, d["admmcg_e1"]):
        assert abs(got - ref) <= 1e-9 * (1 + abs(ref))
    np.testing.assert_allclose(rec.veq, d["admmcg_veq"][: len(keep)], rtol=1e-7, atol=1e-12)


def test_admm_cg_on_device_generated_lp_matches_oracle():
    """The at-scale path: device-side row normalisation + slack standard form + matrix-free ADMM,
    against the oracle run on the downloaded (unscaled) matrix; objective within 1e-6 relative."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device

    n, m, p = 3000, 5000, 0.02
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=4)
    s = a.download()
    iters = 40
    solver = DeviceADMM(a, b, c, lb, ub)
    solver.iterate(iters)
    x = solver.x(n)
    solver.close()
    xo = oracle.lp_admm_cg(c, None, None, s, None, b, lb, ub, nb_iter=iters - 1, nb_iter_plot=10 ** 9)
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < TREE_RTOL
    assert abs(c.dot(x) - c.dot(xo)) <= 1e-6 * abs(c.dot(xo))


@pytest.mark.parametrize("case", CASES)
def test_admm_unbounded_gauss_seidel_iterates_bit_exact(case):
    """xstep="gauss_seidel_unbounded": plain Gauss-Seidel + 1.4 over-relaxation + explicit lambda_ineq (ADMM.py:164-181,:252-256)."""
    lp_admm = _mods()[0]
    d = load_golden("lp_" + case)
    rec = Recorder(d["admmugs_it"])
    x = lp_admm(*solver_args(d), nb_iter=int(d["admmugs_it"][-1]), callback_func=rec, nb_iter_plot=1, order=_mods()[3],
                xstep="gauss_seidel_unbounded")
    assert rec.it == list(d["admmugs_it"])
    assert np.array_equal(np.array(rec.x), d["admmugs_x"])
    assert np.array_equal(x, d["admmugs_x"][-1])
    assert np.array_equal(rec.veq, d["admmugs_veq"]) and np.array_equal(rec.vineq, d["admmugs_vineq"])
    np.testing.assert_allclose(rec.e1, d["admmugs_e1"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("case", ["sc105", "potts50"])
def test_admm_unbounded_gauss_seidel_graph_replay(case):
    """Reporting every 7 iterations (graph-replayed stretches in between) gives the same iterates as reporting every one."""
    lp_admm = _mods()[0]
    d = load_golden("lp_" + case)
    last = int(d["admmugs_it"][-1])
    x = lp_admm(*solver_args(d), nb_iter=last, nb_iter_plot=7, order=_mods()[3], xstep="gauss_seidel_unbounded")
    assert np.array_equal(x, d["admmugs_x"][-1])


@pytest.mark.parametrize("name", ["AFIRO", "KB2", "SC50B"])
def test_netlib_files_through_reader_and_solvers(name):
    """The three netlib problems of the reference's data folder that have no iterate fixture: read with the MPS reader, built like
    tests/test_netlib.py:19-48, solved with both methods -- iterates against the oracle, bit for bit."""
    import os

    from conftest import GOLDEN
    from pysparselp_amd.netlib import get_problem
    from pysparselp_amd.SparseLP import SparseLP

    lp_admm, chambolle_pock_ppd, _, order, _ = _mods()
    d = get_problem(name, data_dir=os.path.join(GOLDEN, "netlib"))
    gt = d["solution"]
    lp = SparseLP()
    lp.add_variables_array(len(d["cost_vector"]), lower_bounds=d["lower_bounds"],
                           upper_bounds=np.minimum(d["upper_bounds"], np.max(gt) * 2), costs=d["cost_vector"])
    lp.add_equality_constraints_sparse(d["a_eq"], d["b_eq"])
    lp.add_inequality_constraints_sparse(d["a_ineq"], d["b_lower"], d["b_upper"])
    lp.convert_to_one_sided_inequality_system()
    args = (lp.costsvector, lp.a_equalities, lp.b_equalities, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds,
            lp.upper_bounds)
    x = lp_admm(*args, nb_iter=300, nb_iter_plot=100, order=order)
    xo = oracle.lp_admm(*args, nb_iter=300, nb_iter_plot=100)
    assert np.array_equal(x, xo)
    x, _ = chambolle_pock_ppd(*args, nb_max_iter=300, nb_iter_plot=100, order=order)
    xo, _ = oracle.chambolle_pock_ppd(*args, nb_max_iter=300, nb_iter_plot=100)
    assert np.array_equal(x, xo)
