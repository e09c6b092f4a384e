"""The single-workgroup, software-pipelined Gauss-Seidel sweeps (register ring of prefetched rows, a row's
accumulator handed from lane to lane; k_gs_sweep_windowed: entries classed at plan time, recent results from an
LDS ring, products with final values formed beforehand; k_gs_sweep_pipelined: every x gathered from memory)
against the oracle's sequential sweep, bit for bit.  The library picks them for systems with many narrow
dependency levels; SLP_GS_PIPELINED=1 forces them here on every size, SLP_GS_WINDOW=0 selects the second.  -m gpu."""
import numpy as np
import pytest
import scipy.sparse

from conftest import Recorder, load_golden, solver_args
from oracle import oracle

pytestmark = pytest.mark.gpu

CASES = ["sc50a", "sc105", "potts8", "potts50", "random0", "random1", "random2"]


@pytest.fixture(params=["window", "gather", "bands"])
def pipelined(monkeypatch, request):
    monkeypatch.setenv("SLP_GS_PIPELINED", "1")
    if request.param == "gather":
        monkeypatch.setenv("SLP_GS_WINDOW", "0")
    if request.param == "bands":  # the windowed kernel on three workgroups wherever a run allows it (tests/test_gpu_gs_bands.py)
        monkeypatch.setenv("SLP_GS_BANDS", "3")
    yield 2 if request.param == "gather" else 3
    monkeypatch.delenv("SLP_GS_PIPELINED", raising=False)
    monkeypatch.delenv("SLP_GS_WINDOW", raising=False)
    monkeypatch.delenv("SLP_GS_BANDS", raising=False)


@pytest.mark.parametrize("n,density", [(1, 1.0), (700, 0.004), (700, 0.03), (700, 0.25), (5000, 0.0008), (40000, 0.00005), (3000, 0.4)])
def test_sweep_random_against_oracle(pipelined, n, density):
    """Rows of 1 ... 64 entries: one lane or up to 16 chained lanes (with longer rows -- the last two cases -- the library
    keeps the per-level kernels); levels wider than 1024 lane slots (several steps per level); +-inf bounds; w != 1; two sweeps."""
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    rng = np.random.RandomState(n)
    k = max(1, int(density * n * n))  # (scipy.sparse.random permutes all n * n cells: a minute for n = 40000)
    b0 = scipy.sparse.coo_matrix((np.ones(k), (rng.randint(0, n, size=k), rng.randint(0, n, size=k))), shape=(n, n)).tocsr()
    b0.sum_duplicates()
    b0.sort_indices()
    b0.data = rng.randn(b0.nnz)
    m = (b0 + scipy.sparse.diags(np.abs(b0).sum(axis=1).A1 + 1.0)).tocsr()
    rhs = rng.randn(n)
    lo = np.where(rng.rand(n) < 0.3, -np.inf, -rng.rand(n))
    hi = np.where(rng.rand(n) < 0.3, np.inf, rng.rand(n))
    x0 = rng.randn(n)
    xo = x0.copy()
    oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=2, w=1.1)
    xg = x0.copy()
    gs = boundedGaussSeidelClass(m)
    gs.solve(rhs, lo, hi, xg, maxiter=2, w=1.1)
    assert np.array_equal(xg, xo)
    if np.diff(m.indptr).max() <= 64:  # (a level with a longer row goes to the per-level kernel)
        assert gs.sweep_kind == pipelined


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("xstep,key", [("gauss_seidel", "admm"), ("gauss_seidel_unbounded", "admmugs")])
def test_admm_iterates_bit_exact(pipelined, case, xstep, key):
    from pysparselp_amd._lib import ORDER_SEQUENTIAL
    from pysparselp_amd.ADMM import lp_admm

    d = load_golden("lp_" + case)
    rec = Recorder(d[key + "_it"])
    x = lp_admm(*solver_args(d), nb_iter=int(d[key + "_it"][-1]), callback_func=rec, nb_iter_plot=1, order=ORDER_SEQUENTIAL, xstep=xstep)
    assert rec.it == list(d[key + "_it"])
    assert np.array_equal(np.array(rec.x), d[key + "_x"])
    assert np.array_equal(x, d[key + "_x"][-1])


def test_potts_grid_takes_the_pipelined_sweep_by_default():
    """Potts 96 x 96: ~190 levels of ~330 rows -- chosen without the override; 12 ADMM iterations against the oracle."""
    from pysparselp_amd.ADMM import lp_admm
    from pysparselp_amd.problems import potts_lp

    lp = potts_lp(96)[0]
    args = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    x = lp_admm(*args, nb_iter=12, nb_iter_plot=5)
    xo = oracle.lp_admm(*args, nb_iter=12, nb_iter_plot=5)
    assert np.array_equal(x, xo)


def test_band_and_grid_systems_pick_the_windowed_sweep(monkeypatch):
    """Without overrides: a 5-point grid (every dependency one level back: LDS ring only) and a band matrix with offsets up to
    37 (dependencies up to 37 levels back: ring, plus gathers for the ones older than three levels) take the windowed kernel;
    three sweeps, w = 1, against the oracle; the gather form gives the same bits."""
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    rng = np.random.RandomState(11)
    s = 300
    n = s * s
    ii = np.arange(n)
    rows = np.concatenate([ii[:-1], ii[1:], ii[:-s], ii[s:]])
    cols = np.concatenate([ii[1:], ii[:-1], ii[s:], ii[:-s]])
    grid = scipy.sparse.coo_matrix((rng.randn(rows.size), (rows, cols)), shape=(n, n)).tocsr()
    nb = 30000
    offs = [-37, -5, -1, 1, 2, 9, 37]
    band = scipy.sparse.diags([rng.randn(nb - abs(o)) for o in offs], offs, shape=(nb, nb), format="csr")
    for m0 in (grid, band):
        m = (m0 + scipy.sparse.diags(np.abs(m0).sum(axis=1).A1 + 1.0)).tocsr()
        m.sort_indices()
        k = m.shape[0]
        rhs, x0 = rng.randn(k), rng.randn(k)
        lo = np.where(rng.rand(k) < 0.3, -np.inf, -rng.rand(k))
        hi = np.where(rng.rand(k) < 0.3, np.inf, rng.rand(k))
        xo = x0.copy()
        oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=3, w=1.0)
        out = []
        for window in (None, "0"):
            if window is None:
                monkeypatch.delenv("SLP_GS_WINDOW", raising=False)
            else:
                monkeypatch.setenv("SLP_GS_WINDOW", window)
            gs = boundedGaussSeidelClass(m)
            assert gs.sweep_kind == (3 if window is None else 2)
            xg = x0.copy()
            gs.solve(rhs, lo, hi, xg, maxiter=3, w=1.0)
            out.append(xg)
        assert np.array_equal(out[0], xo) and np.array_equal(out[1], xo)


def test_randomised_systems_all_sweep_variants():
    """tools/fuzz_gs.py: banded, grid, random and wide-first-level systems; per-level launches, single-workgroup and pipelined
    sweeps (forced and chosen); w in {0.7, 1, 1.1}; 1-3 sweeps; +-inf bounds -- bit for bit against the oracle."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gs

    tally = fuzz_gs.run(30, seed=3)
    assert sum(tally.values()) == 30
