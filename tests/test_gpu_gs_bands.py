"""Gauss-Seidel bands (slp_admm.hip, kGsFetchK): a run of narrow dependency levels cut into row ranges, one workgroup (compute
unit) each, a fetch wave per workgroup carrying the lower bands' results in through LDS -- every row's arithmetic is the
single-workgroup kernel's, so the sweep stays bit for bit the oracle's sequential one.  SLP_GS_BANDS=P forces P bands where a run
allows them (0: never); without it the plan's timing model decides (Potts 256^2: yes).  -m gpu."""
import os

import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu


def _systems():
    rng = np.random.RandomState(5)
    s = 150
    n = s * s
    ii = np.arange(n)
    keep = (ii[:-1] + 1) % s != 0  # a proper 5-point grid: no link from the end of one grid row to the start of the next
    rows = np.concatenate([ii[:-1][keep], ii[1:][keep], ii[:-s], ii[s:]])
    cols = np.concatenate([ii[1:][keep], ii[:-1][keep], ii[s:], ii[:-s]])
    yield "grid", scipy.sparse.coo_matrix((rng.randn(rows.size), (rows, cols)), shape=(n, n)).tocsr()
    nb = 20000
    offs = [-37, -5, -1, 1, 2, 9, 37]   # dependencies up to 37 levels back: ring, far gathers inside a band, external values
    yield "band", scipy.sparse.diags([rng.randn(nb - abs(o)) for o in offs], offs, shape=(nb, nb), format="csr")
    k = 40000                            # random pattern: external values from many lower bands at once
    r = scipy.sparse.coo_matrix((rng.randn(k), (rng.randint(0, 9000, k), rng.randint(0, 9000, k))), shape=(9000, 9000)).tocsr()
    r.sum_duplicates()
    yield "random", r


@pytest.mark.parametrize("bands", ["2", "5", "16"])
def test_forced_bands_match_the_sequential_sweep(monkeypatch, bands):
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    monkeypatch.setenv("SLP_GS_PIPELINED", "1")
    monkeypatch.setenv("SLP_GS_BANDS", bands)
    rng = np.random.RandomState(int(bands))
    used = 0
    for name, m0 in _systems():
        m = (m0 + scipy.sparse.diags(np.abs(m0).sum(axis=1).A1 + 1.0)).tocsr()
        m.sort_indices()
        k = m.shape[0]
        rhs, x0 = rng.randn(k), rng.randn(k)
        lo = np.where(rng.rand(k) < 0.3, -np.inf, -rng.rand(k))
        hi = np.where(rng.rand(k) < 0.3, np.inf, rng.rand(k))
        for w, sweeps in ((1.0, 1), (1.1, 3)):
            xo = x0.copy()
            oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=sweeps, w=w)
            gs = boundedGaussSeidelClass(m)
            assert gs.sweep_kind == 3
            used += gs.num_bands > 0
            for _ in range(2):  # (a second solve on the same plan: the counters are reset per run)
                xg = x0.copy()
                gs.solve(rhs, lo, hi, xg, maxiter=sweeps, w=w)
                assert np.array_equal(xg, xo), (name, bands, w, gs.num_bands)
    assert used >= 4  # (a run may refuse bands: more than 128 external values in one level of a band)


def test_potts_admm_with_bands_is_bit_exact_and_chosen_at_256(monkeypatch):
    """Potts 64 x 64 with 4 bands forced through both x-steps against the oracle; at 256 x 256 the plan takes bands by itself."""
    from pysparselp_amd.ADMM import lp_admm
    from pysparselp_amd.problems import potts_lp

    lp = potts_lp(64)[0]
    args = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    monkeypatch.setenv("SLP_GS_PIPELINED", "1")
    monkeypatch.setenv("SLP_GS_BANDS", "4")
    assert np.array_equal(lp_admm(*args, nb_iter=25, nb_iter_plot=7), oracle.lp_admm(*args, nb_iter=25, nb_iter_plot=7))
    assert np.array_equal(lp_admm(*args, nb_iter=25, nb_iter_plot=7, xstep="gauss_seidel_unbounded"),
                          oracle.lp_admm_gs_unbounded(*args, nb_iter=25, nb_iter_plot=7))
    monkeypatch.delenv("SLP_GS_PIPELINED")
    monkeypatch.delenv("SLP_GS_BANDS")
    from pysparselp_amd.ADMM import ADMMState

    lp = potts_lp(256)[0]
    args = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    x = lp_admm(*args, nb_iter=6, nb_iter_plot=3)
    assert np.array_equal(x, oracle.lp_admm(*args, nb_iter=6, nb_iter_plot=3))
    st = ADMMState.from_lp(*args, None, 2, 3)
    assert st.num_levels() > 500 and st.num_bands() >= 4
    st.close()


@pytest.mark.parametrize("bands", ["3", "8", "16"])
def test_store_waiting_bands_equal_the_default_ones_over_many_sweeps(monkeypatch, bands):
    """by default a band publishes a level as stored once a LATER vector-memory operation of the storing
    wave has completed (in-order completion of a wave's vector-memory operations: a documented precondition, DESIGN.md
    section 3).  ``SLP_GS_BANDS_SAFE=1`` (read when the solver is created) makes every wave wait for its stores themselves
    (``s_waitcnt vmcnt(0)``) before it publishes: both forms must give the oracle's sweep bit for bit -- many sweeps, several
    seeds, 3 / 8 / 16 workgroups (8 and 16 land on different XCDs: the dispatcher deals workgroups round-robin over the 8 XCDs)."""
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    monkeypatch.setenv("SLP_GS_PIPELINED", "1")
    monkeypatch.setenv("SLP_GS_BANDS", bands)
    used = 0
    for seed in (0, 1, 2):
        rng = np.random.RandomState(100 * int(bands) + seed)
        for name, m0 in _systems():
            m = (m0 + scipy.sparse.diags(np.abs(m0).sum(axis=1).A1 + 1.0)).tocsr()
            m.sort_indices()
            k = m.shape[0]
            rhs, x0 = rng.randn(k), rng.randn(k)
            lo = np.where(rng.rand(k) < 0.3, -np.inf, -rng.rand(k))
            hi = np.where(rng.rand(k) < 0.3, np.inf, rng.rand(k))
            xo = x0.copy()
            oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=12, w=1.05)
            got = {}
            for safe in ("0", "1"):
                monkeypatch.setenv("SLP_GS_BANDS_SAFE", safe)
                gs = boundedGaussSeidelClass(m)
                used += gs.num_bands > 0
                xg = x0.copy()
                for _ in range(4):   # 4 x 3 sweeps on one plan
                    gs.solve(rhs, lo, hi, xg, maxiter=3, w=1.05)
                got[safe] = xg
            assert np.array_equal(got["0"], got["1"]), (name, bands, seed)
            assert np.array_equal(got["1"], xo), (name, bands, seed)
    assert used >= 8
