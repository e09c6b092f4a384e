"""The Gauss-Seidel sweep's plan built ON THE DEVICE (csrc/slp_gs_plan_device.h): dependency levels, level
order, bands, lane-slot records and fetch lists from the matrix where it lies -- ``slp_admm_create_lp`` no longer downloads ``M``.
Reference: gaussSiedel.pyx:87-92 (the reference's constructor only inverts a diagonal) and :131-152 (the sweep whose order the plan
preserves).  ``SLP_GS_PLAN=check`` builds BOTH plans and requires every array the sweep reads -- level order, permuted matrix,
inverted diagonal, entry / lane records, header lists, band tables, fetch lists, requirement rows -- to be equal byte for byte;
on top of that the sweep's result against the oracle bit for bit.  -m gpu."""
import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu


def _system(n, k, rng, symmetric):
    b0 = scipy.sparse.coo_matrix((np.ones(k), (rng.randint(0, n, size=k), rng.randint(0, n, size=k))), shape=(n, n)).tocsr()
    if symmetric:
        b0 = (b0 + b0.T).tocsr()
    b0.sum_duplicates()
    b0.sort_indices()
    b0.data = rng.randn(b0.nnz)
    return (b0 + scipy.sparse.diags(np.abs(b0).sum(axis=1).A1 + 1.0)).tocsr()


def _sweep_equals_oracle(m, rng, sweeps=2, w=1.1):
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    n = m.shape[0]
    rhs, x0 = rng.randn(n), rng.randn(n)
    lo = np.where(rng.rand(n) < 0.3, -np.inf, -rng.rand(n))
    hi = np.where(rng.rand(n) < 0.3, np.inf, rng.rand(n))
    xo = x0.copy()
    oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=sweeps, w=w)
    gs = boundedGaussSeidelClass(m)     # (check mode: raises if any array of the device plan differs from the host plan's)
    xg = x0.copy()
    gs.solve(rhs, lo, hi, xg, maxiter=sweeps, w=w)
    assert np.array_equal(xg, xo)
    return gs


@pytest.mark.parametrize("forced", ["0", "1"])
@pytest.mark.parametrize("n,k,symmetric", [(1, 1, False), (2, 1, False), (700, 2000, False), (700, 15000, True), (5000, 20000, True),
                                           (40000, 80000, False), (3000, 3600000, False), (60000, 100000, True)])
def test_device_plan_equals_the_host_plan_on_random_systems(monkeypatch, n, k, symmetric, forced):
    """Unsymmetric patterns (the levels walk M and M^T), rows of 1 ... 1200 entries (levels with a long row are swept chip-wide),
    levels wider than 1024 lane slots, both the default choice of sweep and the single-workgroup sweep forced on every size."""
    monkeypatch.setenv("SLP_GS_PLAN", "check")
    monkeypatch.setenv("SLP_GS_PIPELINED", forced) if forced == "1" else monkeypatch.delenv("SLP_GS_PIPELINED", raising=False)
    rng = np.random.RandomState(n + k)
    _sweep_equals_oracle(_system(n, k, rng, symmetric), rng)


@pytest.mark.parametrize("bands", ["0", "3", "8", "16", None])
def test_device_plan_equals_the_host_plan_with_bands(monkeypatch, bands):
    """A grid-like system (the normal matrix of a Potts LP): runs of narrow levels, with the bands forced to 3 / 8 / 16, forbidden,
    or chosen by the timing model -- band ranges, external references, requirement rows and fetch lists all come from the device."""
    from pysparselp_amd.problems import potts_lp
    from pysparselp_amd.tools import convert_to_standard_form_with_bounds, normal_matrix, precondition_constraints

    monkeypatch.setenv("SLP_GS_PLAN", "check")
    if bands is not None:
        monkeypatch.setenv("SLP_GS_BANDS", bands)
    lp, _, _, _ = potts_lp(96)
    n = lp.costsvector.size
    a, bu = precondition_constraints(lp.a_inequalities, lp.b_upper)
    _, a2, b2, _, _, _ = convert_to_standard_form_with_bounds(lp.costsvector, None, None, a, None, bu, lp.lower_bounds, lp.upper_bounds, np.zeros(n))
    a2, _ = precondition_constraints(a2, b2)
    m = normal_matrix(a2, 2.0, 3.0).tocsr()
    m.sort_indices()
    gs = _sweep_equals_oracle(m, np.random.RandomState(7), sweeps=3, w=1.0)
    assert gs.sweep_kind == 3
    if bands in ("3", "8", "16"):
        assert gs.num_bands > 0
    if bands == "0":
        assert gs.num_bands == 0


def test_whole_lp_admm_setup_without_a_download_of_m(monkeypatch):
    """lp_admm on Potts 64 x 64 in check mode (the whole chain of slp_admm_create_lp on the device, M planned where it lies) against
    the oracle's iterates bit for bit; then the same with the host plan: the same x."""
    from pysparselp_amd.ADMM import lp_admm
    from pysparselp_amd.problems import potts_lp

    lp, _, _, _ = potts_lp(64)
    args = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    want = oracle.lp_admm(*args, nb_iter=12, nb_iter_plot=10 ** 9)
    for mode in ("check", "device", "host"):
        monkeypatch.setenv("SLP_GS_PLAN", mode)
        assert np.array_equal(lp_admm(*args, nb_iter=12, nb_iter_plot=10 ** 9), want), mode


def test_long_chains_and_the_round_one_kernel_fall_back_to_the_host_plan(monkeypatch):
    """A tridiagonal system has n levels of one row: past 65 536 levels of < 32 rows the device plan hands over to the host's single
    pass over the rows; SLP_GS_WINDOW=0 (the round-1 sweep kernel) is planned on the host too.  Same results either way."""
    monkeypatch.setenv("SLP_GS_PLAN", "device")
    n = 70000
    m = scipy.sparse.diags([np.full(n - 1, -1.0), np.full(n, 4.0), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsr()
    m.sort_indices()
    gs = _sweep_equals_oracle(m, np.random.RandomState(3), sweeps=1, w=1.0)
    assert gs.num_levels == n
    monkeypatch.setenv("SLP_GS_WINDOW", "0")
    rng = np.random.RandomState(11)
    _sweep_equals_oracle(_system(5000, 20000, rng, True), rng)
