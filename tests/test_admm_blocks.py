"""Block-splitting ADMM (reference ADMMBlocks.py): the oracle against iterates of the reference
(tests/golden/make_blocks_golden.py), the host-side copies layout, and -- on the GPU -- the matrix-free
device solver against the same iterates."""
import os

import numpy as np
import pytest
import scipy.sparse

from conftest import GOLDEN, Recorder, load_golden, solver_args
from oracle import oracle

CASES = ["sc50a", "sc105", "potts8", "potts50", "random0", "random1"]


def _blocks(case):
    d = np.load(os.path.join(GOLDEN, "admm_blocks.npz"))
    return d, [tuple(r) for r in d[f"{case}_blocks_eq"]], [tuple(r) for r in d[f"{case}_blocks_ineq"]]


def _with_blocks(a, blocks):
    if a is None:
        return None
    a = scipy.sparse.csr_matrix(a.tocsr() if hasattr(a, "tocsr") else a)
    a.__dict__["blocks"] = list(blocks)
    return a


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_iterates(case):
    d, beq, bineq = _blocks(case)
    args = solver_args(load_golden("lp_" + case))
    rec = Recorder(d[f"{case}_it"])
    oracle.lp_admm_block_decomposition(*args, nb_iter=200, nb_iter_plot=1, callback_func=rec, blocks_eq=beq, blocks_ineq=bineq)
    assert rec.it == list(d[f"{case}_it"])
    # bit-identical in the build container (same SuperLU); 1e-12 leaves room for another scipy build on the GPU box
    for got, ref in zip(rec.x, d[f"{case}_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < 1e-12
    np.testing.assert_allclose(rec.e1, d[f"{case}_e1"], rtol=1e-10, atol=1e-10)
    assert float(d[f"{case}_oracle_vs_reference"]) == 0.0


@pytest.mark.parametrize("case", ["potts8", "potts50"])
def test_oracle_matrix_free_form_reproduces_reference_iterates(case):
    """``oracle.lp_admm_blocks_cg`` -- the CPU restatement of the matrix-free (conjugate-gradient) per-block projection, the form
    that exists at BASELINE config 5 and what bench.py times as the cpu_baseline of ``--method admm_blocks`` -- against the
    reference's sparse-LU iterates on the all-inequality fixtures: the CG bar (1e-13 on the residual) leaves <= 1e-9."""
    d, beq, bineq = _blocks(case)
    c, a_eq, be, a_ineq, bl, bu, lb, ub = solver_args(load_golden("lp_" + case))
    assert a_eq is None and not beq
    a = scipy.sparse.csr_matrix(a_ineq)
    blocks = [(a[lo:hi + 1], None if bl is None else bl[lo:hi + 1], bu[lo:hi + 1]) for lo, hi in bineq]
    ref = {int(i): v for i, v in zip(d[f"{case}_it"], d[f"{case}_x"])}
    for primal in (False, True):
        got = {}
        xp, steps = oracle.lp_admm_blocks_cg(c, blocks, lb, ub, nb_iter=max(ref) + 1, primal=primal,
                                             iterate_hook=lambda i: None)
        # the hook sees no iterate: re-run up to every recorded iteration is too slow; the final iterate and two early ones
        for it in sorted(ref)[:2]:
            got[it], _ = oracle.lp_admm_blocks_cg(c, blocks, lb, ub, nb_iter=it + 1, primal=primal)
        got[max(ref)] = xp
        assert steps > 0
        for it, x in got.items():
            assert np.max(np.abs(x - ref[it]) / (1 + np.abs(ref[it]))) < 1e-9, (case, primal, it)


def test_copies_layout():
    from pysparselp_amd.ADMMBlocks import split_by_blocks
    from pysparselp_amd.tools import CsrArrays

    rng = np.random.RandomState(0)
    a = scipy.sparse.random(12, 9, density=0.35, random_state=rng, format="csr")
    a.sort_indices()
    a.data[3] = 0.0  # an explicit zero stays an entry unless its whole column is zero inside the block
    blocks = [(0, 3), (4, 8), (9, 11)]
    a_split, owner, cptr, cidx = split_by_blocks(CsrArrays.from_any(a), blocks)
    dense, off = a.toarray(), 0
    big = a_split.tocsr().toarray()
    for lo, hi in blocks:
        sub = dense[lo:hi + 1]
        ids = np.nonzero(np.abs(sub).sum(axis=0))[0]
        assert np.array_equal(owner[off:off + ids.size], ids)
        assert np.array_equal(big[lo:hi + 1, off:off + ids.size], sub[:, ids])
        assert not big[lo:hi + 1, :off].any() and not big[lo:hi + 1, off + ids.size:].any()  # block diagonal
        off += ids.size
    assert a_split.shape == (12, off) and cptr[-1] == off
    for j in range(9):
        copies = cidx[cptr[j]:cptr[j + 1]]
        assert np.all(owner[copies] == j) and np.all(np.diff(copies) > 0)
    with pytest.raises(ValueError):
        split_by_blocks(CsrArrays.from_any(a), [(0, 3), (3, 11)])
    with pytest.raises(ValueError):
        split_by_blocks(CsrArrays.from_any(a), [(0, 3)])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_iterates_match_reference(case):
    """Matrix-free per-block projections (conjugate gradients to 1e-13) against the reference's sparse-LU iterates."""
    from pysparselp_amd.ADMMBlocks import lp_admm_block_decomposition

    d, beq, bineq = _blocks(case)
    c, a_eq, be, a_ineq, bl, bu, lb, ub = solver_args(load_golden("lp_" + case))
    rec = Recorder(d[f"{case}_it"])
    x = lp_admm_block_decomposition(c, _with_blocks(a_eq, beq), be, _with_blocks(a_ineq, bineq), bl, bu, lb, ub, nb_iter=200,
                                    nb_iter_plot=1, callback_func=rec)
    assert rec.it == list(d[f"{case}_it"])
    for got, ref in zip(rec.x, d[f"{case}_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < 1e-8
    assert np.array_equal(x, rec.x[-1])
    np.testing.assert_allclose(rec.e1, d[f"{case}_e1"], rtol=1e-7, atol=1e-7)


@pytest.mark.gpu
def test_gpu_reporting_cadence_and_solve_method():
    """nb_iter_plot = 7: reports after iterations 0, 7, 14, ...; SparseLP.solve(method="admm_blocks") end to end."""
    from pysparselp_amd.ADMMBlocks import lp_admm_block_decomposition
    from pysparselp_amd.problems import potts_lp

    d, beq, bineq = _blocks("potts8")
    c, a_eq, be, a_ineq, bl, bu, lb, ub = solver_args(load_golden("lp_potts8"))
    rec = Recorder()
    lp_admm_block_decomposition(c, None, None, _with_blocks(a_ineq, bineq), bl, bu, lb, ub, nb_iter=20, nb_iter_plot=7, callback_func=rec)
    assert rec.it == [0, 7, 14]
    ref = {int(i): v for i, v in zip(d["potts8_it"], d["potts8_x"])}
    assert np.max(np.abs(rec.x[0] - ref[0])) < 1e-9
    lp, gt, gt_idx, _ = potts_lp(8)
    x = lp.solve(method="admm_blocks", nb_iter=200, nb_iter_plot=50, ground_truth=gt, ground_truth_indices=gt_idx, get_timing=False)
    assert np.max(np.abs(x - ref[200])) < 1e-7 and lp.itrn_curve == [0, 50, 100, 150, 200]


@pytest.mark.gpu
@pytest.mark.parametrize("m_eq", [300, 0])  # with equality rows: dual form (A A^T + I); all inequalities, m >= n: primal form (I + A^T A)
@pytest.mark.parametrize("min_nnz", ["1", "100000000000"])  # strip kernels / CSR kernels
def test_gpu_row_block_solver_matches_oracle(monkeypatch, min_nnz, m_eq):
    """DeviceBlocks: the rows of a device-generated LP as ONE block (what a rank holds in the multi-GPU form), equality rows
    and two-sided rows included, slack column implicit -- against the oracle's LU form with the single block [0, m - 1].
    (The partitioned code path with the consensus all-reduce is exercised in tests/test_gpu_comm.py.)"""
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceBlocks

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", min_nnz)
    n, m, p = 3000, 4000, 0.004
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=5)
    s = a.download()
    ax = a.matvec(xf)
    rng = np.random.RandomState(8)
    b = b.copy()
    b[:m_eq] = ax[:m_eq]
    bl = np.where(rng.rand(m) < 0.5, -np.inf, ax - rng.rand(m))
    ae, ai = s[:m_eq], s[m_eq:]
    xo = oracle.lp_admm_block_decomposition(c, ae if m_eq else None, b[:m_eq] if m_eq else None, ai, bl[m_eq:], b[m_eq:], lb, ub,
                                            nb_iter=29, nb_iter_plot=10 ** 9, blocks_eq=[(0, m - 1)], blocks_ineq=[])
    sol = DeviceBlocks(a, b, c, lb, ub, m_eq=m_eq, b_lower=bl)
    sol.iterate(30)
    x = sol.x()
    assert 30 < sol.cg_steps() < 30 * 500
    sol.close()
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-8
    # Jacobi-preconditioned conjugate gradients: the same projection (to the CG tolerance), hence the same iterates
    sol = DeviceBlocks(a, b, c, lb, ub, m_eq=m_eq, b_lower=bl, jacobi=True)
    sol.iterate(30)
    assert np.max(np.abs(sol.x() - x) / (1 + np.abs(x))) < 1e-9 and 30 < sol.cg_steps() < 30 * 500
    sol.close()
    if m_eq == 0:  # the two forms of the projection give the same iterates
        monkeypatch.setenv("SLP_BLOCKS_PRIMAL", "0")
        sol = DeviceBlocks(a, b, c, lb, ub, m_eq=0, b_lower=bl)
        sol.iterate(30)
        assert np.max(np.abs(sol.x() - x) / (1 + np.abs(x))) < 1e-9
        sol.close()
    a.close()


@pytest.mark.gpu
def test_gpu_short_very_wide_row_block_with_the_strip_range_split(monkeypatch):
    """Config 5's shape in small (few rows, very many columns): `A x` runs on tall cells whose strips are shared by several
    workgroups (SLP_TALL_SPLIT, what `bench.py --method admm_blocks` switches on) -- partial row sums added in a fixed order, not
    the single chain; the block solver's projections carry a CG tolerance, so the iterates agree with the oracle's LU form as
    without the split, and two runs agree bit for bit."""
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceBlocks

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", "1")
    monkeypatch.setenv("SLP_TALL_SPLIT", "3")
    n, m, p = 400001, 1500, 3e-5   # 12 entries per row over 98 strips of 4096 columns
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=11)
    assert a.spmv_kernel(False) == 6
    s = a.download()
    xo = oracle.lp_admm_block_decomposition(c, None, None, s, None, b, lb, ub, nb_iter=11, nb_iter_plot=10 ** 9, blocks_eq=[(0, m - 1)],
                                            blocks_ineq=[])
    out = []
    for _ in range(2):
        sol = DeviceBlocks(a, b, c, lb, ub)
        sol.iterate(12)
        out.append(sol.x())
        sol.close()
    a.close()
    assert np.array_equal(out[0], out[1])
    assert np.max(np.abs(out[0] - xo) / (1 + np.abs(xo))) < 1e-8


@pytest.mark.gpu
def test_several_row_blocks_on_one_rank_match_the_reference_block_decomposition():
    """DeviceBlocksGroup: three uneven row blocks of a device-resident LP, each its own copy of the variables; against the
    oracle's restatement of lp_admm_block_decomposition (sparse LU per block) with the same row ranges as blocks."""
    from oracle import oracle
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceBlocks, DeviceBlocksGroup

    n, m = 800, 1500
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, 0.01, seed=9)
    s = a.download()
    cuts = [0, 400, 1100, 1500]
    grp = DeviceBlocksGroup(a, cuts, b, c, lb, ub)
    grp.iterate(40)
    x_grp = grp.x()
    assert grp.cg_steps() > 0
    # linking is not idempotent (it replaces every block's copy counts by the group totals): a second link is refused
    from pysparselp_amd import _lib
    with pytest.raises(_lib.SlpError, match="already part of a group"):
        _lib.check(grp._l.slp_blocks_group_link(grp._handles, len(grp.blocks)))
    grp.close()
    x_ref = oracle.lp_admm_block_decomposition(c, None, None, s, None, b, lb, ub, nb_iter=39, nb_iter_plot=10 ** 9, blocks_eq=[],
                                               blocks_ineq=[(lo, hi - 1) for lo, hi in zip(cuts, cuts[1:])])
    assert np.max(np.abs(x_grp - x_ref) / (1 + np.abs(x_ref))) < 1e-8
    # a group of one block is the single-block solver
    one = DeviceBlocksGroup(a, [0, m], b, c, lb, ub)
    one.iterate(10)
    single = DeviceBlocks(a, b, c, lb, ub)
    single.iterate(10)
    assert np.array_equal(one.x(), single.x())
    one.close()
    single.close()
    a.close()


def test_blocks_handed_over_as_callables_are_loaded_per_update():
    """``oracle.lp_admm_blocks_cg`` with blocks given as callables (how tools/c5_oracle_parity.py --stream-blocks keeps one 30 GB block
    of BASELINE config 5 in host memory at a time): the same iterates and conjugate-gradient step counts as with resident blocks, bit
    for bit; every block loaded once for the set-up and once per iteration."""
    import scipy.sparse

    from oracle import oracle

    rng = np.random.RandomState(0)
    n, blocks = 300, []
    for g in range(3):
        a = scipy.sparse.random(80, n, 0.05, format="csr", random_state=rng)
        a.sort_indices()
        a.data = np.round(a.data * 10) / 10 + 0.05
        blocks.append((a, None, a @ rng.rand(n) + 0.5))
    c, lb, ub = rng.randn(n), -np.ones(n), np.ones(n)
    want, steps = oracle.lp_admm_blocks_cg(c, blocks, lb, ub, nb_iter=4)
    loads = []
    lazy = [((lambda a=a, g=g: (loads.append(g), a)[1]), bl, bu) for g, (a, bl, bu) in enumerate(blocks)]
    got, steps2 = oracle.lp_admm_blocks_cg(c, lazy, lb, ub, nb_iter=4)
    assert np.array_equal(got, want) and steps == steps2
    assert loads == [0, 1, 2] * 5
