"""BASELINE config 5 at its PER-RANK shape inside the suite: one block of the block-splitting ADMM
(ADMMBlocks.py:264-307 at scale) = 5e5 constraint rows over all 5e7 variables at density 1e-4 -- 2.5e9 stored entries, the share
of one of eight ranks of the 50M-variable LP (the reference's per-block sparse LU, ADMMBlocks.py:178-243, cannot run at this size:
parity here is kernel-level against the oracle plus the solver's own invariants).

  * ``A x`` through tall cells WITH the strip-range split (what ``bench.py --method admm_blocks`` switches on: 51 row blocks of 9984
    rows, 5 workgroups sharing a block's strips, partial sums added in range order) against the unsplit product (the single chain of
    the CSR sum): the rows hold 5000 entries, so re-associating their sums moves them by ~ sqrt(5000) ulps of the running sum
    (measured 6e-13 of 1 + |A x|): bar 5e-12; the unsplit product and ``A^T y`` bit for bit against the oracle on regenerated row slices /
    a slice-supported y; the adjoint identity;
  * the block solver: after EVERY block update the projection's true residual ``|| rhs - S nu ||`` is below the conjugate-gradient
    bar (1e-13 x ``|| rhs ||``; in this dual form that IS the constraint residual ``A z - z_s - b`` of the projected point);
    two runs bit-identical; a one-block group equals the single-block solver bit for bit.

Needs ~120 GB of device memory (skips below 200 GB free).  -m gpu."""
import os

import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu

N, M, P, SEED = 50_000_000, 500_000, 1e-4, 2


def _free_gb():
    from pysparselp_amd import _lib

    lib = _lib.lib()
    free, total = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
    _lib.check(lib.slp_device_memory(_lib.ptr(free), _lib.ptr(total)))
    return float(free[0]) / 1e9


@pytest.fixture(scope="module")
def shape():
    from pysparselp_amd import _lib
    from pysparselp_amd.problems import random_lp_on_device

    _lib.check(_lib.lib().slp_trim())
    if _free_gb() < 200:
        pytest.skip("less than 200 GB of device memory free")
    os.environ["SLP_TALL_SPLIT"] = "-1"
    a, xf, c, lb, ub, b = random_lp_on_device(N, M, P, seed=SEED)
    yield a, xf, c, lb, ub, b
    a.close()
    os.environ.pop("SLP_TALL_SPLIT", None)


@pytest.mark.timeout(1200)
def test_products_at_the_c5_per_rank_shape(shape):
    from pysparselp_amd.device import DeviceMatrix

    a, xf, c, lb, ub, b = shape
    assert a.nnz > 2 ** 31 and a.spmv_kernel(False) == 6 and a.spmv_kernel(True) == 6
    rng = np.random.RandomState(3)
    x, y = rng.randn(N), rng.randn(M)
    ax_split, aty = a.matvec(x), a.rmatvec(y)
    # the same rows without the split: a second copy of the matrix (counter-based generator), unsplit tall cells
    os.environ["SLP_TALL_SPLIT"] = "0"
    try:
        a2 = DeviceMatrix.random(M, N, P, SEED, 0)
        assert a2.spmv_kernel(False) == 6
        ax = a2.matvec(x)
        aty2 = a2.rmatvec(y)
    finally:
        os.environ["SLP_TALL_SPLIT"] = "-1"
    assert float(np.max(np.abs(ax_split - ax) / (1 + np.abs(ax)))) <= 5e-12       # re-associated partial sums of 5000-entry rows only
    assert np.array_equal(aty, aty2)                                                 # A^T: tall rows blocks, no split either way
    lhs, rhs = float(ax.dot(y)), float(x.dot(aty))
    assert abs(lhs - rhs) <= 1e-10 * (abs(lhs) + np.linalg.norm(ax) * np.linalg.norm(y))
    starts, rows = (0, M // 2, M - 512), 512
    slices, ysub = [], np.zeros(M)
    for r0 in starts:
        host = a2.download_rows(r0, rows)
        assert np.array_equal(oracle.matvec(oracle.as_csr(host), x), ax[r0:r0 + rows]), r0   # the single chain: the oracle's csr_matvec
        slices.append(host)
        ysub[r0:r0 + rows] = y[r0:r0 + rows]
    a2.close()
    want = oracle.rmatvec(oracle.as_csr(scipy.sparse.vstack(slices, format="csr")), np.concatenate([y[r0:r0 + rows] for r0 in starts]))
    assert np.array_equal(a.rmatvec(ysub), want)                                     # csc_matvec order, zero terms left out


@pytest.mark.timeout(1800)
def test_block_solver_invariants_at_the_c5_per_rank_shape(shape):
    from pysparselp_amd.scale import DeviceBlocks, DeviceBlocksGroup

    a, xf, c, lb, ub, b = shape
    runs = []
    for rep in range(2):
        s = DeviceBlocks(a, b, c, lb, ub)
        for it in range(3):
            s.iterate(1)
            res, rhs = s.projection_residual()
            assert rhs > 0 and res <= 1e-12 * rhs, (it, res, rhs)     # the CG bar (1e-13 on the recurrence): the TRUE residual within 10 x
        runs.append((s.x(), s.cg_steps()))
        s.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    assert np.all(np.isfinite(runs[0][0])) and np.all(runs[0][0] >= lb - 1e-12) and np.all(runs[0][0] <= ub + 1e-12)
    g = DeviceBlocksGroup(a, [0, M], b, c, lb, ub)   # a group of one block (a gathered copy of the rows) = the single-block solver
    g.iterate(3)
    xg = g.x()
    g.close()
    assert np.array_equal(xg, runs[0][0])
