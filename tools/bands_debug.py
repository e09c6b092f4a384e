"""Lab: Gauss-Seidel bands on simple systems against the oracle, with where the mismatches are.  python tools/bands_debug.py"""
import os
import sys

import numpy as np
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def systems():
    rng = np.random.RandomState(0)
    n = 6000
    yield "tridiagonal", scipy.sparse.diags([rng.randn(n - 1), rng.randn(n - 1)], [-1, 1], shape=(n, n), format="csr")
    yield "band +-1,+-7", scipy.sparse.diags([rng.randn(n - 7), rng.randn(n - 1), rng.randn(n - 1), rng.randn(n - 7)], [-7, -1, 1, 7], shape=(n, n), format="csr")
    s = 96
    n = s * s
    ii = np.arange(n)
    rows = np.concatenate([ii[:-1], ii[1:], ii[:-s], ii[s:]])
    cols = np.concatenate([ii[1:], ii[:-1], ii[s:], ii[:-s]])
    yield "grid 96", scipy.sparse.coo_matrix((rng.randn(rows.size), (rows, cols)), shape=(n, n)).tocsr()


def main():
    from oracle import oracle
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    os.environ["SLP_GS_PIPELINED"] = "1"
    for name, m in systems():
        m = m.tocsr()
        m = (m + scipy.sparse.diags(np.abs(m).sum(axis=1).A1 + 1.5)).tocsr()
        m.sort_indices()
        n = m.shape[0]
        rng = np.random.RandomState(1)
        rhs, lo, hi, x0 = rng.randn(n), -rng.rand(n) - 5, rng.rand(n) + 5, rng.randn(n)
        xo = x0.copy()
        oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=1, w=1.0)
        for bands in os.environ.get("BANDS_LIST", "0,2,4,16").split(","):
            os.environ["SLP_GS_BANDS"] = bands
            g = boundedGaussSeidelClass(m)
            bad_runs = []
            for rep in range(3):
                xg = x0.copy()
                g.solve(rhs, lo, hi, xg, maxiter=1, w=1.0)
                bad = np.flatnonzero(xg != xo)
                bad_runs.append((bad.size, bad[:6].tolist(), bad[-3:].tolist()))
            print(name, "bands", bands, "->", g.num_bands, "levels", g.num_levels, "mismatches (count, first, last) per repeat:", bad_runs, flush=True)


if __name__ == "__main__":
    main()
