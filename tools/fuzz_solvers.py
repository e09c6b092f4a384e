"""Randomised cross-check of the host-API solvers against the oracle, bit for bit (sequential summation order):
lp_admm with its three x-steps' Gauss-Seidel forms, chambolle_pock_ppd and, within tolerance, the matrix-free and the
block-splitting ADMM -- random small LPs with empty rows / columns, one- and two-sided rows, infinite bounds,
with and without equality rows, warm starts and odd reporting cadences.   python tools/fuzz_solvers.py [--cases 60]"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def random_lp(rng):
    n = int(rng.randint(2, 70))
    me = int(rng.choice([0, 0, 1, 5, 12]))
    mi = int(rng.randint(1, 90))
    dens = float(rng.choice([0.05, 0.2, 0.5]))
    ae = scipy.sparse.random(me, n, density=dens, random_state=rng, format="lil")
    ai = scipy.sparse.random(mi, n, density=dens, random_state=rng, format="lil")
    if rng.rand() < 0.4:
        ai[rng.randint(0, mi), :] = 0
    if rng.rand() < 0.4:
        j = rng.randint(0, n)
        ai[:, j] = 0
        ae[:, j] = 0
    ae, ai = ae.tocsr(), ai.tocsr()
    ae.eliminate_zeros()
    ai.eliminate_zeros()
    ae.data = np.round(rng.randn(ae.nnz) * 100) / 100 + 0.005
    ai.data = np.round(rng.randn(ai.nnz) * 100) / 100 + 0.005
    ae.__dict__["blocks"] = [(0, me - 1)] if me else []
    cut = int(rng.randint(0, mi))
    ai.__dict__["blocks"] = [(0, mi - 1)] if cut in (0, mi - 1) or rng.rand() < 0.5 else [(0, cut), (cut + 1, mi - 1)]
    xf = np.round(rng.randn(n) * 100) / 100
    be = ae @ xf
    bu = ai @ xf + rng.rand(mi)
    bl = ai @ xf - rng.rand(mi)
    mode = rng.choice(["upper", "two", "mixed"])
    if mode == "upper":
        bl = None
    elif mode == "mixed":
        bl[rng.rand(mi) < 0.4] = -np.inf
        drop = rng.rand(mi) < 0.3
        bu[drop & np.isfinite(bl)] = np.inf
    c = np.round(rng.randn(n) * 100) / 100
    t = np.abs(rng.randn(n)) + 0.1
    lb, ub = xf - t, xf + t
    lb[rng.rand(n) < 0.2] = -np.inf
    ub[rng.rand(n) < 0.2] = np.inf
    x0 = None if rng.rand() < 0.5 else np.round(rng.randn(n), 2)
    return c, (ae if me else None), (be if me else None), ai, bl, bu, lb, ub, x0


def run(cases, seed, verbose=False):
    from oracle import oracle
    from pysparselp_amd import ORDER_SEQUENTIAL
    from pysparselp_amd.ADMM import lp_admm
    from pysparselp_amd.ADMMBlocks import lp_admm_block_decomposition
    from pysparselp_amd.ChambollePockPPD import chambolle_pock_ppd

    rng = np.random.RandomState(seed)
    for case in range(cases):
        c, ae, be, ai, bl, bu, lb, ub, x0 = random_lp(rng)
        args = (c, ae, be, ai, bl, bu, lb, ub)
        its, plot = int(rng.randint(1, 60)), int(rng.choice([1, 7, 10, 10 ** 9]))
        if verbose:
            print("case", case, "n", c.size, "me", 0 if ae is None else ae.shape[0], "mi", ai.shape[0], "bl", bl is not None, its, plot, flush=True)
        got, ref = [], []
        x = lp_admm(*args, x0=x0, nb_iter=its, nb_iter_plot=plot, order=ORDER_SEQUENTIAL, callback_func=lambda i, s, *r: got.append((i, s.copy())))
        xo = oracle.lp_admm(*args, x0=x0, nb_iter=its, nb_iter_plot=plot, callback_func=lambda i, s, *r: ref.append((i, s.copy())))
        assert np.array_equal(x, xo) and [g[0] for g in got] == [r[0] for r in ref], f"admm case {case}"
        assert all(np.array_equal(g[1], r[1]) for g, r in zip(got, ref)), f"admm reports case {case}"
        x = lp_admm(*args, x0=x0, nb_iter=its, nb_iter_plot=plot, order=ORDER_SEQUENTIAL, xstep="gauss_seidel_unbounded")
        xo = oracle.lp_admm_gs_unbounded(*args, x0=x0, nb_iter=its, nb_iter_plot=plot)
        assert np.array_equal(x, xo), f"admm unbounded-GS case {case}"
        if bl is None or np.all(np.isfinite(bu) | np.isfinite(bl)):  # (the reference's stacking needs a finite side per row)
            cae = ae if ae is not None else scipy.sparse.csr_matrix((0, c.size))
            cbe = be if be is not None else np.zeros(0)
            x, _ = chambolle_pock_ppd(c, cae, cbe, ai, bl, bu, lb, ub, x0=x0, nb_max_iter=its, nb_iter_plot=plot, order=ORDER_SEQUENTIAL)
            xo, _ = oracle.chambolle_pock_ppd(c, cae, cbe, ai, bl, bu, lb, ub, x0=x0, nb_max_iter=its, nb_iter_plot=plot)
            assert np.array_equal(x, xo), f"chambolle-pock case {case}"
        x = lp_admm(*args, x0=x0, nb_iter=min(its, 30), nb_iter_plot=plot, xstep="cg")
        xo = oracle.lp_admm_cg(*args, x0=x0, nb_iter=min(its, 30), nb_iter_plot=plot)
        # One CG step per iteration does not contract on every LP (e.g. 2 variables under 68 two-sided rows: the iterates
        # jump around and a last-bit difference grows tenfold every 2-3 iterations, in the oracle as much as here).  The
        # oracle run again with c moved in its last bits measures that amplification; the tolerance follows it.
        xo2 = oracle.lp_admm_cg(*((args[0] * (1 + 4e-16),) + args[1:]), x0=x0, nb_iter=min(its, 30), nb_iter_plot=plot)
        if not np.all(np.isfinite(xo)):  # e.g. no stored entry at all and x0 feasible: the first CG step is 0 / 0, in the reference too
            assert not np.all(np.isfinite(x)), f"admm-cg case {case}: the oracle's iterate is not finite, the device's is"
            continue
        tol = max(1e-8, 1e3 * float(np.max(np.abs(xo2 - xo) / (1 + np.abs(xo)))))
        assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < tol, f"admm-cg case {case}: {np.max(np.abs(x - xo))} (tolerance {tol})"
        if all(np.diff(ai.indptr)[lo:hi + 1].sum() > 0 for lo, hi in ai.blocks):  # (a block without entries has no KKT system)
            try:
                xo = oracle.lp_admm_block_decomposition(*args, x0=x0, nb_iter=min(its, 25), nb_iter_plot=10 ** 9)
            except RuntimeError:   # SuperLU: a singular KKT matrix (dependent rows inside a block) -- no reference result
                continue
            if not np.all(np.isfinite(xo)):
                continue
            x = lp_admm_block_decomposition(*args, x0=x0, nb_iter=min(its, 25), nb_iter_plot=10 ** 9)
            assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-6, f"admm-blocks case {case}: {np.max(np.abs(x - xo))}"
    return cases


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=60)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--verbose", action="store_true")
    args = p.parse_args()
    print("ok:", run(args.cases, args.seed, args.verbose), "cases")


if __name__ == "__main__":
    main()
