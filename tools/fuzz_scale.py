"""Randomised cross-check of the at-scale (device-resident) solvers against the oracle on mid-size device-generated LPs:
DeviceCP and DeviceADMM (reuse levels 0 ... 4) over every strip variant and over tall cells (both item forms), with equality
rows and two-sided rows.  DeviceBlocks is checked against the generic split-matrix block solver (the oracle's sparse LU takes minutes
beyond a few thousand rows; both are checked against it in tests/test_admm_blocks.py on small LPs).
python tools/fuzz_scale.py [--cases 16] [--seed 0]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(cases, seed, verbose=False):
    from oracle import oracle
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    rng = np.random.RandomState(seed)
    keys = ("SLP_STRIP_MIN_NNZ", "SLP_VALUE_DICT", "SLP_DICT_VARIANT", "SLP_TALL_R")
    saved = {k: os.environ.get(k) for k in keys}
    for case in range(cases):
        n = int(rng.choice([9000, 20000, 33000]))
        m = int(rng.choice([7000, 25000, 41000]))
        p = float(rng.choice([0.0008, 0.002]))
        fmt = rng.choice(["quads", "pairs", "fp64", "csr", "tall", "tall_fp64"])
        if fmt.startswith("tall"):  # rows sparse inside every 4096-column window, in both orientations: tall cells (slp_tall.hip)
            n = int(rng.choice([90000, 150000]))
            m = int(rng.choice([12000, 25000]))
            p = float(rng.choice([1.5e-4, 3e-4]))
            os.environ["SLP_TALL_R"] = str(int(rng.choice([1024, 2000, 9984])))
        os.environ["SLP_STRIP_MIN_NNZ"] = "100000000000" if fmt == "csr" else "1"
        os.environ["SLP_VALUE_DICT"] = "0" if fmt in ("fp64", "tall_fp64") else "1"
        os.environ["SLP_DICT_VARIANT"] = "2" if fmt == "quads" else "1"
        m_eq = int(rng.choice([0, 0, m // 10]))
        two_sided = rng.rand() < 0.5
        level = int(rng.choice([0, 2, 3, 4]))
        a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=int(rng.randint(0, 1000)))
        if fmt.startswith("tall"):
            want = 6 if fmt == "tall" else 7
            assert a.spmv_kernel(False) == want and a.spmv_kernel(True) == want, (fmt, a.spmv_kernel(False), a.spmv_kernel(True))
        s = a.download()
        ax = a.matvec(xf)
        b = b.copy()
        b[:m_eq] = ax[:m_eq]
        bl = np.where(rng.rand(m) < 0.4, -np.inf, ax - rng.rand(m)) if two_sided else None
        ae, ai = (s[:m_eq], s[m_eq:]) if m_eq else (None, s)
        be = b[:m_eq] if m_eq else None
        bli = None if bl is None else bl[m_eq:]
        if verbose:
            print("case", case, n, m, p, fmt, "m_eq", m_eq, "two_sided", two_sided, "reuse", level, flush=True)
        its = int(rng.randint(5, 40))
        cp = DeviceCP(a, b, c, lb, ub, m_eq=m_eq, b_lower=bl)
        cp.iterate(its)
        x = cp.x()
        cp.close()
        import scipy.sparse
        cae = ae if ae is not None else scipy.sparse.csr_matrix((0, n))
        cbe = be if be is not None else np.zeros(0)
        xo, _ = oracle.chambolle_pock_ppd(c, cae, cbe, ai, bli, b[m_eq:], lb, ub, nb_max_iter=its, nb_iter_plot=10 ** 9)
        assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-10, f"cp case {case}: {np.max(np.abs(x - xo))}"
        if m_eq == 0:  # two independent implementations of the block-splitting ADMM with ONE block: the row-block solver
            # (implicit slack, primal or dual projection) against the generic split-matrix solver of the host API
            from pysparselp_amd.ADMMBlocks import lp_admm_block_decomposition
            from pysparselp_amd.scale import DeviceBlocks

            blk = DeviceBlocks(a, b, c, lb, ub, b_lower=bl, cg_tol=1e-14)
            blk.iterate(6)
            x = blk.x()
            blk.close()
            ai_b = scipy.sparse.csr_matrix(ai)
            ai_b.__dict__["blocks"] = [(0, m - 1)]
            xh = lp_admm_block_decomposition(c, None, None, ai_b, bli, b, lb, ub, nb_iter=5, nb_iter_plot=10 ** 9, cg_tol=1e-14)
            assert np.max(np.abs(x - xh) / (1 + np.abs(xh))) < 1e-8, f"blocks case {case}: {np.max(np.abs(x - xh))}"
        ad = DeviceADMM(a, b, c, lb, ub, m_eq=m_eq, b_lower=bl, reuse=level)   # last: may scale the matrix in place
        ad.iterate(its)
        x = ad.x(n)
        ad.close()
        a.close()
        xo = oracle.lp_admm_cg(c, ae, be, ai, bli, b[m_eq:], lb, ub, nb_iter=its - 1, nb_iter_plot=10 ** 9)
        assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-9, f"admm case {case}: {np.max(np.abs(x - xo))}"
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return cases


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=16)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--verbose", action="store_true")
    args = p.parse_args()
    print("ok:", run(args.cases, args.seed, args.verbose), "cases")


if __name__ == "__main__":
    main()
