"""Randomised cross-check of the Gauss-Seidel sweeps (per-level launches, single-workgroup kernel, pipelined
single-workgroup kernel with mixed wide / narrow segments, with and without the LDS window) against the oracle's
sequential sweep, bit for bit.
python tools/fuzz_gs.py [--cases 120] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def random_system(rng):
    kind = rng.choice(["band", "grid", "random", "blocks"])
    if kind == "band":
        n = int(rng.randint(50, 40000))
        offs = sorted(set(int(o) for o in rng.randint(-40, 41, size=rng.randint(1, 8)) if o != 0))
        m = scipy.sparse.diags([rng.randn(n - abs(o)) for o in offs], offs, shape=(n, n), format="csr") if offs else scipy.sparse.csr_matrix((n, n))
    elif kind == "grid":
        s = int(rng.randint(8, 150))
        n = s * s
        ii = np.arange(n)
        rows = np.concatenate([ii[:-1], ii[1:], ii[:-s], ii[s:]])
        cols = np.concatenate([ii[1:], ii[:-1], ii[s:], ii[:-s]])
        keep = rng.rand(rows.size) < 0.9
        m = scipy.sparse.coo_matrix((rng.randn(keep.sum()), (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    elif kind == "random":
        n = int(rng.randint(1, 30000))
        k = int(n * rng.choice([0.5, 2, 6, 20]))
        m = scipy.sparse.coo_matrix((rng.randn(k), (rng.randint(0, n, k), rng.randint(0, n, k))), shape=(n, n)).tocsr()
    else:  # a wide first level (independent unknowns) followed by a banded tail, like the Potts normal matrix
        n0, n1 = int(rng.randint(3000, 30000)), int(rng.randint(500, 20000))
        n = n0 + n1
        tail = scipy.sparse.diags([rng.randn(n1 - 1), rng.randn(n1 - 1)], [-1, 1], shape=(n1, n1))
        k = 2 * n1  # about two links per tail row (scipy.sparse.random would permute all n1 * n0 cells)
        link = scipy.sparse.coo_matrix((rng.rand(k), (rng.randint(0, n1, size=k), rng.randint(0, n0, size=k))), shape=(n1, n0)).tocsr()
        link.sum_duplicates()
        m = scipy.sparse.bmat([[None, link.T], [link, tail]], format="csr")
    m = m.tocsr()
    m.sum_duplicates()
    m = (m + scipy.sparse.diags(np.abs(m).sum(axis=1).A1 + 1.0 + rng.rand(m.shape[0]))).tocsr()
    m.sort_indices()
    return kind, m


def run(cases, seed, verbose=False):
    from oracle import oracle
    from pysparselp_amd.gaussSiedel import boundedGaussSeidelClass

    rng = np.random.RandomState(seed)
    saved = os.environ.get("SLP_GS_PIPELINED")
    saved_w = os.environ.get("SLP_GS_WINDOW")
    tally = {}
    for case in range(cases):
        kind, m = random_system(rng)
        n = m.shape[0]
        force = rng.choice(["auto", "1", "0"])
        if force == "auto":
            os.environ.pop("SLP_GS_PIPELINED", None)
        else:
            os.environ["SLP_GS_PIPELINED"] = force
        if rng.rand() < 0.35:
            os.environ["SLP_GS_WINDOW"] = "0"
        else:
            os.environ.pop("SLP_GS_WINDOW", None)
        if rng.rand() < 0.3:
            os.environ["SLP_GS_SINKS"] = "0"  # earliest levels for every row (no last level of rows nothing waits for)
        else:
            os.environ.pop("SLP_GS_SINKS", None)
        bands = rng.choice(["auto", "auto", "0", "2", "3", "8", "16"])  # rows of a run of narrow levels over several workgroups
        if bands == "auto":
            os.environ.pop("SLP_GS_BANDS", None)
        else:
            os.environ["SLP_GS_BANDS"] = bands
        rhs = rng.randn(n)
        lo = np.where(rng.rand(n) < 0.3, -np.inf, -rng.rand(n))
        hi = np.where(rng.rand(n) < 0.3, np.inf, rng.rand(n))
        w, sweeps = float(rng.choice([1.0, 1.1, 0.7])), int(rng.choice([1, 2, 3]))
        x0 = rng.randn(n)
        xo = x0.copy()
        oracle.BoundedGaussSeidel(m).solve(rhs, lo, hi, xo, maxiter=sweeps, w=w)
        xg = x0.copy()
        g = boundedGaussSeidelClass(m)
        g.solve(rhs, lo, hi, xg, maxiter=sweeps, w=w)
        if verbose:
            print("case", case, kind, n, m.nnz, "levels", g.num_levels, "pipelined", force, "sweep kind", g.sweep_kind, flush=True)
        tally[(kind, force, g.sweep_kind, "bands" if g.num_bands else "")] = tally.get((kind, force, g.sweep_kind, "bands" if g.num_bands else ""), 0) + 1
        if not np.array_equal(xg, xo):
            raise AssertionError(f"Gauss-Seidel mismatch: case {case} {kind} n={n} nnz={m.nnz} pipelined={force} kind={g.sweep_kind} bands={bands}/{g.num_bands} "
                                 f"max diff {np.max(np.abs(xg - xo))}")
    os.environ.pop("SLP_GS_SINKS", None)
    os.environ.pop("SLP_GS_BANDS", None)
    for name, old in (("SLP_GS_PIPELINED", saved), ("SLP_GS_WINDOW", saved_w)):
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
    return tally


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=120)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--verbose", action="store_true")
    args = p.parse_args()
    tally = run(args.cases, args.seed, args.verbose)
    print("ok:", args.cases, "cases", {f"{k[0]}/{k[1]}/kind{k[2]}{k[3]}": v for k, v in sorted(tally.items())})


if __name__ == "__main__":
    main()
