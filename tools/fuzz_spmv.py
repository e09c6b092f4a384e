"""Randomised cross-check of every SpMV kernel family against the oracle's sequential row sums (bit for bit):
CSR (sequential order), fp64 strips, value-dictionary pairs / quads, wide strips -- shapes around the block and strip
boundaries, empty rows / columns, both orientations.   python tools/fuzz_spmv.py [--cases 150] [--seed 0]"""
import argparse
import faulthandler
import os
import sys

import numpy as np
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(cases, seed, verbose=False):
    from oracle import oracle
    from pysparselp_amd import _lib
    from pysparselp_amd.device import DeviceMatrix

    rng = np.random.RandomState(seed)
    lib = _lib.lib()
    seen = {}
    edges_r = [1, 2, 63, 64, 1023, 1024, 2047, 2048, 2049, 4095, 4096, 4097, 6000]
    edges_c = [1, 7, 3967, 3968, 3969, 5887, 5888, 5889, 7679, 7680, 7681, 12000, 131071, 131072, 131073, 300000]
    saved = {k: os.environ.get(k) for k in ("SLP_STRIP_MIN_NNZ", "SLP_VALUE_DICT", "SLP_DICT_VARIANT", "SLP_TALL", "SLP_TALL_R")}
    for case in range(cases):
        nrow = int(rng.choice(edges_r)) if rng.rand() < 0.7 else int(rng.randint(1, 9000))
        ncol = int(rng.choice(edges_c)) if rng.rand() < 0.7 else int(rng.randint(1, 20000))
        if rng.rand() < 0.2:
            ncol = int(rng.choice([262145, 300000, 400000]))  # wide-strip regime
        per_row = rng.choice([0.5, 3, 12, 40])
        dens = min(1.0, per_row * (1 + (ncol > 100000) * 8) / ncol)
        k = max(1, int(dens * nrow * ncol))  # (scipy.sparse.random permutes nrow * ncol cells: far too slow for wide shapes)
        a = scipy.sparse.coo_matrix((np.ones(k), (rng.randint(0, nrow, size=k), rng.randint(0, ncol, size=k))), shape=(nrow, ncol)).tocsr()
        a.sum_duplicates()
        a.sort_indices()
        mode = rng.choice(["dict", "quads", "fp64", "csr"])
        if mode in ("dict", "quads"):
            vals = np.round(rng.randn(int(rng.choice([1, 3, 200, 2048]))), 3)
            a.data = rng.choice(vals, size=a.nnz)
        else:
            a.data = rng.randn(a.nnz)
        if rng.rand() < 0.3 and nrow > 3:   # a band of empty rows
            lo = rng.randint(0, nrow - 2)
            a = scipy.sparse.vstack([a[:lo], scipy.sparse.csr_matrix((min(5, nrow - lo), ncol)), a[lo + min(5, nrow - lo):]]).tocsr()
        os.environ["SLP_STRIP_MIN_NNZ"] = "100000000000" if mode == "csr" else "1"
        os.environ["SLP_VALUE_DICT"] = "0" if mode == "fp64" else "1"
        os.environ["SLP_DICT_VARIANT"] = "2" if mode == "quads" else "1"
        # tall cells (sparse rows over many strips) take precedence over the wide strips: rule them out now and then, and
        # vary the height of their row blocks
        os.environ["SLP_TALL"] = "0" if (case % 2 if ncol > 100000 else rng.rand() < 0.2) else "1"
        os.environ["SLP_TALL_R"] = str(int(rng.choice([1024, 1500, 4096, 9984])))
        if verbose:
            print("case", case, mode, nrow, ncol, a.nnz, flush=True)
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(40, exit=True)  # a stuck case shows where
        dm = DeviceMatrix.from_csr(a)
        x, y = rng.randn(ncol), rng.randn(nrow)
        oa = oracle.as_csr(a)
        ok = np.array_equal(dm.matvec(x, 1), oracle.matvec(oa, x)) and np.array_equal(dm.rmatvec(y, 1), oracle.rmatvec(oa, y))
        kinds = (lib.slp_matrix_spmv_kernel(dm._h, 0), lib.slp_matrix_spmv_kernel(dm._h, 1))
        for k in kinds:
            seen[k] = seen.get(k, 0) + 1
        dm.close()
        if not ok:
            raise AssertionError(f"SpMV mismatch: case {case} mode {mode} shape {nrow} x {ncol} nnz {a.nnz} kernels {kinds}")
    faulthandler.cancel_dump_traceback_later()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return seen


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=150)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--verbose", action="store_true")
    args = p.parse_args()
    seen = run(args.cases, args.seed, args.verbose)
    print("ok:", args.cases, "cases; kernel codes used (0 CSR, 1 fp64 strips, 2 pairs, 3 quads, 4/5 wide, 6/7 tall cells):", dict(sorted(seen.items())))


if __name__ == "__main__":
    main()
