"""Randomised cross-check of CHUNKED matrices (csrc/slp_chunked.hip) against the oracle's sequential sums, bit for bit: random
shapes in the tall-cell regime (density 2e-5 .. 5e-4: both orientations of every chunk on tall cells) and in the LDS-strip regime
(8e-4 .. 8e-3: dictionary pairs / quads or fp64 strips), random chunkings (2 .. 6 chunks, even inner cuts, one chunk sometimes a few
rows only), empty rows, forced row-block heights, with and without a value dictionary; `A x`, `A^T y` (the continuation of the column
sums from chunk to chunk) and a few Chambolle-Pock iterations against the oracle.  A case whose chunks do not qualify for strip
copies in both orientations is counted as skipped.      python tools/fuzz_chunked.py [--cases 120] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(cases, seed, verbose=False):
    from oracle import oracle
    from pysparselp_amd import _lib
    from pysparselp_amd._lib import SlpError
    from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix
    from pysparselp_amd.scale import DeviceCP

    rng = np.random.RandomState(seed)
    lib = _lib.lib()
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    seen, skipped, solved = {}, 0, 0
    for case in range(cases):
        tall = rng.rand() < 0.6
        if tall:
            nrow, ncol = int(rng.randint(9000, 50000)), int(rng.choice([50001, 131072, 200000, 262145, 300001]))
            dens = float(np.exp(rng.uniform(np.log(3e-5), np.log(4e-4))))
        else:
            nrow, ncol = int(rng.randint(9000, 40000)), int(rng.choice([5888, 7680, 12000, 20000, 23553]))
            dens = float(np.exp(rng.uniform(np.log(1e-3), np.log(6e-3))))
        k = max(1, int(dens * nrow * ncol))
        a = scipy.sparse.coo_matrix((np.ones(k), (rng.randint(0, nrow, size=k), rng.randint(0, ncol, size=k))), shape=(nrow, ncol)).tocsr()
        a.sum_duplicates()
        a.sort_indices()
        dictionary = rng.rand() < 0.7
        a.data = rng.choice(np.round(rng.randn(int(rng.choice([2, 40, 1100]))), 2) + 0.005, size=a.nnz) if dictionary else rng.randn(a.nnz)
        if rng.rand() < 0.3:   # a band of empty rows
            lo = rng.randint(0, nrow - 40)
            a = scipy.sparse.vstack([a[:lo], scipy.sparse.csr_matrix((30, ncol)), a[lo + 30:]]).tocsr()
        nchunks = int(rng.randint(2, 7))
        cuts = sorted(set([0, nrow] + [int(c) & ~1 for c in rng.randint(2000, nrow - 2000, size=nchunks - 1)]))
        if tall and rng.rand() < 0.5:
            os.environ["SLP_TALL_R"] = str(int(rng.choice([1024, 1500, 4096, 9984])))
        else:
            os.environ.pop("SLP_TALL_R", None)
        if verbose:
            print("case", case, "tall" if tall else "strips", nrow, ncol, a.nnz, cuts, "dict" if dictionary else "fp64", flush=True)
        g = ChunkedDeviceMatrix(ncol, expect_chunks=len(cuts) - 1)
        try:
            for r0, r1 in zip(cuts, cuts[1:]):
                ch = DeviceMatrix.from_csr(a[r0:r1])
                if not dictionary:
                    ch.set_format(1)
                g.append(ch)
        except SlpError as e:
            if "does not qualify" not in str(e) and "empty chunk" not in str(e):
                raise
            skipped += 1
            g.close()
            continue
        x, y = rng.randn(ncol), rng.randn(nrow)
        oa = oracle.as_csr(a)
        ok = np.array_equal(g.matvec(x), oracle.matvec(oa, x)) and np.array_equal(g.rmatvec(y), oracle.rmatvec(oa, y))
        kinds = (lib.slp_matrix_spmv_kernel(g._h, 0), lib.slp_matrix_spmv_kernel(g._h, 1))
        for kk in kinds:
            seen[kk] = seen.get(kk, 0) + 1
        if not ok:
            raise AssertionError(f"chunked SpMV mismatch: case {case} shape {nrow} x {ncol} nnz {a.nnz} cuts {cuts} kernels {kinds}")
        if case % 3 == 0 and 5 not in kinds:   # (the fp64 WIDE strips cannot form the |v|^p sums of the preconditioners)
            c, lb, ub = rng.randn(ncol), -rng.rand(ncol), rng.rand(ncol)
            b = oracle.matvec(oa, rng.uniform(-0.5, 0.5, size=ncol)) + rng.rand(nrow)
            want, _ = oracle.chambolle_pock_ppd(c, None, None, oa, None, b, lb, ub, nb_max_iter=4, nb_iter_plot=10 ** 9)
            s = DeviceCP(g, b, c, lb, ub)
            s.iterate(4)
            got = s.x()
            s.close()
            if not np.array_equal(got, want):
                raise AssertionError(f"chunked Chambolle-Pock mismatch: case {case} shape {nrow} x {ncol} cuts {cuts} kernels {kinds}")
            solved += 1
        g.close()
    os.environ.pop("SLP_TALL_R", None)
    os.environ.pop("SLP_STRIP_MIN_NNZ", None)
    return seen, skipped, solved


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=120)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--verbose", action="store_true")
    args = p.parse_args()
    seen, skipped, solved = run(args.cases, args.seed, args.verbose)
    print("ok:", args.cases, "cases,", skipped, "skipped (a chunk without strip copies),", solved, "with Chambolle-Pock iterates;",
          "kernel codes (1 fp64 strips, 2 pairs, 3 quads, 4/5 wide, 6/7 tall cells):", dict(sorted(seen.items())))


if __name__ == "__main__":
    main()
