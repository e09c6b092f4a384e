"""EVERY row of the resident copies of BASELINE config 4 against the CPU oracle, once.

The metric's LP (1e7 variables x 2e7 rows at 1e-4: 2e10 stored entries) lives on one GPU as a ChunkedDeviceMatrix whose CSR
never exists as a whole; tests/test_gpu_c4_full.py checks its products on row slices.  Here, with the chunked matrix resident,
every chunk is regenerated as an ordinary DeviceMatrix (the generator is keyed by the global row), downloaded (30 GB) and
multiplied by the oracle (oracle/slp_oracle.c, 64 threads over independent rows / columns -- no sum is split or reordered):

  ``A x``      dense x: the chunk's rows of the device product, bit for bit against ``oracle.matvec`` (csr_matvec order);
  ``A^T y``    DENSE y: the oracle continues the column sums chunk by chunk (``oracle.rmatvec_acc``: the chain of additions of
               scipy's csc_matvec over the stacked rows); after the last chunk the whole vector bit for bit;
  the same two for ``|A|^p``, p = 1, 2 (the sums behind the Chambolle-Pock preconditioners, ChambollePockPPD.py:134,144,161,172).

Reference products: ChambollePockPPD.py:206,216,235,240.  Writes gpurun_out/c4_all_rows.json (copy to profiles/).
Usage: python tools/c4_all_rows.py [--chunks 8] [--threads 64] [--n ... --m ... --density ...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import oracle  # noqa: E402  (the checker; never on the product path)
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix  # noqa: E402
from pysparselp_amd.problems import random_lp_on_device  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=10_000_000)
    p.add_argument("--m", type=int, default=20_000_000)
    p.add_argument("--density", type=float, default=1e-4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--chunks", type=int, default=8)
    p.add_argument("--threads", type=int, default=64)
    p.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "c4_all_rows.json"))
    args = p.parse_args()
    lib = _lib.lib(0)
    n, m = args.n, args.m
    oracle.set_threads(args.threads)
    rec = {"n": n, "m": m, "density": args.density, "seed": args.seed, "chunks": args.chunks, "oracle_threads": args.threads,
           "what": "every row of the chunked copies against the oracle: A x and A^T y for dense vectors, and the same with |A|^p, "
                   "p = 1, 2 -- all bit for bit"}
    t_all = time.perf_counter()
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, args.density, seed=args.seed, chunks=args.chunks)
    _lib.check(lib.slp_synchronize())
    rec["build_seconds"] = time.perf_counter() - t_all
    rec["stored_entries"] = a.nnz
    rec["kernels"] = [a.spmv_kernel(False), a.spmv_kernel(True)]
    rng = np.random.RandomState(17)
    x, y = rng.randn(n), rng.randn(m)
    powers = (None, 1.0, 2.0)   # None: the matrix itself
    dev_ax = {pw: (a.matvec(x) if pw is None else a.abs_pow_matvec(x, pw)) for pw in powers}
    dev_aty = {pw: (a.rmatvec(y) if pw is None else a.abs_pow_matvec(y, pw, transposed=True)) for pw in powers}
    cpu_aty = {pw: np.zeros(n) for pw in powers}
    cuts = ChunkedDeviceMatrix.cuts(m, args.chunks)
    assert len(cuts) - 1 == a.chunks
    rec["per_chunk"] = []
    rows_checked = entries = 0
    for k, (r0, r1) in enumerate(zip(cuts, cuts[1:])):
        t0 = time.perf_counter()
        chunk = DeviceMatrix.random(r1 - r0, n, args.density, args.seed, r0)   # the same rows again (counter-based generator)
        host = chunk.download()
        chunk.close()
        t_dl = time.perf_counter() - t0
        t0 = time.perf_counter()
        for pw in powers:
            csr = oracle.Csr(host.indptr, host.indices, host.data if pw is None else np.abs(host.data) ** pw, host.shape)
            want = oracle.matvec(csr, x)
            got = dev_ax[pw][r0:r1]
            assert np.array_equal(got, want), ("A x", pw, k, int(np.argmax(got != want)))
            oracle.rmatvec_acc(csr, y[r0:r1], cpu_aty[pw])
            del csr
        rows_checked += r1 - r0
        entries += host.nnz
        rec["per_chunk"].append({"rows": [int(r0), int(r1)], "stored_entries": int(host.nnz), "generate_and_download_seconds": t_dl,
                                 "oracle_seconds": time.perf_counter() - t0})
        print(json.dumps(rec["per_chunk"][-1]), flush=True)
        del host
    for pw in powers:
        bad = np.nonzero(dev_aty[pw] != cpu_aty[pw])[0]
        assert bad.size == 0, ("A^T y", pw, bad[:5], dev_aty[pw][bad[:5]], cpu_aty[pw][bad[:5]])
    assert entries == a.nnz and rows_checked == m
    rec.update({"rows_checked": int(rows_checked), "columns_checked": int(n), "stored_entries_walked_by_the_oracle": int(entries),
                "products_checked": ["A x", "A^T y", "|A| x", "|A|^T y", "|A|^2 x", "(|A|^2)^T y"], "mismatches": 0,
                "seconds": time.perf_counter() - t_all})
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "per_chunk"}), flush=True)
    a.close()


if __name__ == "__main__":
    main()
