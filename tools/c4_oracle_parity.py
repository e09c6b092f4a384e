"""WHOLE SOLVERS on BASELINE config 4 at full size (1e7 variables x 2e7 rows, density 1e-4: 2e10 stored entries), ONE GPU, against
the CPU oracle -- the parity gate of bench.py's default workload, as tests/test_gpu_c3_full.py is for config 3.

The device holds the LP as a ChunkedDeviceMatrix (8 row chunks whose CSR never coexists).  Every chunk's CSR is downloaded
into one host CSR before the chunk is converted and released (240 GB on the host; the GPU boxes have 2.9 TB), and the oracle
(oracle/oracle.py + oracle/slp_oracle.c: the restatement of ChambollePockPPD.py:195-343 and of ADMM.py:143-268 with the use_cg
flags, pinned bit for bit against the imported reference by tests/golden/make_golden.py) runs on it with its loops over
independent rows / columns on 64 threads (bit-identical to one thread: tests/test_oracle_golden.py).

  Chambolle-Pock, 3 iterations:     x must agree BIT FOR BIT (every sum on this path is the reference's chain);
  matrix-free ADMM, 2 iterations:   |dx| <= 1e-9 (1 + |x|), objective within 1e-6 relative (north_star's tolerance).

Takes ~10-15 minutes of box time and ~1 TB of HOST memory: a tool, not a test.  WARNING: the one attempt at full size on the
shared GPU pool (round 4) took the box down before it produced a record -- most likely the host memory (the pool's boxes
report 2.9 TB, what a job may actually use is less).  It therefore refuses to run at full size unless SLP_ALLOW_HUGE_HOST=1 is
set AND the cgroup's memory limit covers the need; `--scale 0.1` (2e9 entries, 30 GB of host arrays) is the supported dry run.
    python tools/c4_oracle_parity.py [--chunks 8] [--cp-iters 3] [--admm-iters 2] [--scale 0.1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import oracle  # noqa: E402  (the checker)
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.admm_cg import DeviceADMM  # noqa: E402
from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix  # noqa: E402
from pysparselp_amd.scale import DeviceCP  # noqa: E402


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--chunks", type=int, default=8)
    p.add_argument("--cp-iters", type=int, default=3)
    p.add_argument("--admm-iters", type=int, default=2)
    p.add_argument("--scale", type=float, default=0.1, help="shrink rows and columns by this factor (1.0 = config 4 itself: ~1 TB of host memory)")
    p.add_argument("--seed", type=int, default=0)
    args = p.parse_args()
    n, m, dens = int(10_000_000 * args.scale), int(20_000_000 * args.scale), 1e-4 / args.scale
    lib = _lib.lib(0)
    threads = min(64, os.cpu_count() or 1)
    rec = {"n": n, "m": m, "density": dens, "seed": args.seed, "chunks": args.chunks, "oracle_threads": threads,
           "host_mem_available_gb": mem_available_gb()}
    need = 5.5 * 12e-9 * n * m * dens  # CSR + CSC + scaled copies, GB
    limit = float("inf")
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            txt = open(path).read().strip()
            if txt != "max":
                limit = min(limit, int(txt) / 2 ** 30)
        except OSError:
            pass
    rec["host_cgroup_limit_gb"] = None if limit == float("inf") else limit
    if min(rec["host_mem_available_gb"], limit) < 1.5 * need:
        raise SystemExit(f"needs ~{need:.0f} GB of host memory with headroom; available {rec['host_mem_available_gb']:.0f}, cgroup limit {limit}")
    if need > 200 and os.environ.get("SLP_ALLOW_HUGE_HOST") != "1":
        raise SystemExit(f"~{need:.0f} GB of host memory: set SLP_ALLOW_HUGE_HOST=1 on a machine of your own (see the module docstring)")

    # ---- device: the chunked matrix; host: the same rows, chunk by chunk, into ONE CSR
    t0 = time.perf_counter()
    cap = int(n * m * dens * 1.002) + 1_000_000
    indptr = np.empty(m + 1, dtype=np.int64)
    indices = np.empty(cap, dtype=np.int32)
    data = np.empty(cap)
    indptr[0] = 0
    cuts = ChunkedDeviceMatrix.cuts(m, args.chunks)
    a = ChunkedDeviceMatrix(n, expect_chunks=len(cuts) - 1)
    b = np.empty(m)
    filled = 0
    for k, (r0, r1) in enumerate(zip(cuts, cuts[1:])):
        ch = DeviceMatrix.random(r1 - r0, n, dens, args.seed, r0)
        got = ch.random_lp_vectors(dens, args.seed, r0, columns=(k == 0))
        if k == 0:
            xf, c, lb, ub = got[:4]
        b[r0:r1] = got[4]
        nz = ch.nnz
        assert filled + nz <= cap
        ptr = np.empty(r1 - r0 + 1, dtype=np.int64)
        _lib.check(lib.slp_matrix_download(ch._h, 0, _lib.ptr(ptr), _lib.ptr(indices[filled:filled + nz]), _lib.ptr(data[filled:filled + nz])))
        indptr[r0 + 1:r1 + 1] = ptr[1:] + filled
        filled += nz
        a.append(ch)
    rec["stored_entries"] = filled
    rec["build_and_download_seconds"] = time.perf_counter() - t0
    host = oracle.Csr.__new__(oracle.Csr)   # views, no 240 GB copies
    host._csc = None
    host.indptr, host.indices, host.data, host.shape = indptr, indices[:filled], data[:filled], (m, n)
    assert a.nnz == filled
    oracle.set_threads(threads)
    try:
        # ---- Chambolle-Pock
        stamps = []
        t0 = time.perf_counter()
        x_cpu, _ = oracle.chambolle_pock_ppd(c, None, None, host, None, b, lb, ub, nb_max_iter=args.cp_iters, nb_iter_plot=10 ** 9,
                                             iterate_hook=lambda *_: stamps.append(time.perf_counter()))
        rec["chambolle_pock_ppd"] = {"iterations": args.cp_iters, "oracle_setup_seconds": stamps[0] - t0,
                                     "oracle_seconds_per_iteration": float(np.mean(np.diff(stamps))) if len(stamps) > 1 else None,
                                     "objective_oracle": float(c.dot(x_cpu))}
        s = DeviceCP(a, b, c, lb, ub)
        s.iterate(args.cp_iters)
        x_gpu = s.x()
        s.close()
        same = bool(np.array_equal(x_gpu, x_cpu))
        rec["chambolle_pock_ppd"].update({"objective_gpu": float(c.dot(x_gpu)), "x_bit_for_bit": same,
                                          "max_abs_difference": float(np.max(np.abs(x_gpu - x_cpu)))})
        print(json.dumps(rec), flush=True)
        # (shrunk dry runs land in the LDS-strip regime, where chunks with few row blocks split their strips over several
        # workgroups: partial sums in strip order, not the single chain -- 1e-12 there)
        assert same or (args.scale < 1.0 and rec["chambolle_pock_ppd"]["max_abs_difference"] <= 1e-12), rec["chambolle_pock_ppd"]
        host._csc = None
        # ---- matrix-free ADMM (reuse level 4 on the device; the oracle writes the iteration as the reference does)
        if args.admm_iters > 0:
            stamps = []
            t0 = time.perf_counter()
            x_cpu = oracle.lp_admm_cg(c, None, None, host, None, b, lb, ub, nb_iter=args.admm_iters - 1, nb_iter_plot=10 ** 9,
                                      iterate_hook=lambda *_: stamps.append(time.perf_counter()))
            rec["admm"] = {"iterations": args.admm_iters, "oracle_setup_seconds": stamps[0] - t0,
                           "oracle_seconds_per_iteration": float(np.mean(np.diff(stamps))) if len(stamps) > 1 else None,
                           "objective_oracle": float(c.dot(x_cpu))}
            s = DeviceADMM(a, b, c, lb, ub)
            s.iterate(args.admm_iters)
            x_gpu = s.x(n)
            s.close()
            err = float(np.max(np.abs(x_gpu - x_cpu) / (1 + np.abs(x_cpu))))
            rec["admm"].update({"objective_gpu": float(c.dot(x_gpu)), "max_scaled_error": err,
                                "objective_relative_difference": abs(float(c.dot(x_gpu)) - float(c.dot(x_cpu))) / abs(float(c.dot(x_cpu)))})
            assert err <= 1e-9 and rec["admm"]["objective_relative_difference"] <= 1e-6, rec["admm"]
    finally:
        oracle.set_threads(1)
    a.close()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "c4_full_oracle_parity.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
