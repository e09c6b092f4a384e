"""One-off validation of bench.py's cpu_baseline extrapolation: the oracle's Chambolle-Pock (CPU restatement of
ChambollePockPPD.py:195-343) and matrix-free ADMM (ADMM.py:182-201 + conjugateGradientLinearSolver.py:30-52), one thread
like the reference, timed at FULL BASELINE config 3 size (1e6 x 2e6, ~2e9 stored entries), next to the same oracle on the
bounded sample bench.py uses (the first m/10 rows, all n columns), iterations only.  The ADMM SETUP (row scalings, standard
form: not part of the timed iterations) runs with ORACLE_SETUP_THREADS threads to keep the tool's wall time down; every
timed iteration runs on ONE thread in the reference's loop order.  Writes the record kept as profiles/r03_cpu_full_c3.json.

    python tools/cpu_full_c3.py > gpurun_out/cpu_full_c3.json      (GPU box: the LP is generated on the device)
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402

N, M, P, SEED = 1_000_000, 2_000_000, 1e-3, 0


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


def timed_cp(rows, iters):
    a = DeviceMatrix.random(rows, N, P, SEED)
    xf, c, lb, ub, b = a.random_lp_vectors(P, SEED)
    t0 = time.perf_counter()
    s = oracle.as_csr(a.download())
    t_down = time.perf_counter() - t0
    a.close()
    stamps = []
    t0 = time.perf_counter()
    x, _ = oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=iters, nb_iter_plot=10 ** 9,
                                     iterate_hook=lambda *_: stamps.append(time.perf_counter()))
    per = np.diff(stamps)
    return {"rows": rows, "stored_entries": s.nnz, "iterations_timed": int(per.size), "seconds_per_iteration": [float(v) for v in per],
            "it_per_s": float(1.0 / per.mean()), "setup_seconds": float(stamps[0] - t0 - per.mean()),
            "download_seconds": float(t_down), "objective": float(c.dot(x))}


def timed_admm(rows, iters):
    a = DeviceMatrix.random(rows, N, P, SEED)
    xf, c, lb, ub, b = a.random_lp_vectors(P, SEED)
    s = oracle.as_csr(a.download())
    a.close()
    stamps = []

    def hook(*_):
        stamps.append(time.perf_counter())
        oracle.set_threads(1)  # everything after the first x-step is timed on one thread

    oracle.set_threads(int(os.environ.get("ORACLE_SETUP_THREADS", "32")))
    t0 = time.perf_counter()
    try:
        x = oracle.lp_admm_cg(c, None, None, s, None, b, lb, ub, nb_iter=iters - 1, nb_iter_plot=10 ** 9, iterate_hook=hook)
    finally:
        oracle.set_threads(1)
    per = np.diff(stamps)
    return {"rows": rows, "stored_entries": s.nnz, "iterations_timed": int(per.size), "seconds_per_iteration": [float(v) for v in per],
            "it_per_s": float(1.0 / per.mean()), "setup_and_first_iteration_seconds_multithreaded": float(stamps[0] - t0),
            "objective": float(c.dot(x))}


def main():
    avail = mem_available_gb()
    out = {"host_cores_present": os.cpu_count(), "threads_used": 1, "mem_available_gb": avail}
    sample = timed_cp(M // 10, 8)
    out["sample"] = sample
    sample_admm = timed_admm(M // 10, 5)
    out["sample_admm"] = sample_admm
    if avail < 250:
        out["full"] = None
        out["note"] = "full-size run skipped: less than 250 GB of host memory available"
    else:
        full = timed_cp(M, 6)
        out["full"] = full
        predicted = sample["it_per_s"] * sample["rows"] / M
        out["chambolle_pock_ppd"] = {"full_size_it_per_s": full["it_per_s"], "predicted_from_sample_it_per_s": predicted,
                                     "ratio_measured_over_predicted": full["it_per_s"] / predicted}
        full_admm = timed_admm(M, 4)
        out["full_admm"] = full_admm
        predicted = sample_admm["it_per_s"] * sample_admm["rows"] / M
        out["admm"] = {"full_size_it_per_s": full_admm["it_per_s"], "predicted_from_sample_it_per_s": predicted,
                       "ratio_measured_over_predicted": full_admm["it_per_s"] / predicted}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
