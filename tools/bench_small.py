"""Cache-resident / launch-latency-bound configurations (BASELINE configs 1-2): iterations per second of the
GPU solvers next to the CPU oracle, on SC105 and on the Potts LP (default 256 x 256).  Not the headline bench.

    python tools/bench_small.py [--potts 256] [--cp-iters 5000] [--admm-iters 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--potts", type=int, default=256)
    p.add_argument("--cp-iters", type=int, default=5000)
    p.add_argument("--admm-iters", type=int, default=300)
    p.add_argument("--cpu-iters", type=int, default=40)
    args = p.parse_args()
    from oracle import oracle
    from pysparselp_amd import _lib
    from pysparselp_amd.ADMM import lp_admm
    from pysparselp_amd.ChambollePockPPD import chambolle_pock_ppd
    from pysparselp_amd.problems import potts_lp

    lib = _lib.lib()
    lp, gt, gt_idx, _ = potts_lp(args.potts)
    a = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    out = {"potts": args.potts, "n": lp.nb_variables, "rows": lp.a_inequalities.shape[0], "nnz": int(lp.a_inequalities.nnz)}

    def timed(fn):
        t0 = time.perf_counter()
        r = fn()
        lib.slp_synchronize()
        return r, time.perf_counter() - t0

    big = 10 ** 9
    # CP (a_eq must be an empty CSR for the reference signature: pass a 0-row matrix)
    import scipy.sparse

    a_cp = (a[0], scipy.sparse.csr_matrix((0, lp.nb_variables)), np.zeros(0)) + a[3:]
    chambolle_pock_ppd(*a_cp, nb_max_iter=50, nb_iter_plot=big)  # warm-up (library init, transposes)
    (x, _), dt = timed(lambda: chambolle_pock_ppd(*a_cp, nb_max_iter=args.cp_iters, nb_iter_plot=big))
    out["cp_gpu_it_per_s"] = args.cp_iters / dt
    (xo, _), dto = timed(lambda: oracle.chambolle_pock_ppd(*a_cp, nb_max_iter=args.cpu_iters, nb_iter_plot=big))
    out["cp_cpu_oracle_it_per_s"] = args.cpu_iters / dto
    xg, _ = chambolle_pock_ppd(*a_cp, nb_max_iter=args.cpu_iters, nb_iter_plot=big)
    out["cp_bit_exact_vs_oracle"] = bool(np.array_equal(xg, xo))
    out["cp_dist_to_ground_truth"] = float(np.mean(np.abs(gt - x[gt_idx])))
    # ADMM (Gauss-Seidel x-step); setup (host SpGEMM for M) excluded by timing a second, longer run
    t0 = time.perf_counter()
    lp_admm(*a, nb_iter=1, nb_iter_plot=big)
    t_setup = time.perf_counter() - t0
    x, dt = timed(lambda: lp_admm(*a, nb_iter=args.admm_iters, nb_iter_plot=big))
    out["admm_setup_s"] = t_setup
    out["admm_gpu_it_per_s_incl_setup"] = (args.admm_iters + 1) / dt
    out["admm_gpu_it_per_s"] = args.admm_iters / max(dt - t_setup, 1e-9)
    xo, dto = timed(lambda: oracle.lp_admm(*a, nb_iter=args.cpu_iters, nb_iter_plot=big))
    t0 = time.perf_counter()
    oracle.lp_admm(*a, nb_iter=0, nb_iter_plot=big)
    t_setup_cpu = time.perf_counter() - t0
    out["admm_cpu_oracle_it_per_s"] = args.cpu_iters / max(dto - t_setup_cpu, 1e-9)
    xg = lp_admm(*a, nb_iter=args.cpu_iters, nb_iter_plot=big)
    out["admm_bit_exact_vs_oracle"] = bool(np.array_equal(xg, xo))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
