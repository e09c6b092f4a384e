"""Block-splitting ADMM (lp_admm_block_decomposition) on the Potts LP: GPU (matrix-free per-block projections)
next to the CPU oracle (sparse LU per block, like the reference).  python tools/bench_blocks.py [--potts 256] [--iters 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--potts", type=int, default=256)
    p.add_argument("--iters", type=int, default=200)
    p.add_argument("--cpu-iters", type=int, default=20)
    args = p.parse_args()
    from oracle import oracle
    from pysparselp_amd import ADMMBlocks
    from pysparselp_amd.problems import potts_lp

    lp = potts_lp(args.potts)[0]
    a_ineq = lp.a_inequalities
    call = (lp.costsvector, None, None, a_ineq, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    out = {"potts": args.potts, "n": int(lp.nb_variables), "rows": int(a_ineq.shape[0]), "blocks": len(a_ineq.blocks)}
    stats = {}
    orig = ADMMBlocks.BlocksState.report

    def spy(self):
        r = orig(self)
        stats["cg_steps"] = float(r[1])
        return r

    ADMMBlocks.BlocksState.report = spy
    ADMMBlocks.lp_admm_block_decomposition(*call, nb_iter=3, nb_iter_plot=1000)  # warm-up (library init, kernels)
    t0 = time.perf_counter()
    xg = ADMMBlocks.lp_admm_block_decomposition(*call, nb_iter=args.iters, nb_iter_plot=args.iters)
    dt = time.perf_counter() - t0
    out["gpu_it_per_s_incl_setup"] = (args.iters + 1) / dt
    out["gpu_cg_steps_per_iteration"] = stats.get("cg_steps", 0.0) / (args.iters + 1)
    t0 = time.perf_counter()
    xo = oracle.lp_admm_block_decomposition(*call, nb_iter=args.cpu_iters, nb_iter_plot=10 ** 9)
    dt = time.perf_counter() - t0
    out["cpu_oracle_it_per_s_incl_factorisation"] = (args.cpu_iters + 1) / dt
    xg2 = ADMMBlocks.lp_admm_block_decomposition(*call, nb_iter=args.cpu_iters, nb_iter_plot=10 ** 9)
    out["max_rel_diff_after_cpu_iters"] = float(np.max(np.abs(xg2 - xo) / (1 + np.abs(xo))))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
