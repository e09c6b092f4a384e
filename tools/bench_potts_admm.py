"""Potts n x n LP, exact Gauss-Seidel ADMM (the shipped method="admm"): iterations per second from the device-side timer
(slp_admm_bench: HIP events around k iterations, no setup, no host work), for a list of SLP_GS_BANDS settings.
python tools/bench_potts_admm.py [--potts 256] [--iters 2000] [--bands auto,0,4,8,16]"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--potts", type=int, default=256)
    p.add_argument("--iters", type=int, default=2000)
    p.add_argument("--bands", default="auto,0,4,8,16")
    p.add_argument("--check", type=int, default=0, help="compare this many iterations with the oracle (bit for bit)")
    args = p.parse_args()
    from pysparselp_amd.ADMM import ADMMState
    from pysparselp_amd.problems import potts_lp

    lp = potts_lp(args.potts)[0]
    a = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    ref = None
    if args.check:
        from oracle import oracle

        ref = oracle.lp_admm(*a, nb_iter=args.check, nb_iter_plot=10 ** 9)
    from pysparselp_amd.ADMM import lp_admm
    out = {"potts": args.potts, "variant": os.environ.get("SLP_LIB_VARIANT", ""), "runs": []}
    for b in args.bands.split(","):
        if b == "auto":
            os.environ.pop("SLP_GS_BANDS", None)
        else:
            os.environ["SLP_GS_BANDS"] = b
        exact = None
        if ref is not None:
            exact = bool(np.array_equal(lp_admm(*a, nb_iter=args.check, nb_iter_plot=10 ** 9), ref))
        st = ADMMState.from_lp(*a, None, 2, 3)
        st.bench(50)
        ms = [st.bench(args.iters) for _ in range(3)]
        out["runs"].append({"SLP_GS_BANDS": b, "bands": st.num_bands(), "levels": st.num_levels(), "it_per_s": round(1000.0 / min(ms), 1),
                            "ms_per_iteration": round(min(ms), 5), "bit_exact_vs_oracle": exact})
        st.close()
    os.environ.pop("SLP_GS_BANDS", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
