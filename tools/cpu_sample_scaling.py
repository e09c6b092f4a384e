"""Is the cpu_baseline's row-sample extrapolation valid at n = 1e7?

bench.py times the oracle (1 thread, iterations only) on the first m/100 rows of config 4 (all 1e7 columns) and scales by the
row ratio.  This tool times the same thing on 1 %, 2 % and 5 % of the rows: if the cost per iteration is linear in the rows at
fixed n the extrapolated full-size rates agree.  Writes gpurun_out/cpu_sample_scaling_c4.json (copy to profiles/).
Usage: python tools/cpu_sample_scaling.py [--fractions 0.01 0.02 0.05]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import oracle  # noqa: E402
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=10_000_000)
    p.add_argument("--m", type=int, default=20_000_000)
    p.add_argument("--density", type=float, default=1e-4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--fractions", type=float, nargs="+", default=[0.01, 0.02, 0.05])
    p.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "cpu_sample_scaling_c4.json"))
    args = p.parse_args()
    _lib.lib(0)
    oracle.set_threads(1)
    rec = {"n": args.n, "m": args.m, "density": args.density, "threads": 1, "samples": []}
    for frac in args.fractions:
        rows = int(round(args.m * frac))
        a = DeviceMatrix.random(rows, args.n, args.density, args.seed)
        xf, c, lb, ub, b = a.random_lp_vectors(args.density, args.seed)
        s = a.download()
        a.close()
        row = {"fraction": frac, "rows": rows, "stored_entries": int(s.nnz)}
        for method in ("admm", "chambolle_pock_ppd"):
            stamps = []

            def hook(i, *_):
                stamps.append(time.perf_counter())

            if method == "admm":
                oracle.lp_admm_cg(c, None, None, s, None, b, lb, ub, nb_iter=3, nb_iter_plot=10 ** 9, iterate_hook=hook)
            else:
                oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=5, nb_iter_plot=10 ** 9, iterate_hook=hook)
            per_iter = (stamps[-1] - stamps[1]) / (len(stamps) - 2)
            row[method] = {"seconds_per_iteration_on_sample": per_iter, "it_per_s_on_sample": 1.0 / per_iter,
                           "extrapolated_full_size_it_per_s": (1.0 / per_iter) * rows / args.m,
                           "seconds_per_1e9_entries": per_iter / (s.nnz / 1e9)}
        rec["samples"].append(row)
        print(json.dumps(row), flush=True)
    for method in ("admm", "chambolle_pock_ppd"):
        v = [r[method]["extrapolated_full_size_it_per_s"] for r in rec["samples"]]
        rec[method + "_extrapolations_spread"] = (max(v) - min(v)) / min(v)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "samples"}), flush=True)


if __name__ == "__main__":
    main()
