"""Convergence at BASELINE config 3 (no CPU run at this size): objective c.x and worst constraint / bound violation of the
iterate after 50 ... 800 iterations of each at-scale solver.  python tools/convergence_c3.py > profiles/rNN_c3_convergence.json"""
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
    p.add_argument("--n", type=int, default=1_000_000)
    p.add_argument("--m", type=int, default=2_000_000)
    p.add_argument("--density", type=float, default=1e-3)
    p.add_argument("--marks", default="50,100,200,400,800")
    args = p.parse_args()
    from pysparselp_amd.device import DeviceMatrix
    from pysparselp_amd.scale import make_solver

    marks = [int(v) for v in args.marks.split(",")]
    a = DeviceMatrix.random(args.m, args.n, args.density, 0)
    xf, c, lb, ub, b = a.random_lp_vectors(args.density, 0)
    out = {"n": args.n, "m": args.m, "density": args.density, "nnz": a.nnz,
           "feasible_point": {"objective": float(c.dot(xf)), "max_row_violation": float(np.max(np.maximum(a.matvec(xf) - b, 0)))}}
    for method in ("chambolle_pock_ppd", "admm"):
        s = make_solver(method, a, b, c, lb, ub)
        rows, done, t0 = [], 0, time.perf_counter()
        for mark in marks:
            s.iterate(mark - done)
            done = mark
            x = s.x() if method != "admm" else s.x(args.n)
            rows.append({"iteration": mark, "objective": float(c.dot(x)),
                         "max_row_violation": float(np.max(np.maximum(a.matvec(x) - b, 0))),
                         "max_bound_violation": float(max(np.max(np.maximum(lb - x, 0)), np.max(np.maximum(x - ub, 0)))),
                         "seconds_since_start": time.perf_counter() - t0})
        s.close()
        out[method] = rows
    a.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
