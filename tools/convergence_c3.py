"""Convergence at BASELINE config 3 (no CPU run at this size): objective c.x and worst constraint / bound violation of the
iterate after 50 ... 800 iterations of each at-scale solver.  python tools/convergence_c3.py > profiles/rNN_c3_convergence.json
Config 4 on one GPU (the LP as a chunked matrix, ~4 minutes):
    python tools/convergence_c3.py --n 10000000 --m 20000000 --density 1e-4 --chunks 8 > profiles/rNN_c4_convergence.json"""
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
    p.add_argument("--chunks", type=int, default=0, help="> 1: the LP as a ChunkedDeviceMatrix of that many row chunks")
    args = p.parse_args()
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import make_solver

    marks = [int(v) for v in args.marks.split(",")]
    a, xf, c, lb, ub, b = random_lp_on_device(args.n, args.m, args.density, seed=0, chunks=max(1, args.chunks))
    out = {"n": args.n, "m": args.m, "density": args.density, "nnz": a.nnz, "chunks": max(1, args.chunks),
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
