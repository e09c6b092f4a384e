"""Long-horizon objective parity on current code: the GPU solvers against the CPU oracle (64 threads
over independent rows / columns: results do not depend on the thread count) over thousands of iterations, at a size where
the CPU can finish -- north_star: "converging to the same objective as the CPU reference within 1e-6 relative".

Two LPs of 2e7 stored entries from the benchmark generator (randomLP.py:14-75 restated, seed 1):
  strips  1e5 variables x 2e5 rows at 1e-3  (100 entries per row: LDS strips -- value dictionary, fp64 entries -- and CSR)
  tall    1e6 variables x 2e5 rows at 1e-4  (0.41 entries per row and 4096 columns, as config 4: tall cells -- both item forms)
Chambolle-Pock (ChambollePockPPD.py:195-343) for 2000 iterations, the matrix-free ADMM (ADMM.py:143-268, use_cg) for 1000;
at iterations 100 / 500 / 1000 / 2000: objective c.x relative to the oracle's, max |dx|, and the worst row violation
max(A x - b) of both.  Bars: objective 1e-6 relative (CP: the iterates are bit-identical, so 0), violation equal to 1e-6.

``--eq-frac 0.1`` (round 6): the 10 %-equality variant of the same LPs (Chambolle-Pock then forms (c + y_eq a_eq) + y_ineq a_ineq
from two products, csrc/slp_cp.hip cp_split_setup) -- the same bars.

Writes gpurun_out/convergence_parity.json (copy to profiles/).  Usage: python tools/convergence_parity.py [--quick] [--eq-frac 0.1]"""
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
from pysparselp_amd.admm_cg import DeviceADMM  # noqa: E402
from pysparselp_amd.problems import random_lp_on_device  # noqa: E402
from pysparselp_amd.scale import DeviceCP  # noqa: E402

SHAPES = {"strips": (100_000, 200_000, 1e-3), "tall": (1_000_000, 200_000, 1e-4)}
FORMATS = {"strips": (("dictionary strips", 0, (2, 3)), ("fp64 strips", 1, (1,)), ("csr", 2, (0,))),
           "tall": (("tall cells, dictionary items", 0, (6,)), ("tall cells, fp64 entries", 1, (7,)))}


def run(shape, checkpoints_cp, checkpoints_admm, threads=64, seed=1, eq_frac=0.0):
    """One LP: the oracle once per solver (iterates kept at the checkpoints), every format of the shape on the GPU.
    ``eq_frac`` > 0: SURVEY 8(d)'s equality variant (randomLP.py:62-68): the first eq_frac * m rows are equalities."""
    n, m, p = SHAPES[shape]
    m_eq = (int(round(eq_frac * m)) & ~1) if eq_frac > 0 else 0
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    oracle.set_threads(threads)
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=seed, m_eq=m_eq)
    host = a.download()
    a.close()
    a_eq, b_eq, a_in, b_in = (host[:m_eq], b[:m_eq], host[m_eq:], b[m_eq:]) if m_eq else (None, None, host, b)
    rec = {"n": n, "m": m, "equality_rows": m_eq, "density": p, "stored_entries": int(host.nnz),
           "objective_at_the_generators_feasible_point": float(c.dot(xf))}
    want = {"cp": {}, "admm": {}}

    def keep(store, points):
        def hook(i, x, *_):
            if i + 1 in points:
                store[i + 1] = np.array(x[:n], copy=True)
        return hook

    t0 = time.perf_counter()
    oracle.chambolle_pock_ppd(c, a_eq, b_eq, a_in, None, b_in, lb, ub, nb_max_iter=max(checkpoints_cp), nb_iter_plot=10 ** 9,
                              iterate_hook=keep(want["cp"], set(checkpoints_cp)))
    rec["oracle_cp_seconds"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.lp_admm_cg(c, a_eq, b_eq, a_in, None, b_in, lb, ub, nb_iter=max(checkpoints_admm) - 1, nb_iter_plot=10 ** 9,
                      iterate_hook=keep(want["admm"], set(checkpoints_admm)))
    rec["oracle_admm_seconds"] = time.perf_counter() - t0

    def violation(x):   # inequality rows: a_i x - b_i ; equality rows: |a_i x - b_i|
        r = host @ x - b
        r[:m_eq] = np.abs(r[:m_eq])
        return float(np.max(r))

    def compare(x, ref):
        og, oc = float(c.dot(x)), float(c.dot(ref))
        return {"objective_gpu": og, "objective_cpu": oc, "objective_relative_gap": abs(og - oc) / abs(oc),
                "max_abs_dx": float(np.max(np.abs(x - ref))), "bit_identical": bool(np.array_equal(x, ref)),
                "max_row_violation_gpu": violation(x), "max_row_violation_cpu": violation(ref)}

    rec["formats"] = {}
    for name, policy, kernels in FORMATS[shape]:
        out = {"chambolle_pock_ppd": {}, "admm": {}}
        a = random_lp_on_device(n, m, p, seed=seed)[0]
        a.set_format(policy)
        cp = DeviceCP(a, b, c, lb, ub, m_eq=m_eq)
        out["chambolle_pock_split_form"] = cp.split_form()
        assert a.spmv_kernel(False) in kernels, (name, a.spmv_kernel(False))
        done = 0
        for k in sorted(checkpoints_cp):
            cp.iterate(k - done)
            done = k
            out["chambolle_pock_ppd"][str(k)] = compare(cp.x(), want["cp"][k])
        cp.close()
        admm = DeviceADMM(a, b, c, lb, ub, m_eq=m_eq)     # (without a dictionary the rows are scaled in place: the matrix is not reused)
        out["admm_reuse_level"] = admm.reuse
        done = 0
        for k in sorted(checkpoints_admm):
            admm.iterate(k - done)
            done = k
            out["admm"][str(k)] = compare(admm.x(n), want["admm"][k])
        admm.close()
        a.close()
        rec["formats"][name] = out
        print(json.dumps({shape: {name: out}}), flush=True)
    os.environ.pop("SLP_STRIP_MIN_NNZ", None)
    return rec


def verdict(rec):
    """Worst gaps over everything: (objective relative, violation difference)."""
    obj = vio = 0.0
    for shape in rec["shapes"].values():
        for fmt in shape["formats"].values():
            for solver in ("chambolle_pock_ppd", "admm"):
                for r in fmt[solver].values():
                    obj = max(obj, r["objective_relative_gap"])
                    vio = max(vio, abs(r["max_row_violation_gpu"] - r["max_row_violation_cpu"]))
    return obj, vio


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--quick", action="store_true", help="200 iterations (the copy that runs inside pytest -m gpu)")
    p.add_argument("--threads", type=int, default=64)
    p.add_argument("--eq-frac", type=float, default=0.0, help="fraction of the rows turned into equalities (randomLP.py:62-68)")
    p.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "convergence_parity.json"))
    args = p.parse_args()
    _lib.lib(0)
    cps = (50, 200) if args.quick else (100, 500, 1000, 2000)
    ads = (50, 200) if args.quick else (100, 500, 1000)
    rec = {"checkpoints_cp": cps, "checkpoints_admm": ads, "oracle_threads": args.threads, "eq_frac": args.eq_frac, "shapes": {}}
    for shape in SHAPES:
        rec["shapes"][shape] = run(shape, cps, ads, threads=args.threads, eq_frac=args.eq_frac)
    obj, vio = verdict(rec)
    rec["worst_objective_relative_gap"], rec["worst_violation_difference"] = obj, vio
    rec["bars"] = {"objective_relative": 1e-6, "violation_difference": 1e-6}
    assert obj <= 1e-6 and vio <= 1e-6, (obj, vio)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "shapes"}), flush=True)


if __name__ == "__main__":
    main()
