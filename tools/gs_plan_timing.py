"""Set-up of lp_admm on the Potts 256x256 LP with the sweep's plan built on the device (default) against the host plan
(SLP_GS_PLAN=host: one download of M, ~600 lines of host code, ~50 MB of uploads): warm seconds of lp_admm(nb_iter=0), with the
SLP_TRACE phases on stderr; and the 201-iteration solve, whose x must not change by a bit.
    SLP_TRACE=1 python tools/gs_plan_timing.py"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.ADMM import lp_admm  # noqa: E402
from pysparselp_amd.problems import potts_lp  # noqa: E402

_lib.lib(0)
out = {}
xs = {}
for size in (256, 512):
    lp, gt, pix, unary = potts_lp(size)
    args = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
    for mode in ("device", "host", "check"):
        os.environ["SLP_GS_PLAN"] = mode
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            lp_admm(*args, nb_iter=0, nb_iter_plot=10 ** 9)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        x = lp_admm(*args, nb_iter=200, nb_iter_plot=10 ** 9)
        out[f"potts{size}_{mode}"] = {"lp_admm_nb_iter_0_seconds": [round(t, 4) for t in ts], "seconds_for_201_iterations": round(time.perf_counter() - t0, 4),
                                      "objective": float(np.dot(lp.costsvector, x))}
        xs[(size, mode)] = x
    assert np.array_equal(xs[(size, "device")], xs[(size, "host")]) and np.array_equal(xs[(size, "check")], xs[(size, "host")])
    out[f"potts{size}_x_bit_identical"] = True
print(json.dumps(out))
