"""Setup cost of lp_admm on the Potts 256x256 LP (BASELINE config 2's problem, ADMM with the explicit M): device chain
(default) against the numpy + scipy chain (SLP_HOST_SETUP=1 SLP_HOST_SPGEMM=1), with SLP_TRACE phases on stderr.
    SLP_TRACE=1 python tools/setup_small.py"""
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
lp, gt, pix, unary = potts_lp(256)
args = (lp.costsvector, None, None, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
out = {}
for name, env in (("device_chain", {}), ("host_chain", {"SLP_HOST_SETUP": "1", "SLP_HOST_SPGEMM": "1"})):
    os.environ.pop("SLP_HOST_SETUP", None)
    os.environ.pop("SLP_HOST_SPGEMM", None)
    os.environ.update(env)
    for rep in range(2):
        t0 = time.perf_counter()
        x = lp_admm(*args, nb_iter=0, nb_iter_plot=10 ** 9)
        out[name + "_seconds_for_setup_plus_1_iteration"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    x = lp_admm(*args, nb_iter=200, nb_iter_plot=10 ** 9)
    out[name + "_seconds_for_201_iterations"] = time.perf_counter() - t0
    out[name + "_objective"] = float(np.dot(lp.costsvector, x))
print(json.dumps(out))
