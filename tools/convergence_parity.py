"""GPU vs CPU oracle over longer horizons on the benchmark distribution (mid size): objective and iterate
agreement after 50 / 150 / 400 iterations of CP and of the matrix-free ADMM (reuse levels 0, 2 and 4).

    python tools/convergence_parity.py
"""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle
from pysparselp_amd.problems import random_lp_on_device
from pysparselp_amd.admm_cg import DeviceADMM
from pysparselp_amd.scale import DeviceCP
n, m, p = 20000, 40000, 0.005
res = {}
for iters in (50, 150, 400):
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=1)
    s = a.download()
    cp = DeviceCP(a, b, c, lb, ub); cp.iterate(iters); xg = cp.x(); cp.close()
    xo, _ = oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=iters, nb_iter_plot=10**9)
    res[f"cp_{iters}"] = dict(obj_gpu=float(c.dot(xg)), obj_cpu=float(c.dot(xo)), rel=float(abs(c.dot(xg)-c.dot(xo))/abs(c.dot(xo))), maxdx=float(np.max(np.abs(xg-xo))),
                             viol=float(np.max(s @ xg - b)))
    for level in (0, 2, 4):
        a2 = random_lp_on_device(n, m, p, seed=1)[0]
        ad = DeviceADMM(a2, b, c, lb, ub, reuse=level); ad.iterate(iters); xg = ad.x(n); rep = ad.report(); ad.close(); a2.close()
        xo = oracle.lp_admm_cg(c, None, None, s, None, b, lb, ub, nb_iter=iters-1, nb_iter_plot=10**9)
        res[f"admm_l{level}_{iters}"] = dict(obj_gpu=float(c.dot(xg)), obj_cpu=float(c.dot(xo)), rel=float(abs(c.dot(xg)-c.dot(xo))/abs(c.dot(xo))), maxdx=float(np.max(np.abs(xg-xo))),
                                           viol=float(np.max(s @ xg - b)), energy=float(rep[0]))
    a.close()
    print(json.dumps({k: v for k, v in res.items() if k.endswith(str(iters))}), flush=True)
