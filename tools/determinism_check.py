import os, sys
sys.path.insert(0, os.getcwd())
os.environ["SLP_STRIP_MIN_NNZ"] = "1"
import numpy as np
from pysparselp_amd.admm_cg import DeviceADMM
from pysparselp_amd.problems import random_lp_on_device
from pysparselp_amd.scale import DeviceCP
for (n, m, p) in [(400000, 300000, 1e-4), (30000, 40000, 1e-3)]:
    for reuse in (0, 2):
        xs = []
        for rep in range(3):
            a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=2)
            s = DeviceADMM(a, b, c, lb, ub, reuse=reuse)
            s.iterate(20)
            xs.append(s.x(n).copy())
            s.close(); a.close()
        print(n, "reuse", reuse, "identical runs:", np.array_equal(xs[0], xs[1]), np.array_equal(xs[0], xs[2]), float(np.max(np.abs(xs[0]-xs[1]))))
