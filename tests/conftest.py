import os
import sys

import numpy as np
import pytest
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    """The oracle's C part is built on demand (gcc, < 1 s)."""
    import subprocess

    so = os.path.join(REPO, "oracle", "liboracle.so")
    src = os.path.join(REPO, "oracle", "slp_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def csr_of(d, tag):
    shape = tuple(int(v) for v in d[f"{tag}_shape"])
    m = scipy.sparse.csr_matrix((d[f"{tag}_data"], d[f"{tag}_indices"], d[f"{tag}_indptr"]), shape=shape)
    m.__dict__["blocks"] = []
    return m


def solver_args(d):
    """(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub) as SparseLP.solve hands them to lp_admm."""
    ae, ai = csr_of(d, "Ae"), csr_of(d, "Ai")
    bl = None if bool(d["bl_none"]) else d["bl"]
    a_eq, beq = (ae, d["be"]) if ae.shape[0] > 0 else (None, None)
    a_ineq = ai if ai.shape[0] > 0 else None
    return d["c"], a_eq, beq, a_ineq, bl, d["bu"], d["lb"], d["ub"]


def lp_from_golden(d, cls):
    """A SparseLP-like object (class ``cls``) holding the fixture's LP."""
    lp = cls()
    n = d["c"].size
    lp.nb_variables = n
    lp.costsvector = d["c"].copy()
    lp.lower_bounds = d["lb"].copy()
    lp.upper_bounds = d["ub"].copy()
    lp.is_integer = np.zeros(n, dtype=bool)
    lp.a_equalities = csr_of(d, "Ae")
    lp.b_equalities = d["be"].copy()
    lp.a_inequalities = csr_of(d, "Ai")
    lp.b_lower = None if bool(d["bl_none"]) else d["bl"].copy()
    lp.b_upper = d["bu"].copy()
    return lp


class Recorder:
    """Collects callback calls of a solver (copies the solution)."""

    def __init__(self, keep=None):
        self.keep = None if keep is None else set(int(k) for k in keep)
        self.it, self.x, self.e1, self.e2, self.veq, self.vineq = [], [], [], [], [], []

    def __call__(self, niter, sol, e1, e2, dur, veq, vineq):
        if self.keep is None or niter in self.keep:
            self.it.append(niter)
            self.x.append(np.array(sol, dtype=np.float64, copy=True))
            self.e1.append(e1)
            self.e2.append(e2)
            self.veq.append(veq)
            self.vineq.append(vineq)
