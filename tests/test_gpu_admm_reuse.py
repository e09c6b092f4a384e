"""8-pass form of the matrix-free ADMM (the CG residual reuses the line search's
products M x and M dir): same reference iterates, same tolerance.  -m gpu."""
import numpy as np
import pytest

from conftest import Recorder, load_golden, solver_args

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["sc50a", "sc105", "potts8", "random1"])
def test_admm_cg_reuse_matches_reference_iterates(case):
    from pysparselp_amd.admm_cg import lp_admm_cg

    d = load_golden("lp_" + case)
    keep = [it for it in d["admmcg_it"] if it <= 200]
    rec = Recorder(keep)
    lp_admm_cg(*solver_args(d), nb_iter=200, callback_func=rec, nb_iter_plot=1, reuse=True)
    assert rec.it == keep
    for got, ref in zip(rec.x, d["admmcg_x"]):
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < 1e-9
