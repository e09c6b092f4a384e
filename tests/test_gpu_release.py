"""slp_matrix_release_csr: once both orientations run on strip copies the CSR entries can be dropped (66 GB -> 17 GB of
matrix data at BASELINE config 3); iterations are unchanged bit for bit, everything that needs the entries fails
loudly.  -m gpu."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_iterations_are_unchanged_after_the_csr_entries_are_released():
    from pysparselp_amd import _lib
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        out = []
        for release in (False, True):
            a, xf, c, lb, ub, b = random_lp_on_device(30000, 40000, 0.001, seed=5)
            cp = DeviceCP(a, b, c, lb, ub)
            admm = DeviceADMM(a, b, c, lb, ub)
            if release:
                a.release_csr()
                a.release_csr()  # idempotent
                with pytest.raises(_lib.SlpError, match="released"):
                    a.download()
                with pytest.raises(_lib.SlpError, match="released"):
                    a.set_format(1)
                with pytest.raises(_lib.SlpError, match="released"):
                    DeviceCP(a, b, c, lb, ub)
            cp.iterate(40)
            admm.iterate(40)
            out.append((cp.x(), admm.x(30000), a.matvec(xf), a.rmatvec(b)))
            cp.close()
            admm.close()
            a.close()
        for u, v in zip(*out):
            assert np.array_equal(u, v)
    finally:
        del os.environ["SLP_STRIP_MIN_NNZ"]


def test_release_needs_strip_copies_in_both_orientations():
    from pysparselp_amd import _lib
    from pysparselp_amd.problems import random_lp_on_device

    a = random_lp_on_device(2000, 3000, 0.01, seed=1)[0]   # far below the strip threshold: CSR kernels
    with pytest.raises(_lib.SlpError, match="only copy"):
        a.release_csr()
    a.close()
