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
                # a NEW Chambolle-Pock solver can still be set up: its preconditioners are products over the strip copies
                late = DeviceCP(a, b, c, lb, ub)
                late.iterate(40)
                x_late = late.x()
                late.close()
            cp.iterate(40)
            admm.iterate(40)
            if release:
                assert np.array_equal(x_late, cp.x())   # set up after the release == set up before it
            # the reference's periodic report (ChambollePockPPD.py:242-329, ADMM.py:213-248) between the two halves of an
            # iteration: formed through the strip copies, so it needs no CSR arrays and does not change with the release
            cp.primal_step()
            cp_report = cp.report()
            cp.dual_step()
            admm.xstep()
            admm_report = admm.report()
            admm.multiplier_step()
            assert np.all(np.isfinite(cp_report)) and np.all(np.isfinite(admm_report))
            out.append((cp.x(), admm.x(30000), a.matvec(xf), a.rmatvec(b), cp_report, admm_report))
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


def test_matrix_guards_raise_clear_errors():
    """bad CSR input is an error at the boundary, not an out-of-bounds device access; a matrix that was
    row-normalised in place is not scaled twice; derived copies are not pulled from under a live solver."""
    import scipy.sparse

    from pysparselp_amd import _lib
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.device import DeviceMatrix
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    l = _lib.lib()
    indptr = np.array([0, 2, 4], dtype=np.int64)
    data = np.ones(4)
    for bad_idx, what in ((np.array([0, 5, 1, 2], dtype=np.int32), "out of range"), (np.array([0, -1, 1, 2], dtype=np.int32), "out of range")):
        h = l.slp_matrix_create(2, 3, _lib.ptr(indptr), _lib.ptr(bad_idx), _lib.ptr(data))
        assert not h and what in _lib.last_error()
    bad_ptr = np.array([0, 3, 2], dtype=np.int64)
    h = l.slp_matrix_create(2, 3, _lib.ptr(bad_ptr), _lib.ptr(np.zeros(2, dtype=np.int32)), _lib.ptr(np.ones(2)))
    assert not h and "non-decreasing" in _lib.last_error()

    a, xf, c, lb, ub, b = random_lp_on_device(2000, 3000, 0.01, seed=2)   # CSR kernels, no dictionary strips: scaling in place
    cp = DeviceCP(a, b, c, lb, ub)
    with pytest.raises(_lib.SlpError, match="still alive"):
        a.set_format(1)                       # would free copies the live solver points into
    with pytest.raises(_lib.SlpError, match="still alive"):
        DeviceADMM(a, b, c, lb, ub)           # in-place row normalisation under a live solver
    cp.close()
    s1 = DeviceADMM(a, b, c, lb, ub)
    s1.iterate(3)
    s1.close()
    with pytest.raises(_lib.SlpError, match="already row-normalised"):
        DeviceADMM(a, b, c, lb, ub)           # the matrix now holds scaled values: scaling again would solve another LP
    a.close()


def test_release_with_equality_rows_keeps_the_reference_order():
    """Chambolle-Pock with equality AND inequality rows keeps the two partial sums of a column apart ((c + s_eq) + s_ineq,
    ChambollePockPPD.py:206,216).  Rounds 2-5: only the CSR walk gave them, such a solver pinned the CSR arrays and the release
    was refused.  Round 6: the solver owns strip copies of A_e^T and A_i^T (csrc/slp_cp.hip cp_split_setup), so the release is
    allowed while it lives and its iterates go on unchanged; a solver created AFTER the release (no CSR to cut the row ranges
    from) forms the two sums by two masked products over the whole copy -- every form bit for bit the oracle's iterate."""
    from oracle import oracle
    from pysparselp_amd._lib import ORDER_SEQUENTIAL
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        m_eq = 5000
        a, xf, c, lb, ub, b_eq = random_lp_on_device(30000, 40000, 0.001, seed=7, m_eq=m_eq)   # the first 5000 rows: a_i x = a_i x_f
        host = a.download()
        want, _ = oracle.chambolle_pock_ppd(c, host[:m_eq], b_eq[:m_eq], host[m_eq:], None, b_eq[m_eq:], lb, ub, nb_max_iter=10,
                                            nb_iter_plot=10 ** 9)
        seq = DeviceCP(a, b_eq, c, lb, ub, order=ORDER_SEQUENTIAL, m_eq=m_eq)
        assert seq.split_form() == 1                # copies of the two row ranges, the solver's own
        seq.iterate(5)
        x5 = seq.x()
        a.release_csr()                             # allowed: no iteration half walks the CSR arrays
        seq.iterate(5)
        x10 = seq.x()
        seq.close()
        late = DeviceCP(a, b_eq, c, lb, ub, m_eq=m_eq)
        assert late.split_form() == 2               # no CSR left to cut row ranges from: masked products over the whole copy
        late.iterate(10)
        y10 = late.x()
        late.close()
        assert not np.array_equal(x5, x10)
        assert np.array_equal(x10, want) and np.array_equal(y10, want)
        a.close()
    finally:
        del os.environ["SLP_STRIP_MIN_NNZ"]
