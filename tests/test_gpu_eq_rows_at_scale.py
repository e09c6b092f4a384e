"""The 10 %-equality variant of the synthetic LP (generator randomLP.py:62-68) on every at-scale format:
tall cells, LDS strips, chunked matrices cut at the boundary between the two kinds of rows -- and cut elsewhere.

Reference order (ChambollePockPPD.py:198-217): ``d = (c + y_eq * a_eq) + y_ineq * a_ineq`` -- a column's equality and
inequality terms are two sums, added to ``c`` one after the other.  The device forms them as two products over copies of
``A_e^T`` and ``A_i^T`` (views over the chunks of a chunked matrix cut at m_eq; copies of the two row ranges for an ordinary
matrix) or, where neither exists, as two products over the copy of the whole ``K^T`` with the other kind of rows masked out of
``y`` (csrc/slp_cp.hip ``cp_split_setup``).  Bar: Chambolle-Pock ``x`` AND its report BIT FOR BIT against
``oracle.chambolle_pock_ppd`` on the downloaded matrix, in every one of those forms; matrix-free ADMM <= 1e-9 against
``oracle.lp_admm_cg`` (ADMM.py:143-268 with the use_cg flags).  Full-size cases: tests/test_gpu_c3_full.py, test_gpu_c4_full.py.
-m gpu."""
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

TALL = dict(n=300_000, m=60_000, density=2e-4, seed=5)     # 60 entries per row, 0.8 per (row, 4096 columns) in both orientations
STRIPS = dict(n=20_000, m=30_000, density=2e-3, seed=6)    # 40 entries per row, 10 per (row, strip): LDS strips
SHAPES = {"tall": TALL, "strips": STRIPS}


@pytest.fixture(autouse=True)
def _small_matrices_take_the_strip_formats():
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    yield
    for k in ("SLP_STRIP_MIN_NNZ", "SLP_CP_SPLIT", "SLP_STRIP_SPLIT", "SLP_TALL_FUSE"):
        os.environ.pop(k, None)


def _lp(shape, chunks, m_eq, cut=True):
    from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix
    from pysparselp_amd.problems import random_lp_on_device

    if cut or chunks <= 1:
        return random_lp_on_device(shape["n"], shape["m"], shape["density"], seed=shape["seed"], chunks=chunks, m_eq=m_eq)
    # a chunking that does NOT respect the boundary between the two kinds of rows
    cuts = ChunkedDeviceMatrix.cuts(shape["m"], chunks)
    assert m_eq not in cuts
    a = ChunkedDeviceMatrix(shape["n"], expect_chunks=len(cuts) - 1)
    for r0, r1 in zip(cuts, cuts[1:]):
        a.append(DeviceMatrix.random(r1 - r0, shape["n"], shape["density"], shape["seed"], r0))
    return (a,) + a.random_lp_vectors(shape["density"], shape["seed"], m_eq=m_eq)


_REF = {}


def _reference(name, m_eq, iters):
    """Host copy of the LP and the oracle's Chambolle-Pock iterate + report for it (cached per shape)."""
    key = (name, m_eq, iters)
    if key not in _REF:
        shape = SHAPES[name]
        a, xf, c, lb, ub, b = _lp(shape, 1, m_eq)
        host = a.download()
        a.close()
        ae, ai = oracle.as_csr(host[:m_eq]), oracle.as_csr(host[m_eq:])
        # the generator's equality right-hand sides are A_e x_f in csr_matvec order (randomLP.py:63)
        assert np.array_equal(b[:m_eq], oracle.matvec(ae, xf))
        reports = []
        x_cpu, _ = oracle.chambolle_pock_ppd(c, ae, b[:m_eq], ai, None, b[m_eq:], lb, ub, nb_max_iter=iters, nb_iter_plot=iters - 1,
                                             callback_func=lambda *r: reports.append(r))
        _REF[key] = dict(vectors=(xf, c, lb, ub, b), ae=ae, ai=ai, x=x_cpu, report=reports[-1], host=host)
    return _REF[key]


CASES = [
    # (shape, chunks, cut at m_eq, SLP_CP_SPLIT, expected form: 1 two copies, 2 masked products)
    ("tall", 1, True, None, 1), ("tall", 1, True, "masked", 2), ("tall", 3, True, None, 1), ("tall", 8, True, None, 1),
    ("tall", 3, False, None, 2), ("tall", 8, True, "masked", 2),
    ("strips", 1, True, None, 1), ("strips", 1, True, "masked", 2), ("strips", 4, True, None, 1), ("strips", 3, False, None, 2),
]


@pytest.mark.parametrize("name, chunks, cut, forced, form", CASES)
def test_chambolle_pock_with_equality_rows_bit_for_bit(name, chunks, cut, forced, form):
    from pysparselp_amd.scale import DeviceCP

    shape = SHAPES[name]
    m_eq = shape["m"] // 10
    iters = 12
    ref = _reference(name, m_eq, iters)
    if forced:
        os.environ["SLP_CP_SPLIT"] = forced
    a, xf, c, lb, ub, b = _lp(shape, chunks, m_eq, cut)
    try:
        for got, want in zip((xf, c, lb, ub, b), ref["vectors"]):
            assert np.array_equal(got, want)               # the same LP whatever the chunking
        if chunks > 1:
            assert a.chunks >= chunks
        want_kernel = (6,) if name == "tall" else (2, 3)
        assert a.spmv_kernel(False) in want_kernel and a.spmv_kernel(True) in want_kernel
        s = DeviceCP(a, b, c, lb, ub, m_eq=m_eq)
        assert s.split_form() == form
        s.iterate(iters - 1)
        s.primal_step()
        rep = s.report()
        s.dual_step()
        x = s.x()
        s.close()
        assert np.array_equal(x, ref["x"]), float(np.max(np.abs(x - ref["x"])))
        # the reference's periodic report at the last iteration (ChambollePockPPD.py:242-329): energy1, energy2, max violations
        _, _, e1, e2, _, veq, vineq = ref["report"]
        assert rep[0] == e1 or abs(rep[0] - e1) <= 1e-9 * abs(e1)      # (two-stage device reductions: rounding only)
        assert abs(rep[1] - e2) <= 1e-9 * abs(e2)
        assert rep[2] == veq and rep[3] == vineq                       # maxima of bit-identical residuals
    finally:
        a.close()


@pytest.mark.parametrize("name, chunks", [("tall", 1), ("tall", 3), ("tall", 8), ("strips", 1), ("strips", 4)])
def test_matrix_free_admm_with_equality_rows(name, chunks):
    from pysparselp_amd.admm_cg import DeviceADMM

    shape = SHAPES[name]
    m_eq = shape["m"] // 10
    ref = _reference(name, m_eq, 12)
    xf, c, lb, ub, b = ref["vectors"]
    iters = 10
    want = oracle.lp_admm_cg(c, ref["ae"], b[:m_eq], ref["ai"], None, b[m_eq:], lb, ub, nb_iter=iters - 1, nb_iter_plot=10 ** 9)
    a = _lp(shape, chunks, m_eq)[0]
    try:
        s = DeviceADMM(a, b, c, lb, ub, m_eq=m_eq)
        s.iterate(iters)
        got = s.x(shape["n"])
        rep = s.report()
        s.close()
        err = float(np.max(np.abs(got - want) / (1 + np.abs(want))))
        assert err <= 1e-9, err
        assert abs(float(c.dot(got)) - float(c.dot(want))) <= 1e-6 * abs(float(c.dot(want)))
        assert np.all(np.isfinite(rep))
    finally:
        a.close()


def test_fused_and_chunk_by_chunk_launches_give_the_same_bits_with_equality_rows():
    from pysparselp_amd.scale import DeviceCP

    m_eq = TALL["m"] // 10
    ref = _reference("tall", m_eq, 12)
    xf, c, lb, ub, b = ref["vectors"]
    for fuse in ("1", "0"):
        os.environ["SLP_TALL_FUSE"] = fuse
        a = _lp(TALL, 8, m_eq)[0]
        try:
            s = DeviceCP(a, b, c, lb, ub, m_eq=m_eq)
            s.iterate(12)
            assert np.array_equal(s.x(), ref["x"]), fuse
            s.close()
        finally:
            a.close()


def test_b_upper_does_not_depend_on_the_strip_range_split():
    """``b_upper = ceil((A x_f + ...) 1000) / 1000`` shows the order of a row's additions in a tenth of the rows.  A row
    block with few row blocks (a 1/8 row partition) builds its LDS-strip copy with a strip-range split (S > 1: partial sums added in
    range order); the generator asks for the SEQUENTIAL order, which such a copy now serves with one workgroup per row block --
    the same ``b_upper`` for every partition of the rows."""
    from pysparselp_amd.problems import random_lp_on_device

    sh = STRIPS
    whole = random_lp_on_device(sh["n"], sh["m"], sh["density"], seed=sh["seed"])
    b_whole = whole[5]
    host = whole[0].download()
    x = np.random.RandomState(3).randn(sh["n"])
    whole[0].close()
    os.environ["SLP_STRIP_SPLIT"] = "4"
    parts = []
    for k in range(8):
        r0, r1 = sh["m"] * k // 8, sh["m"] * (k + 1) // 8
        got = random_lp_on_device(sh["n"], sh["m"], sh["density"], seed=sh["seed"], row_offset=r0, rows=r1 - r0)
        assert got[0].spmv_kernel(False) in (2, 3)
        parts.append(got[5])
        if k == 3:   # the split is live: the ordinary product of this block is partial sums in range order (rounding-level differences)
            split = got[0].matvec(x)
            seq = oracle.matvec(oracle.as_csr(host[r0:r1]), x)
            assert np.max(np.abs(split - seq)) <= 1e-12 and not np.array_equal(split, seq)
            assert np.array_equal(got[0].matvec(x, order=1), seq)     # SLP_ORDER_SEQUENTIAL: the single chain
        got[0].close()
    assert np.array_equal(np.concatenate(parts), b_whole)


@pytest.mark.parametrize("name, m_eq, form", [("tall", 6001, 2), ("tall", 2, 2), ("tall", 59_998, 2), ("strips", 3001, 2), ("strips", 2, 2),
                                               ("strips", 29_000, 1)])
def test_awkward_numbers_of_equality_rows(name, m_eq, form):
    """Edge cases of the split: an ODD number of equality rows (the strip kernels stage y + m_eq by 16-byte loads: no row-range copies),
    two equality rows or nearly all rows equalities (a row range too small for a strip copy of its own) -- the two masked products over
    the whole copy take over, bit for bit the oracle's iterate like every other form."""
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    shape = SHAPES[name]
    a, xf, c, lb, ub, b = random_lp_on_device(shape["n"], shape["m"], shape["density"], seed=shape["seed"], m_eq=m_eq)
    try:
        host = a.download()
        want, _ = oracle.chambolle_pock_ppd(c, host[:m_eq], b[:m_eq], host[m_eq:], None, b[m_eq:], lb, ub, nb_max_iter=8, nb_iter_plot=10 ** 9)
        s = DeviceCP(a, b, c, lb, ub, m_eq=m_eq)
        assert s.split_form() == form, s.split_form()
        s.iterate(8)
        assert np.array_equal(s.x(), want)
        s.close()
    finally:
        a.close()
