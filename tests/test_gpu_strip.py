"""The LDS-tiled strip SpMV (csrc/slp_strip.hip) against the oracle.  It sums every
row with one accumulator in storage order, so it must reproduce the oracle's
csr_matvec BIT FOR BIT -- also for the device-built transpose.  -m gpu."""
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


KERNEL_CODE = {"quads": 3, "pairs": 2, "fp64": 1}


@pytest.fixture(params=["quads", "pairs", "fp64"])
def strips_everywhere(monkeypatch, request):
    """Force the strip format on small matrices (it normally starts at 3e7 stored entries): the two
    value-dictionary variants (the generated coefficients are rounded to 0.01: ~1000 distinct values; quads =
    4096-row blocks with 3-byte entries, pairs = 2048-row blocks with 4-byte entries) and fp64 entries."""
    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", "1")
    monkeypatch.setenv("SLP_VALUE_DICT", "0" if request.param == "fp64" else "1")
    monkeypatch.setenv("SLP_DICT_VARIANT", "1" if request.param == "pairs" else "2")
    yield request.param
    for k in ("SLP_STRIP_MIN_NNZ", "SLP_VALUE_DICT", "SLP_DICT_VARIANT"):
        monkeypatch.delenv(k, raising=False)


def _kernel(dm, transposed=0):
    from pysparselp_amd import _lib

    return _lib.lib().slp_matrix_spmv_kernel(dm._h, transposed)


@pytest.mark.parametrize("n,m,p", [(30000, 2500, 0.001), (20000, 3000, 0.002), (70000, 1100, 0.0008), (8192, 1024, 0.004)])
def test_strip_spmv_bit_exact(strips_everywhere, n, m, p):
    """Several strips, partial last strip / last row block, rows with no entry in a strip."""
    from pysparselp_amd.problems import random_lp_on_device

    a = random_lp_on_device(n, m, p, seed=n % 7)[0]
    s = a.download()
    oa = oracle.as_csr(s)
    rng = np.random.RandomState(1)
    x, y = rng.randn(n), rng.randn(m)
    assert np.array_equal(a.matvec(x), oracle.matvec(oa, x))
    assert np.array_equal(a.rmatvec(y), oracle.rmatvec(oa, y))
    want = KERNEL_CODE[strips_everywhere]
    assert _kernel(a, 0) == want and _kernel(a, 1) in (0, want, 6, 7)  # short transposed rows: CSR kernel or tall cells


def test_value_dictionary_only_when_few_distinct_values(monkeypatch):
    """Continuous coefficients (more than 2048 distinct values): fp64 strips.  +-0.0 and denormals are kept apart."""
    import scipy.sparse
    from pysparselp_amd.device import DeviceMatrix

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", "1")
    rng = np.random.RandomState(5)
    a = scipy.sparse.random(3000, 20000, density=0.002, random_state=rng, format="csr")
    a.sort_indices()
    x = rng.randn(20000)
    dm = DeviceMatrix.from_csr(a)
    assert np.array_equal(dm.matvec(x, 1), oracle.matvec(oracle.as_csr(a), x))
    assert _kernel(dm) == 1
    b = a.copy()
    b.data = rng.choice(np.array([1.0, -1.0, 0.5, -0.0, 0.0, 5e-324, 1e300, -2.5]), size=b.nnz)
    dm = DeviceMatrix.from_csr(b)
    y = rng.randn(3000)
    assert np.array_equal(dm.matvec(x, 1), oracle.matvec(oracle.as_csr(b), x))
    assert np.array_equal(dm.rmatvec(y, 1), oracle.rmatvec(oracle.as_csr(b), y))
    assert _kernel(dm) == 2 and _kernel(dm, 1) == 2  # few row blocks: the pair variant


def test_strip_falls_back_when_a_row_is_too_dense(strips_everywhere):
    """>= 256 entries of one row inside one 8192-column strip: the generic kernel takes over."""
    import scipy.sparse
    from pysparselp_amd.device import DeviceMatrix

    rng = np.random.RandomState(0)
    a = scipy.sparse.random(300, 20000, density=0.003, random_state=rng, format="lil")
    a[7, :400] = 1.5
    a = a.tocsr()
    a.sort_indices()
    dm = DeviceMatrix.from_csr(a)
    x = rng.randn(20000)
    assert np.array_equal(dm.matvec(x, 1), oracle.matvec(oracle.as_csr(a), x))


def test_solvers_on_strip_path_match_oracle(strips_everywhere):
    """CP and matrix-free ADMM through the strip kernels (long rows in both orientations)."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    n, m, p = 30000, 40000, 0.001
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=3)
    s = a.download()
    cp = DeviceCP(a, b, c, lb, ub)
    cp.iterate(40)
    x = cp.x()
    cp.close()
    xo, _ = oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=40, nb_iter_plot=10 ** 9)
    assert np.array_equal(x, xo)  # sequential-order sums on both sides
    admm = DeviceADMM(a, b, c, lb, ub)
    admm.iterate(25)
    x = admm.x(n)
    admm.close()
    xo = oracle.lp_admm_cg(c, None, None, s, None, b, lb, ub, nb_iter=24, nb_iter_plot=10 ** 9)
    assert np.max(np.abs(x - xo) / (1 + np.abs(xo))) < 1e-9
    assert abs(c.dot(x) - c.dot(xo)) <= 1e-6 * abs(c.dot(xo))


@pytest.mark.parametrize("split", ["2", "4"])
def test_strip_split_over_workgroups(strips_everywhere, monkeypatch, split):
    """Few row blocks (a 1/8 row partition): every block's strips are shared by several workgroups and the
    partial row sums are added in strip order -- same terms, another association than the single chain."""
    from pysparselp_amd.problems import random_lp_on_device

    monkeypatch.setenv("SLP_STRIP_SPLIT", split)
    n, m, p = 60000, 3000, 0.001
    a = random_lp_on_device(n, m, p, seed=1)[0]
    s = a.download()
    rng = np.random.RandomState(2)
    x, y = rng.randn(n), rng.randn(m)
    ax, ref = a.matvec(x), oracle.matvec(oracle.as_csr(s), x)
    assert np.max(np.abs(ax - ref)) <= 1e-13 * np.max(np.abs(s).dot(np.abs(x)))
    assert not np.array_equal(ax, ref) or split == "1"
    aty, ref = a.rmatvec(y), oracle.rmatvec(oracle.as_csr(s), y)
    assert np.max(np.abs(aty - ref)) <= 1e-13 * max(1e-300, np.max(np.abs(s).T.dot(np.abs(y))))


@pytest.mark.parametrize("dict_on", ["1", "0"])
def test_wide_strips_bit_exact(monkeypatch, dict_on):
    """Rows too sparse for the LDS tile over a width far beyond an L2 (here 4 x 10^5 columns, ~40 entries per row): strips of
    131072 columns with x gathered from L2 (k_wstrip_spmv), with the value dictionary (kernel code 4) and with fp64 entries (5).
    Since round 3 this shape runs on tall cells (tests/test_gpu_tall.py); the wide strips remain the fallback for what those
    do not take (SLP_TALL=0 here)."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", "1")
    monkeypatch.setenv("SLP_VALUE_DICT", dict_on)
    monkeypatch.setenv("SLP_TALL", "0")
    n, m, p = 400000, 300000, 1e-4
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=2)
    s = a.download()
    oa = oracle.as_csr(s)
    rng = np.random.RandomState(3)
    x, y = rng.randn(n), rng.randn(m)
    assert np.array_equal(a.matvec(x), oracle.matvec(oa, x))
    assert np.array_equal(a.rmatvec(y), oracle.rmatvec(oa, y))
    want = 4 if dict_on == "1" else 5
    assert _kernel(a, 0) == want and _kernel(a, 1) == want
    cp = DeviceCP(a, b, c, lb, ub)
    cp.iterate(10)
    xg = cp.x()
    cp.close()
    xo, _ = oracle.chambolle_pock_ppd(c, None, None, s, None, b, lb, ub, nb_max_iter=10, nb_iter_plot=10 ** 9)
    assert np.array_equal(xg, xo)
    admm = DeviceADMM(a, b, c, lb, ub)   # two-vector passes of the wide kernel
    admm.iterate(6)
    xg = admm.x(n)
    admm.close()
    xo = oracle.lp_admm_cg(c, None, None, s, None, b, lb, ub, nb_iter=5, nb_iter_plot=10 ** 9)
    assert np.max(np.abs(xg - xo) / (1 + np.abs(xo))) < 1e-9


@pytest.mark.parametrize("ndistinct,want", [(1, 2), (2048, 2), (2049, 1)])
def test_value_dictionary_size_limit(monkeypatch, ndistinct, want):
    """At most 2048 distinct stored values go to the dictionary variant; one more and the fp64 strips take over.
    NaN entries never use the dictionary (NaN != NaN would break the table lookup)."""
    import scipy.sparse
    from pysparselp_amd.device import DeviceMatrix

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", "1")
    rng = np.random.RandomState(ndistinct)
    a = scipy.sparse.random(2500, 24000, density=0.002, random_state=rng, format="csr")
    a.sort_indices()
    assert a.nnz > 3 * ndistinct
    vals = np.linspace(-3.0, 3.0, ndistinct) if ndistinct > 1 else np.array([0.25])
    a.data = np.concatenate([vals, rng.choice(vals, size=a.nnz - ndistinct)])  # every value occurs
    x = rng.randn(24000)
    dm = DeviceMatrix.from_csr(a)
    assert np.array_equal(dm.matvec(x, 1), oracle.matvec(oracle.as_csr(a), x))
    assert _kernel(dm) == want
    if ndistinct == 1:
        b = a.copy()
        b.data[5] = np.nan
        dm = DeviceMatrix.from_csr(b)
        got, ref = dm.matvec(x, 1), oracle.matvec(oracle.as_csr(b), x)
        assert _kernel(dm) == 1 and np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])


def test_randomised_shapes_all_kernel_families():
    """tools/fuzz_spmv.py: 120 random shapes around the block / strip boundaries, empty rows, both orientations, every
    kernel family (CSR, fp64 strips, dictionary pairs and quads, wide strips, tall cells) -- bit for bit against the oracle."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_spmv

    seen = fuzz_spmv.run(160, seed=1)
    assert all(seen.get(code, 0) > 0 for code in (0, 1, 2, 3, 4, 5, 6, 7)), str(sorted(seen.items()))


@pytest.mark.parametrize("fmt", ["dict", "fp64"])
@pytest.mark.parametrize("m_eq", [0, 700])
def test_cp_preconditioners_through_the_strip_copies_match_the_oracle(monkeypatch, fmt, m_eq):
    """T and Sigma (ChambollePockPPD.py:122-179) as products with ones over the strip copies -- the dictionary copy with
    a |v|^p value table, the fp64 copy with |v|^p applied on the fly -- bit for bit the oracle's column / row sums, with
    equality and inequality rows summed apart ((0 + s_eq) + s_ineq).  alpha = 0.5: |v|^0.5 is a square root on both sides
    (exact), |v|^1.5 is pow -- numpy's comes from the host's libm or its SIMD kernels, the device's from ocml; the two agree to
    rounding (a few of the ~800 distinct values differ in the last bit), so T is held to 1e-14 relative there."""
    from pysparselp_amd import _lib
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    monkeypatch.setenv("SLP_STRIP_MIN_NNZ", "1")
    monkeypatch.setenv("SLP_VALUE_DICT", "1" if fmt == "dict" else "0")
    n, m = 30000, 14000
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, 0.001, seed=4)
    s = a.download()
    for alpha in (1.0, 0.5):
        cp = DeviceCP(a, b, c, lb, ub, alpha=alpha, m_eq=m_eq)
        assert _kernel(a, 0) in ((2, 3) if fmt == "dict" else (1,)) and _kernel(a, 1) in ((2, 3) if fmt == "dict" else (1,))
        t, sig = np.empty(n), np.empty(m)
        _lib.check(cp._l.slp_cp_get_preconditioners(cp._h, _lib.ptr(t), _lib.ptr(sig)))
        cp.close()
        ae = oracle.as_csr(s[:m_eq]) if m_eq else None
        t_ref, se_ref, si_ref = oracle.cp_setup(ae, oracle.as_csr(s[m_eq:]), alpha)
        assert np.array_equal(t, t_ref) if alpha == 1.0 else np.allclose(t, t_ref, rtol=1e-14, atol=0), alpha
        assert np.array_equal(sig, np.concatenate((se_ref, si_ref)) if m_eq else si_ref), alpha
    a.close()


def test_product_timing_counts_the_products_while_it_is_on():
    """slp_product_timing (bench.py's roofline.timed_region): HIP event pairs around every product that runs through a strip copy
    while the switch is on -- the count is exact, the durations are positive, nothing is recorded while it is off, and switching it
    on again starts a new record."""
    from pysparselp_amd import _lib
    from pysparselp_amd.problems import random_lp_on_device

    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        a, xf, c, lb, ub, b = random_lp_on_device(30000, 40000, 0.001, seed=3)
        lib = _lib.lib()
        assert a.spmv_kernel(False) != 0 and a.spmv_kernel(True) != 0     # strip copies in both orientations
        out = np.zeros(3)
        a.matvec(xf)                                                        # off: not recorded
        _lib.check(lib.slp_product_timing(1))
        for _ in range(3):
            a.matvec(xf)
        a.rmatvec(b)
        _lib.check(lib.slp_product_timing(0))
        a.matvec(xf)                                                        # off again
        _lib.check(lib.slp_product_timing_read(_lib.ptr(out)))
        assert out[0] == 4 and 0.0 < out[2] <= out[1]
        _lib.check(lib.slp_product_timing(1))
        a.rmatvec(b)
        _lib.check(lib.slp_product_timing(0))
        _lib.check(lib.slp_product_timing_read(_lib.ptr(out)))
        assert out[0] == 1 and out[1] > 0.0
        a.close()
    finally:
        del os.environ["SLP_STRIP_MIN_NNZ"]
