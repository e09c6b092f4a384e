"""Chunked matrices (csrc/slp_chunked.hip, ``ChunkedDeviceMatrix``): a constraint matrix handed over / generated in row
chunks whose CSR never coexists -- how BASELINE config 4's 1e7 x 2e7 LP (2e10 entries) becomes resident on ONE GPU.
Reference products: ``a * x`` / ``y * a`` (ChambollePockPPD.py:206,216,235,240 ; ADMM.py:148,262), restated in
oracle/slp_oracle.c; solvers ChambollePockPPD.py:195-343 and ADMM.py:143-268 (use_cg flags), restated in oracle/oracle.py.

Bar: for ANY chunking (K = 1, 3, 8) ``A x`` and ``A^T y`` equal the unchunked product AND the oracle bit for bit -- chunk k
continues the column sums of chunk k - 1, so every sum stays the single chain of the CSR walk; Chambolle-Pock ``x`` bit for
bit vs unchunked and vs the oracle; matrix-free ADMM <= 1e-12 vs unchunked, <= 1e-9 vs the oracle.  Shapes are reduced
copies of config 4's (tall cells in both orientations) and of config 3's (LDS strips).  The copy of a chunk's transpose is
built straight from the chunk's CSR (no transposed CSR): compared with the device transposition of the whole matrix.
The full-size run is tools/c4_full.py (profiles/r04_bench_*_c4_1gpu.json).  -m gpu."""
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_matrices_take_the_strip_formats():
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    yield
    del os.environ["SLP_STRIP_MIN_NNZ"]
    for k in ("SLP_TALL_SPLIT", "SLP_TALL_PASS_NNZ", "SLP_TALL_R"):
        os.environ.pop(k, None)


TALL = dict(n=300_000, m=60_000, density=2e-4, seed=5)     # 60 entries per row, 0.8 per (row, 4096 columns) in both orientations
STRIPS = dict(n=20_000, m=30_000, density=2e-3, seed=6)    # 40 entries per row, 10 per (row, strip): LDS strips


def _lp(shape, chunks):
    from pysparselp_amd.problems import random_lp_on_device

    return random_lp_on_device(shape["n"], shape["m"], shape["density"], seed=shape["seed"], chunks=chunks)


@pytest.fixture(scope="module")
def tall_reference():
    """The unchunked LP, its products and its host copy (the oracle's operand)."""
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    a, xf, c, lb, ub, b = _lp(TALL, 1)
    rng = np.random.RandomState(1)
    x, y = rng.randn(TALL["n"]), rng.randn(TALL["m"])
    assert a.spmv_kernel(False) == 6 and a.spmv_kernel(True) == 6
    host = a.download()
    ref = dict(a=a, vectors=(xf, c, lb, ub, b), x=x, y=y, ax=a.matvec(x), aty=a.rmatvec(y), host=host)
    # the unchunked tall-cell products are the oracle's, bit for bit (also pinned in tests/test_gpu_tall.py)
    assert np.array_equal(ref["ax"], oracle.matvec(oracle.as_csr(host), x))
    assert np.array_equal(ref["aty"], oracle.rmatvec(oracle.as_csr(host), y))
    yield ref
    a.close()


@pytest.mark.parametrize("chunks", [2, 3, 8])
def test_chunked_products_equal_the_unchunked_ones_and_the_oracle_bit_for_bit(tall_reference, chunks):
    ref = tall_reference
    a, xf, c, lb, ub, b = _lp(TALL, chunks)
    try:
        assert a.chunks == chunks and a.shape == (TALL["m"], TALL["n"]) and a.nnz == ref["a"].nnz
        assert a.spmv_kernel(False) == 6 and a.spmv_kernel(True) == 6
        for got, want in zip((xf, c, lb, ub, b), ref["vectors"]):
            assert np.array_equal(got, want)               # the same LP whatever the chunking
        assert np.array_equal(a.matvec(ref["x"]), ref["ax"])
        assert np.array_equal(a.rmatvec(ref["y"]), ref["aty"])
        # twice: the accumulation across chunks starts afresh in every product
        assert np.array_equal(a.rmatvec(ref["y"]), ref["aty"])
        lhs, rhs = float(a.matvec(ref["x"]).dot(ref["y"])), float(ref["x"].dot(a.rmatvec(ref["y"])))
        assert abs(lhs - rhs) <= 1e-10 * (abs(lhs) + 1)
        assert a.bench_spmv(False, reps=2) > 0 and a.bench_spmv(True, reps=2) > 0
        assert 0 < a._l.slp_matrix_format_bytes(a._h, 0) < 12 * a.nnz   # 5-6 bytes per entry, not the CSR's 12
        with pytest.raises(Exception, match="chunked"):
            a.download()
    finally:
        a.close()


@pytest.mark.parametrize("chunks", [3, 8])
def test_one_launch_per_product_equals_the_chunk_by_chunk_launches(tall_reference, chunks, monkeypatch):
    """Round 5: when every chunk runs on tall cells the product of a chunked matrix is ONE launch -- ``A x``: a grid over the
    row blocks of all chunks; ``A^T y``: one workgroup per column block walks the chunks in order, the column sums stay in LDS
    (csrc/slp_tall_spmv.hip ``tall_spmv_fused``).  ``SLP_TALL_FUSE=0`` keeps the round-4 form (a launch per chunk, the sums
    handed over through the output vector): the same bits either way, and the unchunked product's."""
    ref = tall_reference
    lib = ref["a"]._l
    got = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("SLP_TALL_FUSE", fuse)
        a = _lp(TALL, chunks)[0]
        try:
            launches = (int(lib.slp_matrix_product_launches(a._h, 0)), int(lib.slp_matrix_product_launches(a._h, 1)))
            assert launches == ((1, 1) if fuse == "1" else (chunks, chunks))
            got[fuse] = (a.matvec(ref["x"]), a.rmatvec(ref["y"]), a.rmatvec(ref["y"]), a.abs_pow_matvec(ref["x"], 2.0),
                         a.abs_pow_matvec(ref["y"], 1.0, transposed=True))
        finally:
            a.close()
    for u, w in zip(got["1"], got["0"]):
        assert np.array_equal(u, w)
    assert np.array_equal(got["1"][0], ref["ax"]) and np.array_equal(got["1"][1], ref["aty"]) and np.array_equal(got["1"][2], ref["aty"])
    assert int(lib.slp_matrix_product_launches(ref["a"]._h, 0)) == 1


@pytest.mark.parametrize("chunks", [3, 8])
def test_chunked_solvers_equal_the_unchunked_ones_and_the_oracle(tall_reference, chunks):
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.scale import DeviceCP

    ref = tall_reference
    xf, c, lb, ub, b = ref["vectors"]
    host = oracle.as_csr(ref["host"])
    a = _lp(TALL, chunks)[0]
    try:
        iters = 12
        x_cpu, _ = oracle.chambolle_pock_ppd(c, None, None, host, None, b, lb, ub, nb_max_iter=iters, nb_iter_plot=10 ** 9)
        xs = []
        for mat in (ref["a"], a):
            s = DeviceCP(mat, b, c, lb, ub)
            s.iterate(iters - 1)
            s.primal_step()
            rep = s.report()
            s.dual_step()
            xs.append((s.x(), rep))
            s.close()
        assert np.array_equal(xs[0][0], xs[1][0]) and np.array_equal(xs[0][1], xs[1][1])   # iterate AND report: bit for bit
        assert np.array_equal(xs[1][0], x_cpu)
        x_cpu = oracle.lp_admm_cg(c, None, None, host, None, b, lb, ub, nb_iter=iters - 1, nb_iter_plot=10 ** 9)
        xs = []
        for mat in (ref["a"], a):
            s = DeviceADMM(mat, b, c, lb, ub)
            assert s.reuse == 4
            s.iterate(iters)
            xs.append(s.x(TALL["n"]))
            rep = s.report()
            assert np.all(np.isfinite(rep))
            s.close()
        err = float(np.max(np.abs(xs[0] - xs[1]) / (1 + np.abs(xs[0]))))
        assert err <= 1e-12, err
        err = float(np.max(np.abs(xs[1] - x_cpu) / (1 + np.abs(x_cpu))))
        assert err <= 1e-9, err
        assert abs(float(c.dot(xs[1])) - float(c.dot(x_cpu))) <= 1e-6 * abs(float(c.dot(x_cpu)))
    finally:
        a.close()


def test_strip_format_chunks_and_fp64_chunks():
    """Chunks dense enough for the LDS strips (config 3's regime): dictionary strips and, with the dictionary ruled out
    chunk by chunk, fp64 strips; the products and Chambolle-Pock bit for bit, ADMM refused without a dictionary."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix
    from pysparselp_amd.scale import DeviceCP

    sh = STRIPS
    a0, xf, c, lb, ub, b = _lp(sh, 1)
    host = oracle.as_csr(a0.download())
    rng = np.random.RandomState(2)
    x, y = rng.randn(sh["n"]), rng.randn(sh["m"])
    ax, aty = oracle.matvec(host, x), oracle.rmatvec(host, y)
    x_cpu, _ = oracle.chambolle_pock_ppd(c, None, None, host, None, b, lb, ub, nb_max_iter=10, nb_iter_plot=10 ** 9)
    try:
        for policy in (0, 1):
            cuts = ChunkedDeviceMatrix.cuts(sh["m"], 3)
            a = ChunkedDeviceMatrix(sh["n"])
            for r0, r1 in zip(cuts, cuts[1:]):
                chunk = DeviceMatrix.random(r1 - r0, sh["n"], sh["density"], sh["seed"], r0)
                chunk.set_format(policy)
                a.append(chunk)
                assert chunk._h is None
            assert a.spmv_kernel(False) in ((2, 3) if policy == 0 else (1,)), a.spmv_kernel(False)
            assert a.spmv_kernel(True) in ((2, 3) if policy == 0 else (1,)), a.spmv_kernel(True)
            assert np.array_equal(a.matvec(x), ax) and np.array_equal(a.rmatvec(y), aty)
            s = DeviceCP(a, b, c, lb, ub)
            s.iterate(10)
            assert np.array_equal(s.x(), x_cpu), policy
            s.close()
            if policy == 1:
                with pytest.raises(Exception, match="value-dictionary"):
                    DeviceADMM(a, b, c, lb, ub)
            else:
                s = DeviceADMM(a, b, c, lb, ub)
                s.iterate(8)
                s0 = DeviceADMM(a0, b, c, lb, ub)
                s0.iterate(8)
                err = float(np.max(np.abs(s.x(sh["n"]) - s0.x(sh["n"])) / (1 + np.abs(s0.x(sh["n"])))))
                s.close()
                s0.close()
                assert err <= 1e-12, err
            a.close()
    finally:
        a0.close()


def test_build_passes_and_strip_range_split_do_not_change_the_copy():
    """The tall-cell copies are written in passes over ranges of row blocks (bounded temporaries): 1 pass and many passes give
    the same products; with the strip-range split (partial sums per range) chunks still continue each other's sums."""
    ref = None
    for env in ({}, {"SLP_TALL_PASS_NNZ": "400000"}, {"SLP_TALL_PASS_NNZ": "400000", "SLP_TALL_R": "1500"}):
        os.environ.update(env)
        a = _lp(TALL, 3)[0]
        rng = np.random.RandomState(9)
        x, y = rng.randn(TALL["n"]), rng.randn(TALL["m"])
        got = (a.matvec(x), a.rmatvec(y))
        a.close()
        for k in env:
            del os.environ[k]
        if ref is None:
            ref = got
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), env
    os.environ["SLP_TALL_SPLIT"] = "3"
    a = _lp(TALL, 3)[0]
    got = (a.matvec(x), a.rmatvec(y))
    a.close()
    for g, r in zip(got, ref):
        assert float(np.max(np.abs(g - r) / (1 + np.abs(r)))) <= 1e-13


def test_a_host_matrix_uploaded_in_chunks():
    """``ChunkedDeviceMatrix.from_csr``: a scipy CSR handed over in row chunks of a given entry count."""
    import scipy.sparse

    from pysparselp_amd.device import ChunkedDeviceMatrix

    rng = np.random.RandomState(12)
    m, n, k = 30_001, 200_000, 1_500_000
    a = scipy.sparse.coo_matrix((np.round(rng.randn(k), 1) + 0.05, (rng.randint(0, m, size=k), rng.randint(0, n, size=k))), shape=(m, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    g = ChunkedDeviceMatrix.from_csr(a, chunk_entries=400_000)
    try:
        assert g.chunks == 4 and g.shape == a.shape and g.nnz == a.nnz
        x, y = rng.randn(n), rng.randn(m)
        assert np.array_equal(g.matvec(x), oracle.matvec(oracle.as_csr(a), x))
        assert np.array_equal(g.rmatvec(y), oracle.rmatvec(oracle.as_csr(a), y))
    finally:
        g.close()


def test_chunked_api_errors():
    from pysparselp_amd._lib import SlpError
    from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix

    a = ChunkedDeviceMatrix(50_000)
    odd = DeviceMatrix.random(10_001, 50_000, 1e-3, 1, 0)
    a.append(odd)                                   # an odd chunk may only be the last one
    nxt = DeviceMatrix.random(10_000, 50_000, 1e-3, 1, 10_001)
    with pytest.raises(SlpError, match="even number of rows"):
        a.append(nxt)
    nxt.close()
    other = DeviceMatrix.random(10_000, 40_000, 1e-3, 1, 0)
    with pytest.raises(SlpError, match="column count"):
        ChunkedDeviceMatrix(50_000).append(other)
    other.close()
    plain = DeviceMatrix.random(10_000, 50_000, 1e-3, 1, 0)
    extra = DeviceMatrix.random(10_000, 50_000, 1e-3, 1, 0)
    assert plain._l.slp_matrix_chunked_append(plain._h, extra._h) != 0      # the target must come from slp_matrix_chunked_create
    assert b"not a chunked matrix" in plain._l.slp_last_error()
    extra.close()
    plain.close()
    a.close()


def test_randomised_chunkings_match_the_oracle_bit_for_bit():
    """A slim run of tools/fuzz_chunked.py (random shapes in the tall-cell and LDS-strip regimes, 2-6 chunks with random even cuts,
    empty rows, forced block heights, value dictionary or fp64 entries): products and Chambolle-Pock iterates against the oracle."""
    import importlib.util

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_chunked", os.path.join(repo, "tools", "fuzz_chunked.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen, skipped, solved = mod.run(24, seed=21)
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"   # (the fixture removes it again)
    assert sum(seen.values()) >= 2 * 12 and solved >= 3, (seen, skipped, solved)


def _device_products(a):
    """The chunked / unchunked device matrix as the oracle's products-only operand (oracle.ProductsOnly)."""
    return oracle.ProductsOnly(a.shape, a.matvec, a.rmatvec, lambda x, p: a.abs_pow_matvec(x, p), lambda y, p: a.abs_pow_matvec(y, p, transposed=True))


@pytest.mark.parametrize("shape_name, chunks", [("tall", 1), ("tall", 3), ("strips", 1), ("strips", 4)])
def test_abs_power_products_and_the_oracle_iterating_over_device_products(tall_reference, shape_name, chunks):
    """``slp_matrix_spmv_abs_pow`` (the sums behind Chambolle-Pock's preconditioners, ChambollePockPPD.py:122-179) against the
    oracle's products with ``|A| ** p``, bit for bit; and the construction tests/test_gpu_c4_full.py uses at a size whose CSR
    no host holds: the ORACLE's Chambolle-Pock iteration fed with the device's products gives, bit for bit, what the oracle
    gives on the host CSR -- and what the device solver gives."""
    from pysparselp_amd.scale import DeviceCP

    shape = TALL if shape_name == "tall" else STRIPS
    a, xf, c, lb, ub, b = _lp(shape, chunks)
    host = tall_reference["host"] if shape_name == "tall" else _lp(shape, 1)[0].download()
    oa = oracle.as_csr(host)
    rng = np.random.RandomState(3)
    x, y = rng.randn(shape["n"]), rng.randn(shape["m"])
    for p in (1.0, 2.0, 0.5):
        powered = oracle.Csr(oa.indptr, oa.indices, np.abs(oa.data) ** p, oa.shape)
        assert np.array_equal(a.abs_pow_matvec(x, p), oracle.matvec(powered, x)), p
        assert np.array_equal(a.abs_pow_matvec(y, p, transposed=True), oracle.rmatvec(powered, y)), p
    iters = 5
    want, _ = oracle.chambolle_pock_ppd(c, None, None, oa, None, b, lb, ub, nb_max_iter=iters, nb_iter_plot=10 ** 9)
    through, _ = oracle.chambolle_pock_ppd(c, None, None, _device_products(a), None, b, lb, ub, nb_max_iter=iters, nb_iter_plot=10 ** 9)
    assert np.array_equal(through, want)
    s = DeviceCP(a, b, c, lb, ub)
    s.iterate(iters)
    assert np.array_equal(s.x(), want)
    s.close()
    # matrix-free ADMM: the oracle on the CSR, the oracle over the device's products (scalings beside the products) and the device
    from pysparselp_amd.admm_cg import DeviceADMM

    want = oracle.lp_admm_cg(c, None, None, oa, None, b, lb, ub, nb_iter=iters - 1, nb_iter_plot=10 ** 9)
    through = oracle.lp_admm_cg(c, None, None, _device_products(a), None, b, lb, ub, nb_iter=iters - 1, nb_iter_plot=10 ** 9)
    s = DeviceADMM(a, b, c, lb, ub)
    s.iterate(iters)
    got = s.x(shape["n"])
    s.close()
    for u, v in ((through, want), (got, want), (got, through)):
        assert np.max(np.abs(u - v) / (1 + np.abs(v))) <= 1e-9
    a.close()


@pytest.mark.parametrize("eq_frac", [0.0, 0.1])
def test_oracle_streaming_its_own_products_equals_the_device_solvers(eq_frac):
    """What tools/c4_streamed_parity.py does at config 4's full size (profiles/r06_c4_streamed_oracle_parity.json), in small:
    ``oracle.StreamedCsr`` holds the LP one row chunk at a time -- regenerated by the counter-based generator, downloaded, multiplied
    by the oracle's C kernels, dropped -- so NO device product enters the oracle's iteration.  Chambolle-Pock x bit for bit, matrix-free
    ADMM <= 1e-9, with and without equality rows in chunks of their own."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    sh = TALL
    m_eq = (int(sh["m"] * eq_frac)) & ~1
    a, xf, c, lb, ub, b = random_lp_on_device(sh["n"], sh["m"], sh["density"], seed=sh["seed"], chunks=3, m_eq=m_eq)
    try:
        cuts = ChunkedDeviceMatrix.cuts(sh["m"], 3, cut_at=m_eq)
        asked = []

        def source(k):
            asked.append(k)
            blk = DeviceMatrix.random(cuts[k + 1] - cuts[k], sh["n"], sh["density"], sh["seed"], cuts[k])
            host = blk.download()
            blk.close()
            return host

        st = oracle.StreamedCsr(sh["n"], cuts, source)
        ops_eq = st.row_range(0, m_eq) if m_eq else None
        ops_in = st.row_range(m_eq, sh["m"])
        beq = b[:m_eq] if m_eq else None
        want, _ = oracle.chambolle_pock_ppd(c, ops_eq, beq, ops_in, None, b[m_eq:], lb, ub, nb_max_iter=6, nb_iter_plot=10 ** 9)
        s = DeviceCP(a, b, c, lb, ub, m_eq=m_eq)
        s.iterate(6)
        assert np.array_equal(s.x(), want)
        s.close()
        want = oracle.lp_admm_cg(c, ops_eq, beq, ops_in, None, b[m_eq:], lb, ub, nb_iter=4, nb_iter_plot=10 ** 9)
        s = DeviceADMM(a, b, c, lb, ub, m_eq=m_eq)
        s.iterate(5)
        got = s.x(sh["n"])
        s.close()
        assert float(np.max(np.abs(got - want) / (1 + np.abs(want)))) <= 1e-9
        assert len(asked) > 3 * 10            # every product streamed the chunks again (nothing cached, nothing from the device copies)
    finally:
        a.close()
