"""Device-side problem transforms (csrc/slp_spgemm.hip), all bit-exact:
  * M = gamma_eq A^T A + gamma_ineq I (ADMM.py:93-101) against the reference-generated fixture and the oracle's SMMP
    restatement (orc_normal_matrix), including empty columns, sums that cancel to exactly 0 and unsorted rows;
  * the column compaction of SparseLP.remove_fixed_variables (SparseLP.py:632-674) against scipy's ``a[:, free]``,
    and Chambolle-Pock over a DeviceMatrix with fixed variables against the oracle on the host-reduced LP.  -m gpu."""
import numpy as np
import pytest
import scipy.sparse

from conftest import csr_of, load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu


def _same_csr(dev, ref):
    got = dev.download()
    assert got.shape == tuple(ref.shape)
    assert np.array_equal(got.indptr, ref.indptr)
    assert np.array_equal(got.indices, ref.indices)
    assert np.array_equal(got.data, ref.data), np.max(np.abs(got.data - ref.data))


def test_normal_matrix_matches_the_reference_fixture_bit_for_bit():
    from pysparselp_amd.device import DeviceMatrix

    d = load_golden("kernel_kats")
    a3, m_ref = csr_of(d, "setup_A3"), csr_of(d, "setup_M")  # standard-form, row-normalised A and the reference's M
    a = DeviceMatrix.from_csr(a3)
    m = a.normal_matrix(2.0, 3.0)
    _same_csr(m, m_ref)
    assert np.array_equal(1.0 / m.download().diagonal(), d["setup_invD"])
    m.close()
    a.close()


@pytest.mark.parametrize("seed", range(6))
def test_normal_matrix_matches_the_oracle_on_random_matrices(seed):
    from pysparselp_amd.device import DeviceMatrix

    rng = np.random.RandomState(seed)
    rows, cols = int(rng.randint(1, 400)), int(rng.randint(1, 300))
    a = scipy.sparse.random(rows, cols, density=rng.choice([0.01, 0.05, 0.3]), random_state=rng, format="csr")
    a.data = np.round(rng.randn(a.nnz) * 4) / 4          # quarter-integers: sums cancel to exactly 0 now and then
    if seed % 2:                                         # unsorted rows: entry order inside a row is part of the contract
        perm = np.concatenate([s + rng.permutation(e - s) for s, e in zip(a.indptr[:-1], a.indptr[1:])]) if a.nnz else []
        a = scipy.sparse.csr_matrix((a.data[perm], a.indices[perm], a.indptr), shape=a.shape) if a.nnz else a
    if cols > 3:
        a = a.tolil()
        a[:, 1] = 0                                      # an empty column keeps its diagonal entry gamma_ineq
        a = a.tocsr()
        a.eliminate_zeros()
    ref = oracle.normal_matrix(oracle.as_csr(a), 2.0, 3.0)
    dev = DeviceMatrix.from_csr(a)
    m = dev.normal_matrix(2.0, 3.0)
    got = m.download()
    assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
    assert np.array_equal(got.data, ref.data)
    m.close()
    dev.close()


def test_normal_matrix_of_a_matrix_without_columns_is_empty():
    """N == 0 (no columns) used to index position total - 1 = -1 of the sort buffers."""
    import scipy.sparse
    from pysparselp_amd.device import DeviceMatrix

    dev = DeviceMatrix.from_csr(scipy.sparse.csr_matrix((5, 0)))
    m = dev.normal_matrix(2.0, 3.0)
    assert m.shape == (0, 0) and m.nnz == 0
    m.close()
    dev.close()


def test_lp_admm_forms_m_on_the_device_and_still_matches_the_reference():
    """lp_admm uploads the standard-form A once; M comes from the device product (not from scipy on the host)."""
    import os

    from conftest import Recorder, solver_args
    from pysparselp_amd.ADMM import lp_admm

    d = load_golden("lp_sc105")
    keep = [it for it in d["admm_it"] if it <= 300]
    assert os.environ.get("SLP_HOST_SPGEMM") != "1"
    rec = Recorder(keep)
    lp_admm(*solver_args(d), nb_iter=300, callback_func=rec, nb_iter_plot=1)
    assert rec.it == keep
    for got, ref in zip(rec.x, d["admm_x"]):
        assert np.array_equal(got, ref)


def test_remove_columns_matches_scipy_column_selection():
    from pysparselp_amd.device import DeviceMatrix

    rng = np.random.RandomState(3)
    a = scipy.sparse.random(500, 300, density=0.05, random_state=rng, format="csr")
    a.data = np.round(rng.randn(a.nnz) * 100) / 100
    free = rng.rand(300) > 0.3
    shift = np.where(free, 0.0, np.round(rng.randn(300) * 100) / 100)
    dev = DeviceMatrix.from_csr(a)
    red, a_shift = dev.remove_columns(free, shift)
    _same_csr(red, a[:, free])
    assert np.array_equal(a_shift, oracle.matvec(oracle.as_csr(a), shift))
    red.close()
    dev.close()


def test_device_cp_with_fixed_variables_matches_the_oracle_on_the_reduced_lp():
    """A device-resident LP whose bounds pin a fifth of the variables: DeviceCP reduces it on the device like
    SparseLP.solve does on the host (remove_fixed_variables) and must give the oracle's iterates on the reduced LP."""
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    n, m = 3000, 5000
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, 0.01, seed=7)
    s = a.download()
    lb2, ub2 = lb.copy(), ub.copy()
    pin = np.arange(n) % 5 == 0
    lb2[pin] = ub2[pin] = xf[pin]
    fixed = ~(ub2 > lb2)                      # the pinned fifth plus the few the generator fixes itself (t == 0)
    assert fixed.sum() > pin.sum()
    cp = DeviceCP(a, b, c, lb2, ub2, remove_fixed=True, order=1)  # SLP_ORDER_SEQUENTIAL: bit-exact row sums at 30 entries per row
    cp.iterate(60)
    x_red, x_full = cp.x_reduced(), cp.x()
    cp.close()
    a.close()
    free = ~fixed
    shift = np.where(free, 0.0, lb2)
    b_red = b - oracle.matvec(oracle.as_csr(s), shift)               # SparseLP.py:649-650
    x_ref, _ = oracle.chambolle_pock_ppd(c[free], None, None, s[:, free].tocsr(), None, b_red, lb2[free], ub2[free], nb_max_iter=60,
                                         nb_iter_plot=10 ** 9)
    assert np.array_equal(x_red, x_ref)
    assert np.array_equal(x_full[free], x_ref) and np.array_equal(x_full[fixed], lb2[fixed])


def test_setup_chain_on_the_device_matches_the_reference_fixture():
    """ADMM.py:76-91 on the device: row scaling of both blocks, slack standard form, row scaling of the stacked system --
    against the arrays captured from the reference (kernel_kats.npz) and against the numpy restatement in tools.py."""
    from pysparselp_amd import tools
    from pysparselp_amd.device import DeviceMatrix

    d = load_golden("kernel_kats")
    ae, ai = DeviceMatrix.from_csr(csr_of(d, "setup_Ae")), DeviceMatrix.from_csr(csr_of(d, "setup_Ai"))
    ae2, be, _ = ae.precondition_rows(d["setup_be"])
    ai2, bl, bu = ai.precondition_rows(d["setup_bl"], d["setup_bu"])
    h_ae, h_be = tools.precondition_constraints(csr_of(d, "setup_Ae"), d["setup_be"])
    h_ai, h_bl, h_bu = tools.precondition_constraints(csr_of(d, "setup_Ai"), d["setup_bl"], d["setup_bu"])
    _same_csr(ae2, h_ae)
    _same_csr(ai2, h_ai)
    assert np.array_equal(be, h_be) and np.array_equal(bl, h_bl) and np.array_equal(bu, h_bu)
    a2 = DeviceMatrix.standard_form(ae2, ai2)
    b2 = np.concatenate((be, np.zeros(ai.shape[0])))
    a3, b3, _ = a2.precondition_rows(b2)
    _same_csr(a3, csr_of(d, "setup_A3"))
    assert np.array_equal(b3, d["setup_b3"])
    assert np.array_equal(np.concatenate((d["setup_lb"], bl)), d["setup_lb2"])
    assert np.array_equal(np.concatenate((d["setup_ub"], bu)), d["setup_ub2"])
    for m in (ae, ai, ae2, ai2, a2, a3):
        m.close()


@pytest.mark.parametrize("seed", range(5))
def test_setup_transforms_match_the_host_restatement_on_random_blocks(seed):
    """Unsorted rows, empty rows, duplicate (row, column) pairs, values that underflow to 0 when scaled."""
    from pysparselp_amd import tools
    from pysparselp_amd.device import DeviceMatrix

    rng = np.random.RandomState(100 + seed)
    n, me, mi = int(rng.randint(2, 60)), int(rng.randint(0, 40)), int(rng.randint(1, 50))

    def block(rows):
        lens = rng.randint(0, 6, size=rows)
        indptr = np.concatenate(([0], np.cumsum(lens)))
        indices = rng.randint(0, n, size=indptr[-1]).astype(np.int32)  # unsorted, duplicates possible
        data = np.round(rng.randn(indptr[-1]) * 8) / 8
        if data.size > 3:
            data[0] = 5e-324      # scaled by 1/||row|| < 1 it underflows to exactly 0 and is dropped
            data[1] = 1e150      # its square is still finite
        return scipy.sparse.csr_matrix((data, indices, indptr), shape=(rows, n))

    a_eq, a_in = block(me), block(mi)
    be, bl, bu = rng.randn(me), np.where(rng.rand(mi) < 0.3, -np.inf, rng.randn(mi)), np.where(rng.rand(mi) < 0.3, np.inf, rng.randn(mi) + 3)
    c, lb, ub, x0 = rng.randn(n), -rng.rand(n), rng.rand(n), rng.randn(n)
    hae, hbe = tools.precondition_constraints(a_eq, be)
    hai, hbl, hbu = tools.precondition_constraints(a_in, bl, bu)
    c2, ha2, hb2, lb2, ub2, hx = tools.convert_to_standard_form_with_bounds(c, hae if me else None, hbe if me else None, hai, hbl, hbu, lb, ub, x0)
    ha3, hb3 = tools.precondition_constraints(ha2, hb2)
    dae, dai = DeviceMatrix.from_csr(a_eq), DeviceMatrix.from_csr(a_in)
    dae2, dbe, _ = dae.precondition_rows(be)
    dai2, dbl, dbu = dai.precondition_rows(bl, bu)
    _same_csr(dae2, hae)
    _same_csr(dai2, hai)
    assert np.array_equal(dbe, hbe) and np.array_equal(dbl, hbl) and np.array_equal(dbu, hbu)
    da2 = DeviceMatrix.standard_form(dae2 if me else None, dai2)
    _same_csr(da2, ha2)
    da3, db3, _ = da2.precondition_rows(hb2)
    _same_csr(da3, ha3)
    assert np.array_equal(db3, hb3)
    assert np.array_equal(dai2.matvec(x0, 1), hx[n:])       # the slack part of x0: A_ineq' x0 in the scaled block's stored order


@pytest.mark.parametrize("case", ["sc50a", "potts8", "random1"])
def test_lp_admm_device_setup_equals_host_setup(case, monkeypatch):
    """The same iterates bit for bit whether ADMM.py:76-101 runs on the device (default) or in numpy (SLP_HOST_SETUP=1)."""
    from conftest import Recorder, solver_args
    from pysparselp_amd.ADMM import lp_admm

    d = load_golden("lp_" + case)
    runs = []
    for host in ("0", "1"):
        monkeypatch.setenv("SLP_HOST_SETUP", host)
        rec = Recorder()
        x = lp_admm(*solver_args(d), nb_iter=120, callback_func=rec, nb_iter_plot=20)
        runs.append((x, rec))
    assert np.array_equal(runs[0][0], runs[1][0])
    for u, v in zip(runs[0][1].x, runs[1][1].x):
        assert np.array_equal(u, v)
    assert runs[0][1].e1 == runs[1][1].e1 and runs[0][1].veq == runs[1][1].veq
