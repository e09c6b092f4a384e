"""BASELINE config 3 at FULL size (1e6 variables x 2e6 inequality rows, density 1e-3, ~2e9 stored entries) and one
matrix with more than 2^31 stored entries, inside ``pytest -m gpu``.

Parity at this size is established through
  (i)   bit-for-bit agreement of the three kernel families on the full products: value-dictionary strips, fp64 strips
        and the thread-per-row CSR kernel ``k_spmv<1>`` (every one of them sums a row with a single accumulator in
        storage order; ``k_spmv<1>`` is pinned bit-exactly to the oracle at small sizes in test_gpu_parity.py),
  (ii)  the oracle itself on downloaded slices: 2048 rows of A and 2048 rows of the device-built A^T,
  (iii) the adjoint identity <A x, y> = <x, A^T y>,
  (iv)  solver runs: two runs bit-identical, objective after 50 iterations equal to the recorded convergence run
        (profiles/r05_c3_convergence.json) -- a regression check against an earlier GPU run, not a parity claim,
  (v)   WHOLE SOLVERS against the CPU oracle on the full matrix (downloaded once, ~24 GB): Chambolle-Pock iterates bit for
        bit, matrix-free ADMM (the bench's reuse level 4) to 1e-9 per entry and 1e-6 relative in the objective
        (north_star's gate), on the value-dictionary strips and on the general fp64 strips.  The oracle's loops over
        independent rows / columns run multi-threaded there (bit-identical to its one-thread = reference order,
        tests/test_oracle_golden.py::test_oracle_threads_do_not_change_a_bit); needs ~250 GB of host memory (skips below).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M, P, SEED = 1_000_000, 2_000_000, 1e-3, 0


@pytest.fixture(scope="module")
def c3():
    from pysparselp_amd.problems import random_lp_on_device

    a, xf, c, lb, ub, b = random_lp_on_device(N, M, P, seed=SEED)
    yield a, xf, c, lb, ub, b
    a.close()


def _vectors():
    rng = np.random.RandomState(5)
    return rng.randn(N), rng.randn(M)


def test_c3_products_agree_bit_for_bit_across_kernel_families(c3):
    a = c3[0]
    assert 1.98e9 < a.nnz < 2.0e9
    x, y = _vectors()
    got = {}
    for policy, kernels in ((0, (2, 3)), (1, (1,)), (2, (0,))):
        a.set_format(policy)
        assert a.spmv_kernel(False) in kernels and a.spmv_kernel(True) in kernels, (policy, a.spmv_kernel(False), a.spmv_kernel(True))
        got[policy] = (a.matvec(x, order=1), a.rmatvec(y, order=1))  # order 1 = SEQUENTIAL: k_spmv<1> under policy 2
    a.set_format(0)
    for policy in (1, 2):
        assert np.array_equal(got[0][0], got[policy][0]), f"A x: dictionary strips vs policy {policy}"
        assert np.array_equal(got[0][1], got[policy][1]), f"A^T y: dictionary strips vs policy {policy}"
    ax, aty = got[0]
    # (iii) adjoint identity; both sides are sums of 2e9 products of O(1) numbers
    lhs, rhs = float(ax.dot(y)), float(x.dot(aty))
    assert abs(lhs - rhs) <= 1e-10 * (abs(lhs) + np.linalg.norm(ax) * np.linalg.norm(y))
    # (ii) oracle on slices.  Rows: the slice of A itself.
    for r0 in (0, 1_234_567, M - 2048):
        rows = a.download_rows(r0, 2048)
        assert np.array_equal(oracle.matvec(oracle.as_csr(rows), x), ax[r0:r0 + 2048])
    # Columns: rows of the device-built transpose; inside a row of A^T the entries are ordered by increasing row of A,
    # which is the order in which scipy's csc_matvec accumulates `y * A` (oracle: orc_csr_rmatvec).
    for c0 in (0, 543_210, N - 2048):
        cols = a.download_rows(c0, 2048, transposed=True)
        assert np.all(np.diff(cols.indptr) > 0)
        for j in (0, 1000, 2047):
            seg = cols.indices[cols.indptr[j]:cols.indptr[j + 1]]
            assert np.all(np.diff(seg) > 0)  # strictly increasing rows: stable transposition, no duplicates
        assert np.array_equal(oracle.matvec(oracle.as_csr(cols), y), aty[c0:c0 + 2048])
    # the transposed slice holds exactly the entries of A in those columns: compare with regenerated rows of A
    from pysparselp_amd.device import DeviceMatrix

    blk = DeviceMatrix.random(4096, N, P, SEED, 777_000)
    sub = blk.download().tocsc()[:, 543_210:543_210 + 2048].tocoo()
    blk.close()
    cols = a.download_rows(543_210, 2048, transposed=True).tocoo()
    keep = (cols.col >= 777_000) & (cols.col < 777_000 + 4096)
    got_set = set(zip(cols.row[keep].tolist(), (cols.col[keep] - 777_000).tolist(), cols.data[keep].tolist()))
    ref_set = set(zip(sub.col.tolist(), sub.row.tolist(), sub.data.tolist()))
    assert got_set == ref_set and len(ref_set) > 5000


def test_c3_solver_runs_are_deterministic_and_match_the_recorded_convergence(c3):
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.scale import DeviceCP

    a, xf, c, lb, ub, b = c3
    rec = json.load(open(os.path.join(REPO, "profiles", "r05_c3_convergence.json")))
    assert (rec["n"], rec["m"], rec["nnz"]) == (N, M, a.nnz)
    ax = a.matvec(xf)
    assert abs(float(c.dot(xf)) - rec["feasible_point"]["objective"]) < 1e-9
    assert float(np.max(ax - b)) <= 1e-9  # the generator's point is feasible (randomLP.py:43-46)

    def at50(entry):
        e = [r for r in entry if r["iteration"] == 50][0]
        return e["objective"], e["max_row_violation"]

    runs = []
    for _ in range(2):
        s = DeviceCP(a, b, c, lb, ub)
        s.iterate(30)
        x30 = s.x()
        s.iterate(20)
        runs.append((x30, s.x()))
        s.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    obj, viol = at50(rec["chambolle_pock_ppd"])
    x = runs[0][1]
    assert abs(float(c.dot(x)) - obj) <= 1e-9 * abs(obj)
    assert abs(float(np.max(a.matvec(x) - b)) - viol) <= 1e-8 * (1 + viol)

    runs = []
    for _ in range(2):
        s = DeviceADMM(a, b, c, lb, ub)  # value-dictionary strips: the matrix itself is left untouched (deferred row scaling)
        assert s.reuse == 4
        s.iterate(30)
        x30 = s.x(N)
        s.iterate(20)
        runs.append((x30, s.x(N)))
        s.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    obj, viol = at50(rec["admm"])
    x = runs[0][1]
    scale = abs(rec["feasible_point"]["objective"])  # the objective passes through 0 around iteration 50
    assert abs(float(c.dot(x)) - obj) <= 1e-6 * scale
    assert abs(float(np.max(a.matvec(x) - b)) - viol) <= 1e-6 * (1 + viol)


@pytest.fixture(scope="module")
def c4slice():
    """The 1/8 row slice of a 1e7-variable, density-1e-4 LP (BASELINE config 4's per-rank shape at a density that fits):
    2.5e6 x 1e7, ~2.5e9 stored entries (> 2^31: every entry offset is 64-bit)."""
    from pysparselp_amd.device import DeviceMatrix

    a = DeviceMatrix.random(C4S_ROWS, C4S_N, C4S_DENSITY, C4S_SEED, 0)
    yield a
    a.close()


C4S_N, C4S_ROWS, C4S_DENSITY, C4S_SEED = 10_000_000, 2_500_000, 1e-4, 1


def test_more_than_2_31_stored_entries(c4slice):
    """Tall cells (csrc/slp_tall.hip) in both orientations vs the thread-per-row CSR kernel bit for bit, the oracle on row
    slices of both orientations, the adjoint identity."""
    a, n, rows = c4slice, C4S_N, C4S_ROWS
    assert a.nnz > 2 ** 31
    rng = np.random.RandomState(6)
    x, y = rng.randn(n), rng.randn(rows)
    assert a.spmv_kernel(False) == 6 and a.spmv_kernel(True) == 6
    ax, aty = a.matvec(x, order=1), a.rmatvec(y, order=1)
    a.set_format(2)
    assert np.array_equal(a.matvec(x, order=1), ax)
    assert np.array_equal(a.rmatvec(y, order=1), aty)
    lhs, rhs = float(ax.dot(y)), float(x.dot(aty))
    assert abs(lhs - rhs) <= 1e-10 * (abs(lhs) + np.linalg.norm(ax) * np.linalg.norm(y))
    for r0 in (0, rows - 1024):  # the last rows sit behind offset 2^31
        sl = a.download_rows(r0, 1024)
        assert np.array_equal(oracle.matvec(oracle.as_csr(sl), x), ax[r0:r0 + 1024])
    for c0 in (0, n - 4096):
        sl = a.download_rows(c0, 4096, transposed=True)
        assert np.array_equal(oracle.matvec(oracle.as_csr(sl), y), aty[c0:c0 + 4096])
    a.set_format(0)


def test_chunked_c4slice_equals_the_unchunked_one_at_full_size(c4slice):
    """The same 2.5e9-entry slice as a ChunkedDeviceMatrix (4 row chunks whose CSR never coexists, csrc/slp_chunked.hip): products
    and Chambolle-Pock iterates bit for bit the unchunked ones -- which the next test pins against the CPU oracle at this size --
    and the matrix-free ADMM to 1e-12.  BASELINE config 4 on one GPU (bench.py's default workload) is eight such chunks."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    a, n, rows, dens, seed = c4slice, C4S_N, C4S_ROWS, C4S_DENSITY, C4S_SEED
    a.set_format(0)
    xf, c, lb, ub, b = a.random_lp_vectors(dens, seed, 0)
    ch, xf2, c2, lb2, ub2, b2 = random_lp_on_device(n, rows, dens, seed=seed, chunks=4)
    try:
        assert ch.chunks == 4 and ch.nnz == a.nnz and np.array_equal(b2, b) and np.array_equal(c2, c)
        rng = np.random.RandomState(8)
        x, y = rng.randn(n), rng.randn(rows)
        assert np.array_equal(ch.matvec(x), a.matvec(x)) and np.array_equal(ch.rmatvec(y), a.rmatvec(y))
        xs = []
        for mat in (a, ch):
            s = DeviceCP(mat, b, c, lb, ub)
            s.iterate(3)
            xs.append(s.x())
            s.close()
        assert np.array_equal(xs[0], xs[1])
        xs = []
        for mat in (a, ch):
            s = DeviceADMM(mat, b, c, lb, ub)
            s.iterate(3)
            xs.append(s.x(n))
            s.close()
        assert float(np.max(np.abs(xs[0] - xs[1]) / (1 + np.abs(xs[0])))) <= 1e-12
    finally:
        ch.close()


def test_c4slice_whole_solvers_match_the_cpu_oracle_at_full_size(c4slice):
    """The per-rank solvers of config 4 on the FULL slice against the oracle on the downloaded matrix: Chambolle-Pock bit for
    bit, matrix-free ADMM 1e-9 / 1e-6 in the objective.  Skips (visibly) when the host lacks the memory for the oracle's
    copies."""
    import time

    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.scale import DeviceCP

    if _mem_available_gb() < 300:
        pytest.skip("less than 300 GB of host memory available for the oracle's copies of the 2.5e9-entry slice")
    a, n, rows, dens, seed = c4slice, C4S_N, C4S_ROWS, C4S_DENSITY, C4S_SEED
    a.set_format(0)
    xf, c, lb, ub, b = a.random_lp_vectors(dens, seed, 0)
    record = {"n": n, "rows": rows, "stored_entries": a.nnz, "oracle_threads": min(64, os.cpu_count() or 1)}
    oracle.set_threads(record["oracle_threads"])
    try:
        host = oracle.as_csr(a.download())
        stamps = []
        x_cpu, _ = oracle.chambolle_pock_ppd(c, None, None, host, None, b, lb, ub, nb_max_iter=3, nb_iter_plot=10 ** 9,
                                             iterate_hook=lambda *_: stamps.append(time.perf_counter()))
        record["chambolle_pock_oracle_seconds_per_iteration"] = float(np.mean(np.diff(stamps)))
        host._csc = None
        s = DeviceCP(a, b, c, lb, ub)
        s.iterate(3)
        assert np.array_equal(s.x(), x_cpu)
        s.close()
        x_cpu = oracle.lp_admm_cg(c, None, None, host, None, b, lb, ub, nb_iter=2, nb_iter_plot=10 ** 9)
        del host
        s = DeviceADMM(a, b, c, lb, ub)
        assert a.spmv_kernel(False) == 6 and a.spmv_kernel(True) == 6
        s.iterate(3)
        x_gpu = s.x(n)
        s.close()
        err = float(np.max(np.abs(x_gpu - x_cpu) / (1 + np.abs(x_cpu))))
        record["admm_max_scaled_error"] = err
        assert err <= 1e-9, err
        assert abs(float(c.dot(x_gpu)) - float(c.dot(x_cpu))) <= 1e-6 * abs(float(c.dot(x_cpu)))
    finally:
        oracle.set_threads(1)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "c4slice_oracle_parity.json"), "w") as f:
        json.dump(record, f, indent=1)


def test_c3_equality_variant_whole_solvers_match_the_cpu_oracle_at_full_size(c3):
    """SURVEY.md 8(d)'s 10 %-equality variant (randomLP.py:62-68) at config 3's FULL size: the first 2e5 rows are equalities
    ``a_i x = a_i x_f``.  Chambolle-Pock forms ``(c + y_eq * a_eq) + y_ineq * a_ineq`` (ChambollePockPPD.py:206,216) from two
    products over strip copies of ``A_e^T`` and ``A_i^T`` -- row-range copies built for the solver on the ordinary matrix, the
    chunks' own copies on a chunked matrix cut at m_eq, two masked products where neither exists -- x BIT FOR BIT against the
    oracle on the downloaded matrix in all three forms; matrix-free ADMM 1e-9 / 1e-6 in the objective."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    a, xf, c, lb, ub, _ = c3
    if _mem_available_gb() < 250:
        pytest.skip("less than 250 GB of host memory available for the full-size oracle run")
    m_eq = M // 10
    a.set_format(0)
    b = a.random_lp_vectors(P, SEED, m_eq=m_eq)[4]
    cp_iters, admm_iters = 4, 3
    oracle.set_threads(min(64, os.cpu_count() or 1))
    record = {"n": N, "m": M, "equality_rows": m_eq, "stored_entries": a.nnz}
    try:
        host = a.download()
        ae, ai = oracle.as_csr(host[:m_eq]), oracle.as_csr(host[m_eq:])
        del host
        assert np.array_equal(b[:m_eq], oracle.matvec(ae, xf))     # b_eq = A_e x_f in csr_matvec order (randomLP.py:63)
        x_cpu, _ = oracle.chambolle_pock_ppd(c, ae, b[:m_eq], ai, None, b[m_eq:], lb, ub, nb_max_iter=cp_iters, nb_iter_plot=10 ** 9)
        ae._csc = ai._csc = None
        forms = {}
        for name, env in (("row-range copies", None), ("masked products", "masked")):
            if env:
                os.environ["SLP_CP_SPLIT"] = env
            try:
                s = DeviceCP(a, b, c, lb, ub, m_eq=m_eq)
            finally:
                os.environ.pop("SLP_CP_SPLIT", None)
            forms[name] = s.split_form()
            s.iterate(cp_iters)
            x_gpu = s.x()
            s.close()
            assert np.array_equal(x_gpu, x_cpu), (name, float(np.max(np.abs(x_gpu - x_cpu))))
        assert forms == {"row-range copies": 1, "masked products": 2}, forms
        ch, _, _, _, _, b2 = random_lp_on_device(N, M, P, seed=SEED, chunks=2, m_eq=m_eq)   # (two chunks: the equality rows, the others)
        try:
            assert ch.chunks == 2 and np.array_equal(b2, b)
            s = DeviceCP(ch, b, c, lb, ub, m_eq=m_eq)
            assert s.split_form() == 1
            s.iterate(cp_iters)
            assert np.array_equal(s.x(), x_cpu)
            s.close()
        finally:
            ch.close()
        record["chambolle_pock_ppd"] = {"iterations": cp_iters, "x_bit_for_bit": True, "forms": forms, "objective": float(c.dot(x_cpu))}
        x_cpu = oracle.lp_admm_cg(c, ae, b[:m_eq], ai, None, b[m_eq:], lb, ub, nb_iter=admm_iters - 1, nb_iter_plot=10 ** 9)
        del ae, ai
        s = DeviceADMM(a, b, c, lb, ub, m_eq=m_eq)
        s.iterate(admm_iters)
        x_gpu = s.x(N)
        s.close()
        err = float(np.max(np.abs(x_gpu - x_cpu) / (1 + np.abs(x_cpu))))
        record["admm"] = {"iterations": admm_iters, "max_scaled_error": err, "objective": float(c.dot(x_cpu))}
        assert err <= 1e-9, err
        assert abs(float(c.dot(x_gpu)) - float(c.dot(x_cpu))) <= 1e-6 * abs(float(c.dot(x_cpu)))
    finally:
        oracle.set_threads(1)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "c3_eq10_oracle_parity.json"), "w") as f:
        json.dump(record, f, indent=1)


def _mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


def test_c3_whole_solvers_match_the_cpu_oracle_at_full_size(c3):
    """(v) of the module docstring.  Reference: ChambollePockPPD.py:195-343; ADMM.py:143-268 with the use_cg flags
    (:182-201) + conjugateGradientLinearSolver.py:30-52, restated in oracle/oracle.py and pinned by tests/golden."""
    import time

    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    a, xf, c, lb, ub, b = c3
    if _mem_available_gb() < 250:
        pytest.skip("less than 250 GB of host memory available for the full-size oracle run")
    cp_iters, admm_iters = 6, 5
    record = {"n": N, "m": M, "stored_entries": a.nnz, "oracle_threads": min(64, os.cpu_count() or 1)}
    oracle.set_threads(record["oracle_threads"])
    try:
        t0 = time.perf_counter()
        host = oracle.as_csr(a.download())
        record["download_seconds"] = time.perf_counter() - t0
        # ---- Chambolle-Pock: every kernel on this path sums in the reference's order -> bit for bit
        stamps = []
        t0 = time.perf_counter()
        x_cpu, _ = oracle.chambolle_pock_ppd(c, None, None, host, None, b, lb, ub, nb_max_iter=cp_iters, nb_iter_plot=10 ** 9,
                                             iterate_hook=lambda *_: stamps.append(time.perf_counter()))
        record["chambolle_pock_ppd"] = {"iterations": cp_iters, "oracle_setup_seconds": stamps[0] - t0,
                                        "oracle_seconds_per_iteration": float(np.mean(np.diff(stamps))),
                                        "objective_oracle": float(c.dot(x_cpu))}
        host._csc = None  # 24 GB back before the ADMM chain builds its own copies
        for policy in (0, 1):  # value-dictionary strips, fp64 strips
            a.set_format(policy)
            s = DeviceCP(a, b, c, lb, ub)
            s.iterate(cp_iters)
            x_gpu = s.x()
            s.close()
            assert np.array_equal(x_gpu, x_cpu), (policy, float(np.max(np.abs(x_gpu - x_cpu))))
        a.set_format(0)
        record["chambolle_pock_ppd"]["objective_gpu"] = float(c.dot(x_gpu))
        # ---- matrix-free ADMM at the bench's reuse level (4): same mathematics, dot products and recurrences round
        # differently -> 1e-9 per entry, 1e-6 relative in the objective (north_star)
        stamps = []
        t0 = time.perf_counter()
        x_cpu = oracle.lp_admm_cg(c, None, None, host, None, b, lb, ub, nb_iter=admm_iters - 1, nb_iter_plot=10 ** 9,
                                  iterate_hook=lambda *_: stamps.append(time.perf_counter()))
        record["admm"] = {"iterations": admm_iters, "oracle_setup_seconds": stamps[0] - t0,
                          "oracle_seconds_per_iteration": float(np.mean(np.diff(stamps))), "objective_oracle": float(c.dot(x_cpu))}
        del host
        obj = float(c.dot(x_cpu))
        for policy in (0, 1):
            # policy 1 scales the rows in place (no dictionary): a matrix of its own
            mat = a if policy == 0 else random_lp_on_device(N, M, P, seed=SEED)[0]
            mat.set_format(policy)
            s = DeviceADMM(mat, b, c, lb, ub)
            assert s.reuse == 4
            s.iterate(admm_iters)
            x_gpu = s.x(N)
            s.close()
            if policy == 1:
                mat.close()
            err = float(np.max(np.abs(x_gpu - x_cpu) / (1 + np.abs(x_cpu))))
            record["admm"][f"max_scaled_error_policy{policy}"] = err
            assert err <= 1e-9, (policy, err)
            assert abs(float(c.dot(x_gpu)) - obj) <= 1e-6 * abs(obj), (policy, float(c.dot(x_gpu)), obj)
        record["admm"]["objective_gpu"] = float(c.dot(x_gpu))
    finally:
        oracle.set_threads(1)
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "c3_full_oracle_parity.json"), "w") as f:
        json.dump(record, f, indent=1)
