"""Tall-cell SpMV (csrc/slp_tall.hip): rows that are long over a width far beyond the caches but sparse inside every
LDS-sized window -- the per-rank slice of a 1e7-variable LP.  Reference products: ``a * x`` / ``y * a``
(ChambollePockPPD.py:206,216,235,240; scipy csr_matvec / csc_matvec), restated in oracle/slp_oracle.c.

Bar: bit for bit against the oracle (every row is one sequential chain in storage order) for any number of row blocks,
incl. rows with more entries in a cell than a packet takes (continuation packets), empty rows / cells / strips, a ragged
last strip, an odd column count, a single row block; solver iterates through it (Chambolle-Pock bit for bit, ADMM 1e-9).
The full-size slice (2.5e6 x 1e7, 2.5e9 entries) is in tests/test_gpu_c3_full.py.  -m gpu."""
import os

import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_matrices_take_the_tall_format():
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    yield
    del os.environ["SLP_STRIP_MIN_NNZ"]
    os.environ.pop("SLP_TALL_R", None)


def _check(a_host, rows_per_block=None, transposed_too=True, policies=((0, 6), (1, 7))):
    from pysparselp_amd.device import DeviceMatrix

    if rows_per_block:
        os.environ["SLP_TALL_R"] = str(rows_per_block)
    else:
        os.environ.pop("SLP_TALL_R", None)
    a = DeviceMatrix.from_csr(a_host)
    try:
        rng = np.random.RandomState(3)
        x, y = rng.randn(a_host.shape[1]), rng.randn(a_host.shape[0])
        ax, aty = oracle.matvec(oracle.as_csr(a_host), x), oracle.rmatvec(oracle.as_csr(a_host), y)
        kern = None
        for policy, want in policies:   # value-dictionary items (5 bytes) / fp64 entries (4 + 8 bytes)
            a.set_format(policy)
            assert a.spmv_kernel(False) == want, (policy, a.spmv_kernel(False))
            assert np.array_equal(a.matvec(x), ax), policy
            if transposed_too:
                k = a.spmv_kernel(True)
                assert np.array_equal(a.rmatvec(y), aty), (policy, k)
                kern = k if policy == 0 else kern
        return kern
    finally:
        a.close()


def _random(m, n, density, seed, decimals=2):
    """Uniformly placed entries with rounded values (few distinct values: a value dictionary exists)."""
    rng = np.random.RandomState(seed)
    k = int(round(density * m * n))
    a = scipy.sparse.coo_matrix((np.ones(k), (rng.randint(0, m, size=k), rng.randint(0, n, size=k))), shape=(m, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    a.data = np.round(rng.randn(a.nnz), decimals)
    a.data[a.data == 0] = 0.5
    return a


@pytest.mark.parametrize("rows_per_block", [None, 1024, 1500, 9984])
def test_random_sparse_wide_matrix_matches_the_oracle_bit_for_bit(rows_per_block):
    a = _random(20000, 50001, 1e-4, 1)            # 5 entries per row over 13 strips (the last one ragged, odd width)
    kern_t = _check(a, rows_per_block)
    assert kern_t in (0, 6)                        # the transpose (2 entries per row) may stay on the CSR kernel


def test_transpose_of_a_wide_slice_runs_on_tall_cells_too():
    a = _random(60000, 9000, 3e-4, 2)              # rows: 2.7 entries over 3 strips; columns: 18 entries over 15 strips
    assert _check(a, 2048) == 6


def test_rows_longer_than_a_packet_takes_continue_in_later_packets():
    rng = np.random.RandomState(4)
    base = _random(6000, 40000, 1e-4, 5).tocoo()
    rows, cols, vals = [base.row], [base.col], [base.data]
    for r, (c0, k) in ((7, (100, 40)), (8, (4096 * 3 + 5, 7)), (4000, (4096 * 9 - 13, 30)), (5999, (0, 13))):
        cc = c0 + np.sort(rng.choice(200, size=k, replace=False))   # k entries inside one or two strips
        keep = ~((rows[0] == r) & np.isin(cols[0], cc))
        rows[0], cols[0], vals[0] = rows[0][keep], cols[0][keep], vals[0][keep]
        rows.append(np.full(k, r)); cols.append(cc); vals.append(np.where(np.round(rng.randn(k), 1) == 0, 0.3, np.round(rng.randn(k), 1)))
    a = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=base.shape).tocsr()
    a.sort_indices()
    assert np.diff(a.indptr).max() >= 40
    _check(a, 1024)
    _check(a, None)


def test_empty_rows_cells_and_strips():
    a = _random(5000, 70000, 5e-5, 6).tocoo()
    drop = ((a.row >= 100) & (a.row < 400))                                  # empty rows
    drop |= (a.col >= 4096 * 2) & (a.col < 4096 * 5)                         # three empty strips: every row block skips those cells
    drop |= (a.row >= 1024) & (a.row < 2048) & (a.col >= 4096 * 7)           # a row block whose last cells are empty
    a = scipy.sparse.coo_matrix((a.data[~drop], (a.row[~drop], a.col[~drop])), shape=a.shape).tocsr()
    a.sort_indices()
    _check(a, 1024)


def test_one_row_block_and_a_single_strip_of_work():
    a = _random(900, 30000, 2e-4, 7)
    _check(a, None, transposed_too=False)           # nrow < 1024: one block of 900 rows
    b = scipy.sparse.hstack([_random(3000, 3000, 2e-4, 8), scipy.sparse.csr_matrix((3000, 40000))]).tocsr()
    b.sort_indices()
    _check(b, 1024, transposed_too=False)           # all entries inside the first strip


def test_solvers_iterate_on_tall_cells_like_the_oracle():
    """Chambolle-Pock bit for bit, matrix-free ADMM (reuse 4, deferred row scaling) to 1e-9, on a reduced-row slice of
    the 1e7-variable shape: 12 000 rows x 400 000 columns at density 1e-4 (40 entries per row, 0.41 per (row, strip));
    then the same with the value dictionary ruled out: fp64 entries, rows scaled in place by the ADMM setup."""
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    n, m, dens = 400_000, 12_000, 1e-4
    os.environ["SLP_TALL_R"] = "1536"
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, dens, seed=3)
    try:
        assert a.spmv_kernel(False) == 6 and a.spmv_kernel(True) == 6
        host = oracle.as_csr(a.download())
        x_cpu, _ = oracle.chambolle_pock_ppd(c, None, None, host, None, b, lb, ub, nb_max_iter=25, nb_iter_plot=10 ** 9)
        s = DeviceCP(a, b, c, lb, ub)
        s.iterate(25)
        assert np.array_equal(s.x(), x_cpu)
        s.close()
        x_cpu = oracle.lp_admm_cg(c, None, None, host, None, b, lb, ub, nb_iter=69, nb_iter_plot=10 ** 9)
        s = DeviceADMM(a, b, c, lb, ub)
        s.iterate(70)                                 # across the level-4 refresh at 64
        x_gpu = s.x(n)
        s.close()
        assert float(np.max(np.abs(x_gpu - x_cpu) / (1 + np.abs(x_cpu)))) <= 1e-9
        a.set_format(1)
        s = DeviceADMM(a, b, c, lb, ub)               # no dictionary: the rows of `a` are normalised in place
        assert a.spmv_kernel(False) == 7 and a.spmv_kernel(True) == 7
        s.iterate(70)
        x_gpu = s.x(n)
        s.close()
        assert float(np.max(np.abs(x_gpu - x_cpu) / (1 + np.abs(x_cpu)))) <= 1e-9
    finally:
        a.close()


def test_full_width_of_the_1e7_variable_shape_with_few_rows():
    """All 1e7 columns (2442 strips, an 80 MB x) with 24 000 rows at density 1e-4 (2.4e7 entries; row blocks of 1024 rows
    -> 24 workgroups): both orientations bit for bit against the oracle, Chambolle-Pock iterates bit for bit."""
    from pysparselp_amd.problems import random_lp_on_device
    from pysparselp_amd.scale import DeviceCP

    n, m, dens = 10_000_000, 24_000, 1e-4
    os.environ["SLP_TALL_R"] = "1024"
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, dens, seed=9)
    try:
        assert a.spmv_kernel(False) == 6
        host = oracle.as_csr(a.download())
        rng = np.random.RandomState(1)
        x, y = rng.randn(n), rng.randn(m)
        assert np.array_equal(a.matvec(x), oracle.matvec(host, x))
        assert np.array_equal(a.rmatvec(y), oracle.rmatvec(host, y))
        x_cpu, _ = oracle.chambolle_pock_ppd(c, None, None, host, None, b, lb, ub, nb_max_iter=8, nb_iter_plot=10 ** 9)
        s = DeviceCP(a, b, c, lb, ub)
        s.iterate(8)
        assert np.array_equal(s.x(), x_cpu)
        s.close()
    finally:
        a.close()


def test_strip_range_split_for_short_very_wide_blocks():
    """SLP_TALL_SPLIT (opt-in): few, tall row blocks whose strips are shared by S workgroups, partial row sums added in
    range order -- deterministic, equal to the oracle to the rounding of the re-association (1e-13 of the sum of
    magnitudes), NOT bit for bit; without the switch the same matrix is bit-exact (the default)."""
    from pysparselp_amd.device import DeviceMatrix

    a_host = _random(3000, 600001, 2e-5, 12)         # 12 entries per row over 147 strips
    x = np.random.RandomState(2).randn(a_host.shape[1])
    ref = oracle.matvec(oracle.as_csr(a_host), x)
    mag = abs(a_host).dot(np.abs(x))
    got = {}
    for split in ("0", "2", "-1"):
        os.environ["SLP_TALL_SPLIT"] = split
        try:
            a = DeviceMatrix.from_csr(a_host)
            assert a.spmv_kernel(False) == 6
            got[split] = (a.matvec(x), a.matvec(x))
            a.close()
        finally:
            del os.environ["SLP_TALL_SPLIT"]
        assert np.array_equal(got[split][0], got[split][1])          # run-to-run deterministic
    assert np.array_equal(got["0"][0], ref)
    for split in ("2", "-1"):
        assert np.max(np.abs(got[split][0] - ref) / (1e-300 + mag)) < 1e-13
    assert not np.array_equal(got["2"][0], ref) or not np.array_equal(got["-1"][0], ref)  # (the split really happened)


def test_single_pass_build_repeats_with_the_exact_sizes_when_its_room_was_short(monkeypatch):
    """The copy is written in ONE pass over the cells (ticket order, decoupled look-back over the cells' sizes) into room sized
    by an estimate; a cell that would not fit writes nothing and the pass is repeated with the sizes it has found.  Forced here
    by halving the room (SLP_TALL_BUILD_ROOM): same copy -- byte count, products against the oracle bit for bit -- as the
    build whose first attempt fits, with both item forms, for a single and for many row blocks, and with the strip-range split."""
    from pysparselp_amd import _lib
    from pysparselp_amd.device import DeviceMatrix

    a_host = _random(40_000, 300_000, 2e-4, seed=21)
    rng = np.random.RandomState(4)
    x, y = rng.randn(a_host.shape[1]), rng.randn(a_host.shape[0])
    ax, aty = oracle.matvec(oracle.as_csr(a_host), x), oracle.rmatvec(oracle.as_csr(a_host), y)
    for r in (None, 1024):
        if r:
            monkeypatch.setenv("SLP_TALL_R", str(r))
        sizes = []
        for room in (None, "0.5", "0.01"):
            if room:
                monkeypatch.setenv("SLP_TALL_BUILD_ROOM", room)
            else:
                monkeypatch.delenv("SLP_TALL_BUILD_ROOM", raising=False)
            a = DeviceMatrix.from_csr(a_host)
            try:
                for policy, want in ((0, 6), (1, 7)):
                    a.set_format(policy)
                    assert a.spmv_kernel(False) == want and a.spmv_kernel(True) == want
                    assert np.array_equal(a.matvec(x), ax) and np.array_equal(a.rmatvec(y), aty), (r, room, policy)
                    sizes.append((int(_lib.lib().slp_matrix_format_bytes(a._h, 0)), int(_lib.lib().slp_matrix_format_bytes(a._h, 1))))
            finally:
                a.close()
        assert sizes[0:2] == sizes[2:4] == sizes[4:6], sizes
    monkeypatch.delenv("SLP_TALL_BUILD_ROOM", raising=False)


def test_b_upper_of_the_generated_lp_through_the_product_copies():
    """``random_lp_on_device(chunks=...)`` takes ``b_upper = ceil((A x_f + ...) 1000) / 1000`` from the chunked matrix's own
    product once all chunks stand (the CSR kernel it used per chunk gathered every x from L2: 11 x the CSR's bytes in HBM traffic at
    1e7 columns): the same vector as from the CSR of the unchunked matrix, and x_f is feasible for it."""
    from pysparselp_amd.problems import random_lp_on_device

    n, m, dens, seed = 200_000, 60_000, 5e-4, 6
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, dens, seed=seed)
    ch, xf2, c2, lb2, ub2, b2 = random_lp_on_device(n, m, dens, seed=seed, chunks=3)
    try:
        assert ch.chunks == 3 and ch.spmv_kernel(False) == 6
        assert np.array_equal(xf, xf2) and np.array_equal(c, c2) and np.array_equal(lb, lb2) and np.array_equal(ub, ub2)
        assert np.array_equal(b, b2)
        # (the ceiling is taken of (A x_f) * 1000 as rounded: b_upper may lie an ulp below A x_f, as in the reference)
        assert np.all(ch.matvec(xf) <= b2 + 1e-12) and np.all(lb <= xf) and np.all(xf <= ub)
        ax = oracle.matvec(oracle.as_csr(a.download()), xf)
        assert np.all(ax <= b2 + 1e-12) and np.all(b2 - ax < 1e-3 + 12.0)   # (|rand_sparse| adds to a few rows)
        assert np.array_equal(ch.matvec(xf), ax)
    finally:
        a.close()
        ch.close()


@pytest.mark.parametrize("width", [2048, 1024])
def test_narrower_strips_for_denser_matrices(width, monkeypatch):
    """The build halves the strips (4096 -> 2048 -> 1024 columns, the x-tile with them) until a cell holds ~0.3-0.6 entries per
    row; items keep their 12-bit column field.  Forced widths on the shapes of the tests above (ragged last strip, odd width,
    continuation packets, empty cells, a transposed copy), then densities at which the build picks them itself: bit for bit."""
    monkeypatch.setenv("SLP_TALL_C", str(width))
    a = _random(20000, 50001, 1e-4, 1)
    _check(a, None)
    _check(a, 1500)
    assert _check(_random(60000, 9000, 3e-4, 2), 2048) == 6
    b = _random(5000, 70000, 5e-5, 6).tocoo()
    drop = ((b.row >= 100) & (b.row < 400)) | ((b.col >= width * 2) & (b.col < width * 5)) | ((b.row >= 1024) & (b.row < 2048) & (b.col >= width * 7))
    b = scipy.sparse.coo_matrix((b.data[~drop], (b.row[~drop], b.col[~drop])), shape=b.shape).tocsr()
    b.sort_indices()
    _check(b, 1024)
    monkeypatch.delenv("SLP_TALL_C")
    dens = {2048: 2e-4, 1024: 4.5e-4}[width]      # 0.82 / 1.84 entries per row and 4096 columns
    _check(_random(30000, 120001, dens, 9), None, policies=((0, 6),))    # (fp64 entries go to the LDS strips at these densities)


def test_strip_width_follows_the_cost_model():
    """``slp_matrix_strip_width``: the width ``tall_build`` chose per copy.  The metric's density (0.41 entries per row and 4096
    columns) stays on 4096-column strips; twice the density takes 2048 (the row blocks keep their height), 4.5 times takes 1024;
    ``SLP_TALL_C`` overrides (profiles/r05_tall_density_sweep.log: the chosen width was the fastest of the three in all cases)."""
    from pysparselp_amd import _lib
    from pysparselp_amd.device import DeviceMatrix

    lib = _lib.lib()
    for dens, cols, want in ((1e-4, 400_000, 4096), (2e-4, 200_000, 2048), (4.5e-4, 100_000, 1024)):
        a = DeviceMatrix.random(2_500_000, cols, dens, 3, 0)    # 1e8 entries; row blocks of 9766 rows where the width allows
        try:
            assert a.spmv_kernel(False) == 6
            assert int(lib.slp_matrix_strip_width(a._h, 0)) == want, (dens, int(lib.slp_matrix_strip_width(a._h, 0)))
        finally:
            a.close()
    os.environ["SLP_TALL_C"] = "1024"
    try:
        a = DeviceMatrix.random(2_500_000, 400_000, 1e-4, 3, 0)
        assert int(lib.slp_matrix_strip_width(a._h, 0)) == 1024
        a.close()
    finally:
        del os.environ["SLP_TALL_C"]

