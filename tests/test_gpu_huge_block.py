"""A device block with more than 2^32 stored entries (its own module: it needs most of the device memory, no other
fixture may be alive).  -m gpu."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def test_more_than_2_32_stored_entries():
    """A device block with more than 2^32 stored entries: 5e6 x 1e7 at density 1e-4 (4.98e9 entries, 60 GB of CSR per
    orientation) -- the share of ONE of FOUR ranks of a 1e7 x 2e7 LP, which the 32-bit sort positions of the transposition
    used to rule out.  The transposition now moves whole (column, row, value) records; the products run on
    tall cells.  Tall cells vs the thread-per-row CSR kernel bit for bit in both orientations, the oracle on row slices of
    both orientations (the last ones sit behind offset 2^32), the adjoint identity."""
    from pysparselp_amd import _lib
    from pysparselp_amd.device import DeviceMatrix

    n, rows, dens, seed = 10_000_000, 5_000_000, 1e-4, 2
    lib = _lib.lib()
    free, total = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
    _lib.check(lib.slp_trim())
    _lib.check(lib.slp_device_memory(_lib.ptr(free), _lib.ptr(total)))
    if free[0] < 250e9:
        pytest.skip("less than 250 GB of device memory free")
    a = DeviceMatrix.random(rows, n, dens, seed, 0)
    try:
        assert a.nnz > 2 ** 32
        rng = np.random.RandomState(7)
        x, y = rng.randn(n), rng.randn(rows)
        a.set_format(2)                      # CSR kernels: builds the transposed copy (the > 2^32 sort)
        ax, aty = a.matvec(x, order=1), a.rmatvec(y, order=1)
        for r0 in (0, rows - 1024):
            sl = a.download_rows(r0, 1024)
            assert np.array_equal(oracle.matvec(oracle.as_csr(sl), x), ax[r0:r0 + 1024])
        for c0 in (0, n - 4096):
            sl = a.download_rows(c0, 4096, transposed=True)
            assert np.all(np.diff(sl.indptr) > 0)
            assert np.array_equal(oracle.matvec(oracle.as_csr(sl), y), aty[c0:c0 + 4096])
        lhs, rhs = float(ax.dot(y)), float(x.dot(aty))
        assert abs(lhs - rhs) <= 1e-10 * (abs(lhs) + np.linalg.norm(ax) * np.linalg.norm(y))
        _lib.check(lib.slp_trim())
        a.set_format(0)
        assert a.spmv_kernel(False) == 6 and a.spmv_kernel(True) == 6
        assert np.array_equal(a.matvec(x), ax)
        assert np.array_equal(a.rmatvec(y), aty)
    finally:
        a.close()
        _lib.check(lib.slp_trim())
