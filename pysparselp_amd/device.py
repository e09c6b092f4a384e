"""Device-resident sparse matrix handle (``slp_matrix`` of include/slp_hip.h)."""
import numpy as np

import scipy.sparse

from . import _lib
from ._lib import ORDER_AUTO


class DeviceMatrix:
    """CSR matrix living in HBM, in both orientations (the transposed copy is
    built on the device the first time ``A^T y`` is needed)."""

    def __init__(self, handle, shape):
        self._l = _lib.lib()
        self._h = handle
        self.shape = (int(shape[0]), int(shape[1]))

    @classmethod
    def from_csr(cls, a):
        l = _lib.lib()
        indptr, indices, data = _lib.csr_arrays(a)
        h = _lib.check_handle(l.slp_matrix_create(a.shape[0], a.shape[1], _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(data)))
        return cls(h, a.shape)

    @classmethod
    def random(cls, nrow, ncol, density, seed, row_offset=0):
        """Rows ``row_offset .. row_offset+nrow`` of the synthetic benchmark matrix
        (distribution of the reference's randomLP.rand_sparse, generated on the GPU)."""
        l = _lib.lib()
        h = _lib.check_handle(l.slp_matrix_random(int(nrow), int(ncol), float(density), int(seed), int(row_offset)))
        return cls(h, (nrow, ncol))

    def gather_rows(self, rows, scale=None):
        """New DeviceMatrix whose row r is ``scale[r] *`` row ``rows[r]`` of this one, built on the device."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        scale = np.ones(rows.size) if scale is None else _lib.f64(np.broadcast_to(scale, rows.shape))
        h = _lib.check_handle(self._l.slp_matrix_gather_rows(self._h, rows.size, _lib.ptr(rows), _lib.ptr(scale)))
        return DeviceMatrix(h, (rows.size, self.shape[1]))

    def close(self):
        if getattr(self, "_h", None):
            self._l.slp_matrix_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def nnz(self):
        return int(self._l.slp_matrix_nnz(self._h))

    def matvec(self, x, order=ORDER_AUTO):
        x = _lib.f64(x)
        assert x.size == self.shape[1]
        y = np.empty(self.shape[0])
        _lib.check(self._l.slp_matrix_spmv(self._h, _lib.ptr(x), _lib.ptr(y), int(order)))
        return y

    def rmatvec(self, y, order=ORDER_AUTO):
        y = _lib.f64(y)
        assert y.size == self.shape[0]
        out = np.empty(self.shape[1])
        _lib.check(self._l.slp_matrix_spmv_t(self._h, _lib.ptr(y), _lib.ptr(out), int(order)))
        return out

    def abs_pow_matvec(self, x, p, transposed=False):
        """``(|A| ** p) x`` or, ``transposed``, ``(|A| ** p)^T x`` on the strip / tall-cell copy of that orientation (the sums
        behind Chambolle-Pock's preconditioners, ChambollePockPPD.py:122-179)."""
        x = _lib.f64(x)
        assert x.size == self.shape[0 if transposed else 1]
        out = np.empty(self.shape[1 if transposed else 0])
        _lib.check(self._l.slp_matrix_spmv_abs_pow(self._h, int(bool(transposed)), float(p), _lib.ptr(x), _lib.ptr(out)))
        return out

    def download(self, transposed=False):
        """scipy CSR copy of the device arrays (``transposed=True``: the device-built A^T)."""
        nrow, ncol = (self.shape[1], self.shape[0]) if transposed else self.shape
        indptr = np.empty(nrow + 1, dtype=np.int64)
        _lib.check(self._l.slp_matrix_download(self._h, int(transposed), _lib.ptr(indptr), None, None))
        nnz = int(indptr[-1])
        indices = np.empty(nnz, dtype=np.int32)
        data = np.empty(nnz)
        _lib.check(self._l.slp_matrix_download(self._h, int(transposed), None, _lib.ptr(indices), _lib.ptr(data)))
        return scipy.sparse.csr_matrix((data, indices, indptr), shape=(nrow, ncol))

    def download_rows(self, row0, count, transposed=False):
        """scipy CSR copy of rows ``row0 .. row0 + count`` (of the device-built A^T when ``transposed``)."""
        ncol = self.shape[0] if transposed else self.shape[1]
        indptr = np.empty(count + 1, dtype=np.int64)
        _lib.check(self._l.slp_matrix_download_rows(self._h, int(transposed), int(row0), int(count), _lib.ptr(indptr), None, None))
        nnz = int(indptr[-1] - indptr[0])
        indices = np.empty(nnz, dtype=np.int32)
        data = np.empty(nnz)
        _lib.check(self._l.slp_matrix_download_rows(self._h, int(transposed), int(row0), int(count), None, _lib.ptr(indices), _lib.ptr(data)))
        return scipy.sparse.csr_matrix((data, indices, indptr - indptr[0]), shape=(count, ncol))

    def normal_matrix(self, gamma_eq, gamma_ineq):
        """``gamma_eq A^T A + gamma_ineq I`` as a new DeviceMatrix (ADMM.py:93-101), formed on the device in scipy's
        accumulation order (bit-identical values, sorted rows, exact zeros dropped)."""
        h = _lib.check_handle(self._l.slp_matrix_normal(self._h, float(gamma_eq), float(gamma_ineq)))
        return DeviceMatrix(h, (self.shape[1], self.shape[1]))

    def precondition_rows(self, b=None, b2=None):
        """``precondition_constraints(a, b, b2, alpha=2)`` of the reference (tools.py:272-290) on the device:
        ``(A_scaled, b_scaled, b2_scaled)`` -- rows of unit 2-norm stored in scipy's order for ``diags(1/s) * A``."""
        b = None if b is None else _lib.f64(b).copy()
        b2 = None if b2 is None else _lib.f64(b2).copy()
        h = _lib.check_handle(self._l.slp_matrix_precondition_rows(self._h, _lib.ptr(b), _lib.ptr(b2)))
        return DeviceMatrix(h, self.shape), b, b2

    @staticmethod
    def standard_form(a_eq, a_ineq):
        """``[[A_eq, 0], [A_ineq, -I]]`` (tools.py:88-127) from two DeviceMatrix blocks (``a_eq`` may be None)."""
        l = _lib.lib()
        h = _lib.check_handle(l.slp_matrix_standard_form(None if a_eq is None else a_eq._h, a_ineq._h))
        me = 0 if a_eq is None else a_eq.shape[0]
        return DeviceMatrix(h, (me + a_ineq.shape[0], a_ineq.shape[1] + a_ineq.shape[0]))

    def remove_columns(self, keep, shift=None):
        """``(A[:, keep], A @ shift)``: the column compaction of ``SparseLP.remove_fixed_variables``
        (SparseLP.py:632-674) on the device; ``keep`` is a boolean mask, entries stay in storage order.
        ``A @ shift`` (csr_matvec order, unreduced matrix) is ``None`` without ``shift``."""
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        assert keep.size == self.shape[1]
        a_shift = None
        if shift is not None:
            shift = _lib.f64(shift)
            assert shift.size == self.shape[1]
            a_shift = np.empty(self.shape[0])
        h = _lib.check_handle(self._l.slp_matrix_remove_columns(self._h, _lib.ptr(keep), _lib.ptr(shift), _lib.ptr(a_shift)))
        return DeviceMatrix(h, (self.shape[0], int(keep.sum()))), a_shift

    def release_csr(self):
        """Keep only the strip copies (both orientations must run on them): products and solver iterations keep working,
        everything that needs the CSR entries raises."""
        _lib.check(self._l.slp_matrix_release_csr(self._h))

    def set_format(self, policy):
        """0: best available copy; 1: no value dictionary (fp64 strip entries); 2: CSR kernels only."""
        _lib.check(self._l.slp_matrix_set_format(self._h, int(policy)))

    def spmv_kernel(self, transposed=False):
        return int(self._l.slp_matrix_spmv_kernel(self._h, int(transposed)))

    def bench_spmv(self, transposed=False, order=ORDER_AUTO, reps=20):
        """Average GPU milliseconds of one SpMV launch on resident vectors (HIP events)."""
        ms = np.zeros(1)
        _lib.check(self._l.slp_matrix_bench_spmv(self._h, int(transposed), int(order), int(reps), _lib.ptr(ms)))
        return float(ms[0])

    def random_lp_vectors(self, density, seed, row_offset=0, columns=True, m_eq=0):
        """``(feasible_x, c, lb, ub, b_upper)`` of the synthetic LP whose rows this matrix holds
        (``columns=False``: only ``b_upper``, the others ``None`` -- a row chunk of a chunked matrix).
        ``m_eq``: the first ``m_eq`` rows of this matrix are equalities, ``b[:m_eq] = A_e feasible_x`` (randomLP.py:62-68)."""
        n, m = self.shape[1], self.shape[0]
        xf, c, lb, ub = (np.empty(n), np.empty(n), np.empty(n), np.empty(n)) if columns else (None, None, None, None)
        b = np.empty(m)
        _lib.check(self._l.slp_random_lp_vectors_eq(self._h, float(density), int(seed), int(row_offset), int(m_eq), _lib.ptr(xf),
                                                    _lib.ptr(c), _lib.ptr(lb), _lib.ptr(ub), _lib.ptr(b)))
        return xf, c, lb, ub, b


class ChunkedDeviceMatrix(DeviceMatrix):
    """A constraint matrix assembled from row chunks whose CSR never coexists (``slp_matrix_chunked_*`` of
    include/slp_hip.h): every appended chunk is converted into its product copies for both orientations and its CSR is
    released, so an LP larger than one CSR copy of itself fits one GPU.  Products and the at-scale solvers
    (``DeviceCP``, ``DeviceADMM``) take it like a ``DeviceMatrix``; ``A^T y`` continues the column sums from chunk to
    chunk, bit for bit the unchunked product."""

    def __init__(self, ncol, expect_chunks=0, expect_rows=0):
        """``expect_chunks``: how many chunks will be appended (optional; lets the library take the next chunk's buffers from
        the driver beside the work on it -- ``slp_matrix_chunked_expect``).  ``expect_rows``: the row count of the whole matrix
        (optional): every chunk then brings its exact share of a multiple of the CU count of tall row blocks, so that chunks of
        unequal sizes (equality rows cut off into chunks of their own) fill the one grid of a product as equal chunks do."""
        l = _lib.lib()
        super().__init__(_lib.check_handle(l.slp_matrix_chunked_create(int(ncol))), (0, ncol))
        if expect_rows:
            _lib.check(l.slp_matrix_chunked_expect_rows(self._h, int(expect_chunks), int(expect_rows)))
        elif expect_chunks:
            _lib.check(l.slp_matrix_chunked_expect(self._h, int(expect_chunks)))

    def append(self, chunk):
        """Takes ownership of ``chunk`` (a ``DeviceMatrix`` with its CSR); every chunk but the last needs an even row count."""
        _lib.check(self._l.slp_matrix_chunked_append(self._h, chunk._h))
        chunk._h = None  # the chunked matrix owns it now
        self.shape = (self.shape[0] + chunk.shape[0], self.shape[1])
        return self

    @classmethod
    def from_csr(cls, a, chunk_entries=2_500_000_000, cut_at=0):
        """A host scipy CSR matrix uploaded in row chunks of about ``chunk_entries`` stored entries (even inner cuts): the way a
        host LP whose CSR does not fit the device twice over gets resident.  Every chunk must qualify for strip copies.
        ``cut_at`` (even): a row that must be a chunk boundary -- the number of equality rows in front, so that Chambolle-Pock
        finds the equality and the inequality rows in chunks of their own (``DeviceCP``: the reference's
        ``(c + y_eq * a_eq) + y_ineq * a_ineq``, ChambollePockPPD.py:206,216, as two products over the chunks' copies)."""
        cuts = cls.balanced_cuts(a.indptr, chunk_entries)
        if 0 < cut_at < a.shape[0]:
            assert cut_at % 2 == 0, "every chunk but the last needs an even row count"
            indptr = np.asarray(a.indptr, dtype=np.int64)
            head = cls.balanced_cuts(indptr[:cut_at + 1], chunk_entries)
            tail = cls.balanced_cuts(indptr[cut_at:] - indptr[cut_at], chunk_entries)
            cuts = head + [cut_at + c for c in tail[1:]]
        g = cls(a.shape[1], expect_chunks=len(cuts) - 1, expect_rows=a.shape[0])
        for r0, r1 in zip(cuts, cuts[1:]):
            g.append(DeviceMatrix.from_csr(a[r0:r1]))
        return g

    @staticmethod
    def balanced_cuts(indptr, chunk_entries):
        """Row boundaries of k = ceil(nnz / chunk_entries) chunks holding nnz / k stored entries each (even inner cuts).
        Cutting greedily at ``chunk_entries`` would leave whatever remains for the last chunk -- for an entry count slightly
        above a multiple of ``chunk_entries`` a sliver that does not qualify for strip copies, refused by the LAST append after
        every other chunk had been uploaded and converted."""
        indptr = np.asarray(indptr, dtype=np.int64)
        m, nnz = indptr.size - 1, int(indptr[-1])
        k = max(1, -(-nnz // int(chunk_entries)))
        cuts = [0]
        for i in range(1, k):
            r = int(np.searchsorted(indptr, nnz * i // k, side="left")) & ~1
            if cuts[-1] < r < m:
                cuts.append(r)
        cuts.append(m)
        return cuts

    @property
    def chunks(self):
        return int(self._l.slp_matrix_chunks(self._h))

    @staticmethod
    def cuts(rows, chunks, cut_at=0):
        """Row boundaries of ``chunks`` nearly equal chunks of ``rows`` rows, every inner boundary even.  ``cut_at`` (even,
        inside the rows): a row that must be a boundary -- the equality rows in front of it and the inequality rows behind it
        are then chunked on their own, in proportion (at least one chunk each, so ``chunks`` is raised to 2 if need be)."""
        if 0 < cut_at < rows:
            assert cut_at % 2 == 0, "every chunk but the last needs an even row count"
            chunks = max(2, chunks)

            # the chunks in front of the cut and behind it in proportion to the rows (the library gives every chunk its exact share of
            # the row blocks, slp_matrix_chunked_expect_rows: unequal chunks cost nothing in the products)
            head = min(chunks - 1, max(1, int(round(chunks * cut_at / rows))))
            return ChunkedDeviceMatrix.cuts(cut_at, head) + [cut_at + c for c in ChunkedDeviceMatrix.cuts(rows - cut_at, chunks - head)[1:]]
        cuts = [(rows * k // chunks) & ~1 for k in range(chunks)] + [rows]
        return [c for i, c in enumerate(cuts) if i == 0 or c > cuts[i - 1]]
