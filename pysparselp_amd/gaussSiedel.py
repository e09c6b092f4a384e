"""Projected Gauss-Seidel on the GPU.

Drop-in for the reference's only native hot-path module,
``pysparselp/gaussSiedel.pyx`` (``boundedGaussSeidelClass``, :83-153): same
constructor, same ``solve(b, lower_bounds, upper_bounds, x, maxiter, w, order)``
that updates ``x`` in place and returns it.  The sweep keeps the sequential
data dependence of the reference (level schedule, see slp_admm.hip), so the
result is bit-identical to the Cython loop.
"""
import numpy as np

import scipy.sparse

from . import _lib


class boundedGaussSeidelClass:  # noqa: N801  (name kept from the reference)
    def __init__(self, A):  # noqa: N803
        assert scipy.sparse.isspmatrix_csr(A)
        assert A.dtype == np.float64
        self.A = A
        self._l = _lib.lib()
        indptr, indices, data = _lib.csr_arrays(A)
        assert A.shape[0] == A.shape[1]
        self._h = _lib.check_handle(self._l.slp_gs_create(A.shape[0], _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(data)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.slp_gs_destroy(self._h)
            self._h = None

    @property
    def num_levels(self):
        return int(self._l.slp_gs_num_levels(self._h))

    @property
    def sweep_kind(self):
        """0 launch per level, 1 one workgroup, 2 pipelined runs of narrow levels, 3 the same with the LDS window (slp_hip.h)."""
        return int(self._l.slp_gs_sweep_kind(self._h))

    @property
    def num_bands(self):
        """Workgroups that share the runs of narrow levels (bands of rows, slp_admm.hip); 0: every run on one workgroup."""
        return int(self._l.slp_gs_num_bands(self._h))

    def solve(self, b, lower_bounds, upper_bounds, x, maxiter=3, w=1, order=None):
        """``maxiter`` sweeps of ``x_i <- clamp(x_i + w (b_i - M_i x) / M_ii, lower_i, upper_i)``
        in natural row order (the reference accepts ``order`` and ignores it, :132-134)."""
        n = self.A.shape[0]
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous and x.flags.writeable):
            raise TypeError("x must be a writable C-contiguous float64 array (it is updated in place)")
        b = _lib.f64(b)
        lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lower_bounds, dtype=np.float64), (n,)))
        hi = np.ascontiguousarray(np.broadcast_to(np.asarray(upper_bounds, dtype=np.float64), (n,)))
        assert b.size == n and x.size == n
        _lib.check(self._l.slp_gs_solve(self._h, _lib.ptr(b), _lib.ptr(lo), _lib.ptr(hi), _lib.ptr(x), int(maxiter), float(w)))
        return x
