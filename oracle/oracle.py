"""CPU oracle: a restatement of PySparseLP's Chambolle-Pock and ADMM solvers.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pysparselp_amd/`` imports this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker or as the timed CPU
baseline -- never as the product path.

Parity: PINNED.  ``tests/golden/make_golden.py`` (run in the build container,
where the reference can be imported from /root/reference) checks every
function below bit-for-bit against the reference's own iterates, and
``tests/test_oracle_golden.py`` re-checks it on every run against the
committed fixtures and against the reference's golden curves
(tests/netlib_curves_SC105.json, tests/test_pott_segmentation_curves.json,
tests/test_kmedians.py:14, tests/test_l1_svm_results.json).

All arithmetic goes through ``oracle/slp_oracle.c`` (plain C, no FMA) or
elementwise numpy; no scipy.sparse arithmetic is used here, so the oracle does
not depend on the scipy version of the machine it runs on.  Matrices are the
three raw CSR arrays, entries kept in the order they are given (the reference
never sorts rows on this path except where stated below).

Reference citations are ``file:line`` under /root/reference/pysparselp/.
"""
import ctypes
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(
                "oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())"
            )
        lib = ctypes.CDLL(path)
        i64, dbl, vp = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
        lib.orc_csr_matvec.argtypes = [i64, vp, vp, vp, vp, vp]
        lib.orc_csr_rmatvec.argtypes = [i64, i64, vp, vp, vp, vp, vp]
        lib.orc_bounded_gauss_seidel.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, dbl]
        lib.orc_csr_to_csc.argtypes = [i64, i64, vp, vp, vp, vp, vp, vp]
        lib.orc_normal_matrix.argtypes = [i64, i64, vp, vp, vp, vp, vp, vp, dbl, dbl, vp, vp, vp]
        lib.orc_normal_matrix.restype = i64
        lib.orc_row_scale_l2.argtypes = [i64, vp, vp, vp]
        lib.orc_csc_rmatvec.argtypes = [i64, vp, vp, vp, vp, vp]
        lib.orc_csr_rmatvec_acc.argtypes = [i64, i64, vp, vp, vp, vp, vp]
        lib.orc_csr_rmatvec_acc.restype = i64
        lib.orc_scale_rows_reversed.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.orc_stack_standard_form.argtypes = [i64, i64, i64] + [vp] * 9
        lib.orc_sort_rows.argtypes = [i64, vp, vp, vp]
        lib.orc_sort_rows.restype = i64
        lib.orc_set_threads.argtypes = [ctypes.c_int]
        lib.orc_get_threads.restype = ctypes.c_int
        for f in (lib.orc_csr_matvec, lib.orc_csr_rmatvec, lib.orc_bounded_gauss_seidel,
                  lib.orc_csr_to_csc, lib.orc_row_scale_l2, lib.orc_csc_rmatvec, lib.orc_scale_rows_reversed,
                  lib.orc_stack_standard_form, lib.orc_set_threads):
            f.restype = None
        lib.orc_set_threads(int(os.environ.get("ORACLE_THREADS", "1")))
        _LIB = lib
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def set_threads(n):
    """Threads for the loops over independent rows / columns (default 1, or ``ORACLE_THREADS``): results do not depend on
    it -- no sum is split or reordered (slp_oracle.c) -- only the wall time of the full-size parity runs does.  The
    timed ``cpu_baseline`` of bench.py keeps 1: the reference is single-threaded."""
    _lib().orc_set_threads(int(n))


def get_threads():
    return int(_lib().orc_get_threads())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Csr:
    """Raw CSR triple (int64 indptr, int32 indices, float64 data) + shape."""

    def __init__(self, indptr, indices, data, shape):
        self._csc = None  # CSC arrays, built on first use by rmatvec when several threads are asked for
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = _f64(data)
        self.shape = (int(shape[0]), int(shape[1]))
        assert self.indptr.size == self.shape[0] + 1
        assert self.indices.size == self.data.size == int(self.indptr[-1])

    @property
    def nnz(self):
        return int(self.indptr[-1])

    def copy(self):
        return Csr(self.indptr.copy(), self.indices.copy(), self.data.copy(), self.shape)

    def toscipy(self):
        import scipy.sparse

        return scipy.sparse.csr_matrix(
            (self.data.copy(), self.indices.copy(), self.indptr.copy()), shape=self.shape
        )


class ProductsOnly:
    """A matrix this oracle knows only through its products -- for sizes whose CSR no host holds (BASELINE config 4: 240 GB
    per orientation).  The caller supplies ``A x``, ``y A`` and the same for ``|A| ** p`` (device copies, AFTER the caller has
    checked exactly those products against ``matvec`` / ``rmatvec`` below on row slices); the iteration that consumes them
    stays this file's restatement of the reference.  Only ``chambolle_pock_ppd`` on one-sided rows takes it."""

    def __init__(self, shape, matvec, rmatvec, abs_pow_matvec=None, abs_pow_rmatvec=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self._matvec, self._rmatvec = matvec, rmatvec
        self._abs_pow_matvec, self._abs_pow_rmatvec = abs_pow_matvec, abs_pow_rmatvec

    def powered(self, p):
        return ProductsOnly(self.shape, lambda x: self._abs_pow_matvec(x, p), lambda y: self._abs_pow_rmatvec(y, p))


class StreamedCsr(ProductsOnly):
    """A matrix this oracle holds ONE ROW CHUNK AT A TIME: ``chunk_source(k)`` hands over rows ``cuts[k] .. cuts[k + 1]`` as a
    CSR (scipy or ``Csr``; rows sorted by column) whenever a product needs them, and the chunk is dropped again -- BASELINE
    config 4's 2e10 entries (240 GB of CSR) never sit in host memory at once, yet every product is this file's own:
      ``a * x``   ``matvec`` of every chunk on its rows (csr_matvec: one chain per row);
      ``y * a``   ``rmatvec_acc`` chunk after chunk: every column's chain of additions continues in row order -- bit for bit
                  ``rmatvec`` of the stacked matrix (csc_matvec order; tests/test_oracle_golden.py);
      ``|a| ** p``  the same with the chunk's values mapped (``powered``: the sums behind ChambollePockPPD.py:134,144,161,172).
    ``rows=(r0, r1)`` restricts the operator to a row range that starts and ends at cuts (the equality / the inequality rows of
    an LP whose chunks were cut at m_eq).  ``cache_bytes`` > 0: chunks are kept, up to that many bytes, instead of asked for again.
    It IS a ``ProductsOnly`` to the solvers below (they consume nothing but the four products) -- with the difference that no
    product comes from the device."""

    def __init__(self, ncol, cuts, chunk_source, power=None, rows=None, cache_bytes=0, _cache=None):
        cuts = [int(c) for c in cuts]
        r0, r1 = (cuts[0], cuts[-1]) if rows is None else (int(rows[0]), int(rows[1]))
        assert r0 in cuts and r1 in cuts and r0 <= r1
        self._all_cuts, self._source, self._power, self._ncol = cuts, chunk_source, power, int(ncol)
        self._ks = [k for k in range(len(cuts) - 1) if cuts[k] >= r0 and cuts[k + 1] <= r1]
        self._r0 = r0
        self._cache_bytes, self._cache = int(cache_bytes), ({} if _cache is None else _cache)
        self.products = 0     # products formed so far (each one streams every chunk once)
        ProductsOnly.__init__(self, (r1 - r0, ncol), self._mv, self._rmv, lambda x, p: self.powered(p)._mv(x), lambda y, p: self.powered(p)._rmv(y))

    def powered(self, p):
        return StreamedCsr(self._ncol, self._all_cuts, self._source, power=p, rows=(self._r0, self._r0 + self.shape[0]),
                           cache_bytes=self._cache_bytes, _cache=self._cache)

    def row_range(self, r0, r1):
        return StreamedCsr(self._ncol, self._all_cuts, self._source, power=self._power, rows=(r0, r1), cache_bytes=self._cache_bytes,
                           _cache=self._cache)

    def stacked_rmatvec(self, front, y):
        """``y * [front; self]`` for the row range right in front of this one (the equality rows of the same LP): ONE chain of
        additions per column over both ranges, as the stored standard form [A_e 0; A_i -I] has it (ADMM.py:148)."""
        assert isinstance(front, StreamedCsr) and front._source is self._source and front._r0 + front.shape[0] == self._r0
        assert front._power == self._power
        return self.row_range(front._r0, self._r0 + self.shape[0])._rmv(y)

    def _chunks(self):
        for k in self._ks:
            c = self._cache.get(k)
            if c is None:
                c = as_csr(self._source(k))
                assert c.shape == (self._all_cuts[k + 1] - self._all_cuts[k], self._ncol)
                held = sum(v.data.nbytes + v.indices.nbytes + v.indptr.nbytes for v in self._cache.values())
                if held + c.data.nbytes + c.indices.nbytes + c.indptr.nbytes <= self._cache_bytes:
                    self._cache[k] = c
            if self._power is not None:
                c = Csr(c.indptr, c.indices, np.abs(c.data) ** self._power, c.shape)
            yield self._all_cuts[k] - self._r0, self._all_cuts[k + 1] - self._r0, c

    # (a zero vector: every term is +-0.0 and every chain starts at +0.0 -- the product is +0.0 in every entry whatever the
    # stored values, so nothing is streamed for it; the solvers' first iterations multiply by x0 = 0 and lambda = 0)
    def _mv(self, x):
        x = _f64(x)
        if not x.any():
            return np.zeros(self.shape[0])
        out = np.empty(self.shape[0])
        for r0, r1, c in self._chunks():
            out[r0:r1] = matvec(c, x)
        self.products += 1
        return out

    def _rmv(self, y):
        y = _f64(y)
        out = np.zeros(self._ncol)
        if not y.any():
            return out
        for r0, r1, c in self._chunks():
            rmatvec_acc(c, np.ascontiguousarray(y[r0:r1]), out)
        self.products += 1
        return out


def as_csr(a):
    """Accept a Csr, a scipy CSR matrix or None; entry order is preserved."""
    if a is None or isinstance(a, (Csr, ProductsOnly)):
        return a
    return Csr(a.indptr, a.indices, a.data, a.shape)


# --------------------------------------------------------------------------
# kernels (scipy.sparse._sparsetools semantics, see slp_oracle.c)
# --------------------------------------------------------------------------
def matvec(a, x):
    """``a * x`` (csr_matvec)."""
    x = _f64(x)
    assert x.size == a.shape[1]
    if isinstance(a, ProductsOnly):
        return a._matvec(x)
    y = np.empty(a.shape[0])
    _lib().orc_csr_matvec(a.shape[0], _p(a.indptr), _p(a.indices), _p(a.data), _p(x), _p(y))
    return y


def rmatvec(a, y):
    """``y * a`` for a CSR ``a`` (csc_matvec over the transposed view)."""
    y = _f64(y)
    assert y.size == a.shape[0]
    if isinstance(a, ProductsOnly):
        return a._rmatvec(y)
    out = np.empty(a.shape[1])
    if get_threads() > 1 and a.nnz > 1_000_000:
        # columns are independent: the same chain of additions per column from the (stable) CSC arrays
        if a._csc is None:
            a._csc = to_csc(a)
        cptr, crow, cdata = a._csc
        _lib().orc_csc_rmatvec(a.shape[1], _p(cptr), _p(crow), _p(cdata), _p(y), _p(out))
        return out
    _lib().orc_csr_rmatvec(a.shape[0], a.shape[1], _p(a.indptr), _p(a.indices), _p(a.data), _p(y), _p(out))
    return out


def rmatvec_acc(a, y, out):
    """``out += y * a`` continuing the chains of additions in ``out`` (in place; rows of ``a`` sorted by column): the next row
    chunk of a stacked matrix -- the result over all chunks is ``rmatvec`` of the stacked matrix bit for bit."""
    y = _f64(y)
    assert y.size == a.shape[0] and out.size == a.shape[1] and out.dtype == np.float64 and out.flags.c_contiguous
    bad = _lib().orc_csr_rmatvec_acc(a.shape[0], a.shape[1], _p(a.indptr), _p(a.indices), _p(a.data), _p(y), _p(out))
    if bad:
        raise ValueError(f"rmatvec_acc: row {bad - 1} is not sorted by column")
    return out


def to_csc(a):
    """CSC arrays of ``a`` (rows increasing inside every column)."""
    cptr = np.empty(a.shape[1] + 1, dtype=np.int64)
    crow = np.empty(a.nnz, dtype=np.int32)
    cdata = np.empty(a.nnz)
    _lib().orc_csr_to_csc(a.shape[0], a.shape[1], _p(a.indptr), _p(a.indices), _p(a.data), _p(cptr), _p(crow), _p(cdata))
    return cptr, crow, cdata


def normal_matrix(a, gamma_eq, gamma_ineq):
    """``(gamma_eq * a.T * a + gamma_ineq * I).tocsr()`` (ADMM.py:93-101)."""
    cptr, crow, cdata = to_csc(a)
    n = a.shape[1]
    args = (a.shape[0], n, _p(a.indptr), _p(a.indices), _p(a.data), _p(cptr), _p(crow), _p(cdata),
            float(gamma_eq), float(gamma_ineq))
    nnz = _lib().orc_normal_matrix(*args, None, None, None)
    mp = np.empty(n + 1, dtype=np.int64)
    mj = np.empty(nnz, dtype=np.int32)
    mx = np.empty(nnz)
    _lib().orc_normal_matrix(*args, _p(mp), _p(mj), _p(mx))
    return Csr(mp, mj, mx, (n, n))


def diagonal(m):
    """``m.diagonal()`` of a CSR matrix without duplicate entries."""
    d = np.zeros(m.shape[0])
    rows = np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))
    on = m.indices == rows
    d[rows[on]] = m.data[on]
    return d


class BoundedGaussSeidel:
    """gaussSiedel.pyx:83-153 ``boundedGaussSeidelClass``."""

    def __init__(self, m):
        self.m = as_csr(m)
        self.invD = 1 / diagonal(self.m)  # gaussSiedel.pyx:91-92

    def solve(self, b, lower_bounds, upper_bounds, x, maxiter=3, w=1, order=None):
        assert x.dtype == np.float64 and x.flags.c_contiguous
        m = self.m
        _lib().orc_bounded_gauss_seidel(
            m.shape[0], _p(m.indptr), _p(m.indices), _p(m.data), _p(self.invD),
            _p(_f64(b)), _p(_f64(lower_bounds)), _p(_f64(upper_bounds)), _p(x), int(maxiter), float(w))
        return x


# --------------------------------------------------------------------------
# problem transforms (tools.py)
# --------------------------------------------------------------------------
def precondition_constraints(a, b, b2=None):
    """tools.py:272-290 with alpha=2 (the only value used: ADMM.py:77,82,91).

    ``sigma * a`` is a sparse-sparse product (SMMP), whose output lists every
    row in the REVERSE of the order in which its columns were first touched;
    with one term per entry that is the reverse of the input row.  The values
    are ``sigma_i * a_ik`` (one rounding).  ``sigma * b`` is a csr_matvec over
    a diagonal: ``sigma_i * b_i`` (+-inf stays +-inf).
    """
    a = as_csr(a)
    inv_s = np.empty(a.shape[0])
    _lib().orc_row_scale_l2(a.shape[0], _p(a.indptr), _p(a.data), _p(inv_s))
    # scaled, every row reversed, exact zeros dropped (SMMP): slp_oracle.c orc_scale_rows_reversed
    counts = np.empty(a.shape[0], dtype=np.int64)
    args = (a.shape[0], _p(a.indptr), _p(a.indices), _p(a.data), _p(inv_s))
    _lib().orc_scale_rows_reversed(*args, _p(counts), None, None, None)
    ptr = np.zeros(a.shape[0] + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    ind = np.empty(int(ptr[-1]), dtype=np.int32)
    dat = np.empty(int(ptr[-1]))
    _lib().orc_scale_rows_reversed(*args, None, _p(ptr), _p(ind), _p(dat))
    a_p = Csr(ptr, ind, dat, a.shape)
    bp = inv_s * _f64(b) if b is not None else None
    if b2 is None:
        return a_p, bp
    return a_p, bp, inv_s * _f64(b2)


def _sorted_rows(indptr, indices, data, shape):
    """COO->CSR as scipy does it for hstack/vstack results: columns sorted
    inside every row (stable), duplicates summed in that order.  ``indices`` / ``data`` are sorted in place."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    dups = _lib().orc_sort_rows(shape[0], _p(indptr), _p(indices), _p(data))
    if dups == 0:
        return Csr(indptr, indices, data, shape)
    r = np.repeat(np.arange(shape[0]), np.diff(indptr))
    j, v = indices, data
    new = np.ones(r.size, dtype=bool)
    new[1:] = (r[1:] != r[:-1]) | (j[1:] != j[:-1])
    starts = np.nonzero(new)[0]
    # left to right in storage order like scipy's csr_sum_duplicates (np.add.reduceat is not sequential on
    # runs of three or more duplicates; np.add.at is)
    run = np.cumsum(new) - 1
    summed = np.zeros(starts.size)
    np.add.at(summed, run, v)
    r, j = r[starts], j[starts]
    ptr = np.zeros(shape[0] + 1, dtype=np.int64)
    np.add.at(ptr, r + 1, 1)
    return Csr(np.cumsum(ptr), j, summed, shape)


def convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0):
    """tools.py:88-127: A=[[Ae,0],[Ai,-I]], b=[be;0], bounds=[lb;b_lower],[ub;b_upper],
    x0=[x0; Ai*x0], c=[c;0].  Requires a_ineq (the reference hits an unbound
    name otherwise, tools.py:92,127)."""
    a_eq, a_ineq = as_csr(a_eq), as_csr(a_ineq)
    if a_ineq is None:
        raise UnboundLocalError("a_eq2 (reference fails the same way when a_ineq is None)")
    ni, n = a_ineq.shape
    # rows [Ai, -I]: one (n+i, -1) entry appended to every row, then every row sorted (slp_oracle.c)
    me = 0 if a_eq is None else a_eq.shape[0]
    lens = np.diff(a_ineq.indptr) + 1
    if a_eq is not None:
        lens = np.concatenate((np.diff(a_eq.indptr), lens))
        b2 = np.hstack((_f64(beq), np.zeros(ni)))
    else:
        b2 = np.zeros(ni)
    ptr = np.zeros(me + ni + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    ind = np.empty(int(ptr[-1]), dtype=np.int32)
    dat = np.empty(int(ptr[-1]))
    e = a_eq if a_eq is not None else Csr(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0), (0, n))
    _lib().orc_stack_standard_form(me, ni, n, _p(e.indptr), _p(e.indices), _p(e.data), _p(a_ineq.indptr), _p(a_ineq.indices),
                                   _p(a_ineq.data), _p(ptr), _p(ind), _p(dat))
    a2 = _sorted_rows(ptr, ind, dat, (me + ni, n + ni))
    if b_lower is None:
        b_lower = np.full(ni, -np.inf)
    if b_upper is None:
        b_upper = np.full(ni, np.inf)
    lb2 = np.hstack((_f64(lb), _f64(b_lower)))
    ub2 = np.hstack((_f64(ub), _f64(b_upper)))
    x02 = np.hstack((_f64(x0), matvec(a_ineq, x0)))
    c2 = np.hstack((_f64(c), np.zeros(ni)))
    return c2, a2, b2, lb2, ub2, x02


def one_sided(a_ineq, b_lower, b_upper):
    """ChambollePockPPD.py:74-88: [Ai[upper finite]; -Ai[lower finite]]."""
    a_ineq = as_csr(a_ineq)
    if a_ineq is None or b_lower is None:
        return a_ineq, b_upper
    up = np.nonzero(b_upper != np.inf)[0]
    lo = np.nonzero(b_lower != -np.inf)[0]

    def take(rows, sign):
        cnt = np.diff(a_ineq.indptr)[rows]
        ptr = np.concatenate(([0], np.cumsum(cnt)))
        src = np.repeat(a_ineq.indptr[rows], cnt) + (np.arange(ptr[-1]) - np.repeat(ptr[:-1], cnt))
        return ptr, a_ineq.indices[src], sign * a_ineq.data[src]

    if len(lo) > 0 and len(up) > 0:
        p1, j1, v1 = take(up, 1.0)
        p2, j2, v2 = take(lo, -1.0)
        a = Csr(np.concatenate((p1, p1[-1] + p2[1:])), np.concatenate((j1, j2)),
                np.concatenate((v1, v2)), (len(up) + len(lo), a_ineq.shape[1]))
    elif len(lo) > 0:
        a = Csr(a_ineq.indptr, a_ineq.indices, -a_ineq.data, a_ineq.shape)
    else:
        a = a_ineq
    return a, np.hstack((_f64(b_upper)[up], -_f64(b_lower)[lo]))


# --------------------------------------------------------------------------
# Chambolle-Pock (ChambollePockPPD.py:36-346)
# --------------------------------------------------------------------------
def cp_setup(a_eq, a_ineq, alpha=1):
    """Diagonal preconditioners T, Sigma_eq, Sigma_ineq (ChambollePockPPD.py:122-179)."""
    def powered(a, p):
        if isinstance(a, ProductsOnly):
            return a.powered(p)
        cp = Csr(a.indptr, a.indices, np.abs(a.data) ** p, a.shape)
        if get_threads() > 1 and a.nnz > 1_000_000:  # the CSC arrays of |A|^p are those of A with the same map applied
            if a._csc is None:
                a._csc = to_csc(a)
            cp._csc = (a._csc[0], a._csc[1], np.abs(a._csc[2]) ** p)
        return cp

    tmp = 0
    for a in (a_eq, a_ineq):
        if a is not None:
            tmp = tmp + rmatvec(powered(a, 2 - alpha), np.ones(a.shape[0]))  # :134,144 column sums, row order
    tmp[tmp == 0] = 1  # :152
    diag_t = 1 / tmp

    def sigma(a):
        if a is None:
            return None
        s = matvec(a.powered(alpha) if isinstance(a, ProductsOnly) else Csr(a.indptr, a.indices, np.abs(a.data) ** alpha, a.shape),
                   np.ones(a.shape[1]))  # :161,172
        s[s == 0] = 1
        return 1 / s

    return diag_t, sigma(a_eq), sigma(a_ineq)


def chambolle_pock_ppd(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=None, alpha=1, theta=1,
                       nb_max_iter=100, callback_func=None, max_time=None, nb_iter_plot=10,
                       iterate_hook=None):
    """Same contract as the reference function; ``iterate_hook(niter, x, y_eq,
    y_ineq)`` (oracle-only) sees the state after every iteration's primal
    update and before its dual update, i.e. exactly what a report would see."""
    c, lb, ub = _f64(c), _f64(lb), _f64(ub)
    start = time.perf_counter()
    a_eq, a_ineq = as_csr(a_eq), as_csr(a_ineq)
    if a_eq is not None and a_eq.shape[0] == 0:  # :70-72
        a_eq, beq = None, None
    if a_ineq is not None and a_ineq.shape[0] == 0:
        a_ineq = None
    if a_ineq is not None:
        a_ineq, b_ineq = one_sided(a_ineq, b_lower, b_upper)  # :74-88
        b_ineq = _f64(b_ineq)
    x = _f64(x0).copy() if x0 is not None else np.zeros(c.size)  # :91-94
    n = c.size
    if a_eq is None and a_ineq is None:  # :147-151
        x = np.zeros_like(lb)
        x[c > 0] = lb[c > 0]
        x[c < 0] = ub[c < 0]
        return x
    diag_t, sig_eq, sig_ineq = cp_setup(a_eq, a_ineq, alpha)
    if a_eq is not None:
        beq = _f64(beq)
        y_eq = np.zeros(a_eq.shape[0])
    else:
        y_eq = None
    y_ineq = np.zeros(a_ineq.shape[0]) if a_ineq is not None else None
    x3 = x
    niter = 0
    while niter < nb_max_iter:  # :195
        d = c
        if a_eq is not None:
            d = d + rmatvec(a_eq, y_eq)  # :206
        if a_ineq is not None:
            d = d + rmatvec(a_ineq, y_ineq)  # :216
        x2 = x - diag_t * d  # :220
        np.maximum(x2, lb, x2)
        np.minimum(x2, ub, x2)
        x3 = (1 + theta) * x2 - theta * x  # :226
        x = x2
        if a_eq is not None:
            r_eq = matvec(a_eq, x3) - beq  # :235
        if a_ineq is not None:
            r_ineq = matvec(a_ineq, x3) - b_ineq  # :240
        if iterate_hook is not None:
            iterate_hook(niter, x, y_eq, y_ineq)
        if niter % nb_iter_plot == 0:  # :242-329
            elapsed = time.perf_counter() - start
            if (max_time is not None) and elapsed > max_time:
                break
            energy1 = c.dot(x)
            x4 = lb.copy()
            x4[d < 0] = ub[d < 0]  # :260-261
            energy2 = c.dot(x4)
            max_violated_equality = 0
            max_violated_inequality = 0
            if a_eq is not None:
                energy1 += y_eq.dot(matvec(a_eq, x) - beq)
                energy2 += y_eq.dot(matvec(a_eq, x4) - beq)
                max_violated_equality = np.max(np.abs(r_eq))
            if a_ineq is not None:
                energy1 += y_ineq.dot(matvec(a_ineq, x) - b_ineq)
                energy2 += y_ineq.dot(matvec(a_ineq, x4) - b_ineq)
                max_violated_inequality = np.max(r_ineq)
            # :283 overwrites it with the violation at x itself (force_integer=False);
            # the reference dereferences a_ineq unconditionally there.
            if a_ineq is None:
                raise AttributeError("'NoneType' object has no attribute (reference: ChambollePockPPD.py:283)")
            max_violated_inequality = np.max(matvec(a_ineq, x) - b_ineq)
            if callback_func is not None:
                callback_func(niter, x, energy1, energy2, elapsed, max_violated_equality, max_violated_inequality)
        if a_eq is not None:
            y_eq = y_eq + sig_eq * r_eq  # :334
        if a_ineq is not None:
            y_ineq = y_ineq + sig_ineq * r_ineq  # :339-341
            np.maximum(y_ineq, 0, y_ineq)
        niter += 1
    return x[:n], None


# --------------------------------------------------------------------------
# ADMM (ADMM.py:47-269, projected Gauss-Seidel x-step: the shipped flags :66-71)
# --------------------------------------------------------------------------
def admm_setup(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=None, gamma_eq=2, gamma_ineq=3,
               use_preconditioning=True):
    """ADMM.py:73-101: row scaling, standard form, second row scaling, M, Atb."""
    c = _f64(c)
    a_eq, a_ineq = as_csr(a_eq), as_csr(a_ineq)
    if x0 is None:
        x0 = np.zeros(c.size)
    if a_eq is not None:
        a_eq, beq = precondition_constraints(a_eq, beq)
    if a_ineq is not None:
        a_ineq, b_lower, b_upper = precondition_constraints(a_ineq, b_lower, b_upper)
    c2, a, b, lb2, ub2, x = convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0)
    if use_preconditioning:
        a, b = precondition_constraints(a, b)
    m = normal_matrix(a, gamma_eq, gamma_ineq)
    atb = rmatvec(a, b)  # :95
    return dict(c=c2, a=a, b=b, lb=lb2, ub=ub2, x0=x, m=m, atb=atb)


def lp_admm(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=None, gamma_eq=2, gamma_ineq=3, nb_iter=100,
            callback_func=None, max_time=None, use_preconditioning=True, nb_iter_plot=10, iterate_hook=None):
    n = np.asarray(c).size
    s = admm_setup(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0, gamma_eq, gamma_ineq, use_preconditioning)
    c, a, b, lb, ub, x, m, atb = (s[k] for k in ("c", "a", "b", "lb", "ub", "x0", "m", "atb"))
    x = x.copy()
    xp = np.maximum(x, 0)  # :98
    lambda_eq = np.zeros(a.shape[0])
    lambda_ineq = np.zeros(x.shape)
    bs = BoundedGaussSeidel(m)  # :139

    def energy(x, xp, lambda_eq, lambda_ineq):  # :124-132
        r = matvec(a, x) - b
        return (c.dot(x) + 0.5 * gamma_eq * np.sum(r ** 2) + 0.5 * gamma_ineq * np.sum((x - xp) ** 2)
                + lambda_eq.dot(matvec(a, x) - b) + lambda_ineq.dot(x - xp))

    start = time.perf_counter()
    i = 0
    while i <= nb_iter:  # :143 (nb_cg_iter == 1): nb_iter+1 sweeps
        y = -c + gamma_eq * atb + gamma_ineq * xp - rmatvec(a, lambda_eq) - lambda_ineq  # :148
        bs.solve(y, lb, ub, x, maxiter=1, w=1)  # :162
        if iterate_hook is not None:
            iterate_hook(i, x[0:n], x, lambda_eq)
        if i % nb_iter_plot == 0:  # :213-248
            elapsed = time.perf_counter() - start
            if max_time is not None and elapsed > max_time:
                break
            energy1 = energy(x, xp, lambda_eq, lambda_ineq)
            r = matvec(a, x) - b
            max_violated_equality = np.max(np.abs(r))
            max_violated_inequality = max(0, -np.min(x))
            if callback_func is not None:
                callback_func(i, x[0:n], energy1, energy1, elapsed, max_violated_equality, max_violated_inequality)
        xp = x  # :259 (alias)
        lambda_eq = lambda_eq + gamma_eq * (matvec(a, x) - b)  # :261-263
        i += 1
    return x[0:n]


# --------------------------------------------------------------------------
# ADMM, conjugate-gradient x-step (ADMM.py:182-201 + conjugateGradientLinearSolver.py:30-52):
# the reference's own alternative to the Gauss-Seidel sweep, selected by its
# hard-coded flags (use_cg=True, use_bounded_gauss_siedel=False at ADMM.py:66-71).
# It is the ADMM form that still exists when M = gamma_eq A^T A + gamma_ineq I
# cannot be formed (hard part 2), because M only enters
# through products M v.
#   explicit_m=True : M v by csr_matvec on the explicit M, exactly the reference.
#   explicit_m=False: M v = gamma_eq A^T (A v) + gamma_ineq v, matrix-free (what
#                     the device does); same maths, fp64 rounding differences only.
# --------------------------------------------------------------------------
def lp_admm_cg(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=None, gamma_eq=2, gamma_ineq=3, nb_iter=100,
               callback_func=None, max_time=None, use_preconditioning=True, nb_iter_plot=10, explicit_m=False,
               iterate_hook=None):
    n = np.asarray(c).size
    c = _f64(c)
    if isinstance(a_ineq, ProductsOnly):
        assert (a_eq is None or isinstance(a_eq, ProductsOnly)) and b_lower is None and use_preconditioning and not explicit_m
        return _lp_admm_cg_over_products(c, a_ineq, b_upper, lb, ub, x0, gamma_eq, gamma_ineq, nb_iter, callback_func, max_time,
                                         nb_iter_plot, iterate_hook, a_eq, beq)
    a_eq, a_ineq = as_csr(a_eq), as_csr(a_ineq)
    if x0 is None:
        x0 = np.zeros(c.size)
    if a_eq is not None:
        a_eq, beq = precondition_constraints(a_eq, beq)
    if a_ineq is not None:
        a_ineq, b_lower, b_upper = precondition_constraints(a_ineq, b_lower, b_upper)
    c, a, b, lb, ub, x = convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0)
    if use_preconditioning:
        a, b = precondition_constraints(a, b)
    m = normal_matrix(a, gamma_eq, gamma_ineq) if explicit_m else None
    return _admm_cg_iterations(n, c, lambda v: matvec(a, v), lambda y: rmatvec(a, y), a.shape[0], b, lb, ub, x, gamma_eq, gamma_ineq,
                               nb_iter, callback_func, max_time, nb_iter_plot, iterate_hook,
                               (lambda v: matvec(m, v)) if explicit_m else None)


def _lp_admm_cg_over_products(c, ops, b_upper, lb, ub, x0, gamma_eq, gamma_ineq, nb_iter, callback_func, max_time, nb_iter_plot,
                              iterate_hook, ops_eq=None, beq=None):
    """``lp_admm_cg`` for an LP ``A_e x = b_e, A_i x <= b_upper`` known through its products only (``ProductsOnly``: a size whose
    CSR no host holds; ``ops_eq`` None: all inequalities).  The same set-up and the same iteration, with the two row scalings of
    ADMM.py:77,82,91 (tools.py:272-290) and the slack columns of tools.py:88-127 (inequality rows only) carried BESIDE the products
    instead of inside stored entries: with K = [A_e; A_i], s1 = 1 / |row of K|, s2 = 1 / |row of [s1 A_e, 0; s1 A_i, -I]|,
        a v   = s2 * (s1 * (K v[:n]) - [0; v[n:]])          a^T y = [K^T (s1 * s2 * y) ; -(s2 * y)[m_e:]]
    Equal to the stored-entry form in exact arithmetic; in fp64 the roundings differ (s * (sum of a x) for sum of (s a) x; the
    norms from one sum of squares), as the device's deferred row scaling does.  Pinned against the stored-entry form on host
    CSRs by tests/test_oracle_golden.py (1e-12)."""
    mi, n = ops.shape
    me = ops_eq.shape[0] if ops_eq is not None else 0
    m = me + mi
    ones = np.ones(n)

    def k_mv(v):      # K v, equality rows first
        return ops._matvec(v) if not me else np.concatenate((ops_eq._matvec(v), ops._matvec(v)))

    def k_rmv(y):     # K^T y: the stacked matrix's single chain per column = the equality rows' chain continued by the others'
        if not me:
            return ops._rmatvec(y)
        if hasattr(ops, "stacked_rmatvec"):
            return ops.stacked_rmatvec(ops_eq, y)
        return ops_eq._rmatvec(np.ascontiguousarray(y[:me])) + ops._rmatvec(np.ascontiguousarray(y[me:]))

    rowsq = ops._abs_pow_matvec(ones, 2.0)  # sum_j a_ij^2, storage order (orc_row_scale_l2)
    if me:
        rowsq = np.concatenate((ops_eq._abs_pow_matvec(ones, 2.0), rowsq))
    norm1 = np.sqrt(rowsq)
    norm1[norm1 == 0] = 1
    s1 = 1 / norm1
    bu1 = s1[me:] * _f64(b_upper)
    slack_sq = np.concatenate((np.zeros(me), np.ones(mi)))
    norm2 = np.sqrt((s1 * s1) * rowsq + slack_sq)   # rows of [s1 A_e, 0; s1 A_i, -I]
    norm2[norm2 == 0] = 1
    s2 = 1 / norm2
    s12 = s1 * s2
    x0 = np.zeros(n) if x0 is None else _f64(x0)
    x = np.hstack((x0, s1[me:] * ops._matvec(x0)))            # tools.py:125  [x0 ; Ai x0]
    c2 = np.hstack((c, np.zeros(mi)))
    lb2 = np.hstack((_f64(lb), np.full(mi, -np.inf)))
    ub2 = np.hstack((_f64(ub), bu1))
    b = s2 * np.concatenate((s1[:me] * _f64(beq) if me else np.zeros(0), np.zeros(mi)))

    def a_mv(v):
        return s2 * (s1 * k_mv(np.ascontiguousarray(v[:n])) - np.concatenate((np.zeros(me), v[n:])))

    def a_rmv(y):
        return np.hstack((k_rmv(s12 * y), -(s2 * y)[me:]))

    return _admm_cg_iterations(n, c2, a_mv, a_rmv, m, b, lb2, ub2, x, gamma_eq, gamma_ineq, nb_iter, callback_func, max_time,
                               nb_iter_plot, iterate_hook, None)


def _admm_cg_iterations(n, c, a_mv, a_rmv, nrows, b, lb, ub, x, gamma_eq, gamma_ineq, nb_iter, callback_func, max_time, nb_iter_plot,
                        iterate_hook, m_explicit):
    """The iteration of ``lp_admm_cg`` (ADMM.py:143-268 with the use_cg flags) over the standard-form operator ``a``."""
    atb = a_rmv(b)
    if m_explicit is not None:
        m_apply = m_explicit
    else:
        def m_apply(v):
            return gamma_eq * a_rmv(a_mv(v)) + gamma_ineq * v

    xp = np.maximum(x, 0)  # :98
    lambda_eq = np.zeros(nrows)
    lambda_ineq = np.zeros(x.shape)
    speed = np.zeros(x.shape)  # :136
    alpha = 1.4  # :140

    def energy(x, xp, lambda_eq, lambda_ineq):  # :124-132
        r = a_mv(x) - b
        return (c.dot(x) + 0.5 * gamma_eq * np.sum(r ** 2) + 0.5 * gamma_ineq * np.sum((x - xp) ** 2)
                + lambda_eq.dot(a_mv(x) - b) + lambda_ineq.dot(x - xp))

    start = time.perf_counter()
    i = 0
    while i <= nb_iter:  # :143
        y = -c + gamma_eq * atb + gamma_ineq * xp - a_rmv(lambda_eq) - lambda_ineq  # :148
        xprev = x.copy()  # :184
        direction = speed  # :190-194: exact line search along the previous displacement
        t = -direction.dot(m_apply(x) - y)
        if abs(t) > 0:
            step_length = t / (direction.dot(m_apply(direction)))
            x = x + step_length * direction
        # :199 conjgrad(m, y, maxiter=1, x0=x)  (conjugateGradientLinearSolver.py:36-46)
        r = y - m_apply(x)
        p = r
        rsold = r.dot(r)
        a_p = m_apply(p)
        alpha_cg = rsold / (p.dot(a_p))
        x = x + alpha_cg * p
        speed = x - xprev  # :200
        x = alpha * x + (1 - alpha) * xp  # :201 over-relaxation
        if iterate_hook is not None:
            iterate_hook(i, x[0:n], x, lambda_eq)
        if i % nb_iter_plot == 0:  # :213-248
            elapsed = time.perf_counter() - start
            if max_time is not None and elapsed > max_time:
                break
            energy1 = energy(x, xp, lambda_eq, lambda_ineq)
            r = a_mv(x) - b
            max_violated_equality = np.max(np.abs(r))
            max_violated_inequality = max(0, -np.min(x))
            if callback_func is not None:
                callback_func(i, x[0:n], energy1, energy1, elapsed, max_violated_equality, max_violated_inequality)
        xp = x.copy() + lambda_ineq / gamma_ineq  # :253-256
        xp = np.maximum(xp, lb)
        xp = np.minimum(xp, ub)
        lambda_ineq = lambda_ineq + gamma_ineq * (x - xp)
        lambda_eq = lambda_eq + gamma_eq * (a_mv(x) - b)  # :261-263
        i += 1
    return x[0:n]


# --------------------------------------------------------------------------
# ADMM, unbounded Gauss-Seidel x-step with over-relaxation (ADMM.py:164-181, :252-256): the third flag-selected
# branch of the reference (use_unbounded_gauss_siedel=True, use_bounded_gauss_siedel=False).  One plain SOR sweep
# of M x = y (gaussSiedel.pyx:21-79), x = 1.4 x - 0.4 xp, then the explicit projection / lambda_ineq update.
# --------------------------------------------------------------------------
def lp_admm_gs_unbounded(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=None, gamma_eq=2, gamma_ineq=3, nb_iter=100,
                         callback_func=None, max_time=None, use_preconditioning=True, nb_iter_plot=10):
    n = np.asarray(c).size
    s = admm_setup(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0, gamma_eq, gamma_ineq, use_preconditioning)
    c, a, b, lb, ub, x, m, atb = (s[k] for k in ("c", "a", "b", "lb", "ub", "x0", "m", "atb"))
    x = x.copy()
    xp = np.maximum(x, 0)
    lambda_eq = np.zeros(a.shape[0])
    lambda_ineq = np.zeros(x.shape)
    d = diagonal(m)
    invd = 1 / d
    alpha = 1.4
    lib = _lib()
    lib.orc_gauss_seidel.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_double]
    lib.orc_gauss_seidel.restype = None

    def energy(x, xp, lambda_eq, lambda_ineq):
        r = matvec(a, x) - b
        return (c.dot(x) + 0.5 * gamma_eq * np.sum(r ** 2) + 0.5 * gamma_ineq * np.sum((x - xp) ** 2)
                + lambda_eq.dot(matvec(a, x) - b) + lambda_ineq.dot(x - xp))

    start = time.perf_counter()
    i = 0
    while i <= nb_iter:
        y = -c + gamma_eq * atb + gamma_ineq * xp - rmatvec(a, lambda_eq) - lambda_ineq  # :148
        lib.orc_gauss_seidel(m.shape[0], _p(m.indptr), _p(m.indices), _p(m.data), _p(d), _p(invd), _p(_f64(y)), _p(x), 1, 1.0)  # :179
        x = alpha * x + (1 - alpha) * xp  # :181
        if i % nb_iter_plot == 0:
            elapsed = time.perf_counter() - start
            if max_time is not None and elapsed > max_time:
                break
            energy1 = energy(x, xp, lambda_eq, lambda_ineq)
            r = matvec(a, x) - b
            if callback_func is not None:
                callback_func(i, x[0:n], energy1, energy1, elapsed, np.max(np.abs(r)), max(0, -np.min(x)))
        xp = x.copy() + lambda_ineq / gamma_ineq  # :253-256
        xp = np.maximum(xp, lb)
        xp = np.minimum(xp, ub)
        lambda_ineq = lambda_ineq + gamma_ineq * (x - xp)
        lambda_eq = lambda_eq + gamma_eq * (matvec(a, x) - b)
        i += 1
    return x[0:n]


# ---------------------------------------------------------------------------------------------------
# ADMM with one copy of the variables per block of constraints (ADMMBlocks.py:45-352).
# The per-block equality-constrained least-squares problems are solved exactly with scipy's SuperLU
# (scipy.sparse.linalg.splu), like in the reference (:224): third-party code, present on the GPU box.
def _blocks_of(a, given):
    if given is not None:
        return [tuple(int(v) for v in b) for b in given]
    return [tuple(int(v) for v in b) for b in getattr(a, "blocks", [])] if a is not None else []


def standard_form_blocks(blocks_eq, blocks_ineq, m_eq):
    """tools.py:104-111: the blocks of [A_eq 0; A_ineq -I] are those of A_eq followed by those of A_ineq shifted by the
    number of equality rows."""
    return list(blocks_eq) + [(lo + m_eq, hi + m_eq) for lo, hi in blocks_ineq]


def lp_admm_block_decomposition(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0=None, gamma_ineq=0.7, nb_iter=100,
                                callback_func=None, max_time=None, use_preconditioning=True, use_lu=True, nb_iter_plot=10,
                                blocks_eq=None, blocks_ineq=None):
    """ADMMBlocks.py:45-352 (``blocks_*``: the (first_row, last_row) groups the modelling layer records, SparseLP.py:93-95;
    read from the matrices' ``blocks`` attribute when not given).  ``max_time=None`` means no limit (the reference
    compares ``elapsed > None``, :312, which raises in Python 3)."""
    import time

    import scipy.sparse
    import scipy.sparse.linalg

    blocks_eq, blocks_ineq = _blocks_of(a_eq, blocks_eq), _blocks_of(a_ineq, blocks_ineq)
    n = np.asarray(c).size
    if x0 is None:
        x0 = np.zeros(n)
    m_eq = 0 if a_eq is None else as_csr(a_eq).shape[0]
    c, a, b, lb, ub, x0 = convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0)  # :79-81
    blocks = standard_form_blocks(blocks_eq, blocks_ineq, m_eq)
    a_sp = scipy.sparse.csr_matrix((a.data, a.indices, a.indptr), shape=a.shape)
    xp = np.minimum(np.maximum(x0.copy(), lb), ub)  # :84-86
    nb_used = np.zeros(x0.shape)
    ids_of, lus, beqs = [], [], []
    for lo, hi in blocks:  # :178-243, one group per block (:119)
        id_rows = np.arange(lo, hi + 1)
        sub_a = a_sp[id_rows, :]
        t = np.array(np.abs(sub_a).sum(axis=0)).ravel()
        ids = np.nonzero(t)[0]
        ids_of.append(ids)
        nb_used[ids] += 1
        sub_a2 = sub_a[:, ids]
        m = scipy.sparse.vstack((
            scipy.sparse.hstack((gamma_ineq * scipy.sparse.eye(sub_a2.shape[1], sub_a2.shape[1]), sub_a2.T)),
            scipy.sparse.hstack((sub_a2, scipy.sparse.csc_matrix((sub_a2.shape[0], sub_a2.shape[0])))),
        )).tocsr()
        lus.append(scipy.sparse.linalg.splu(m.tocsc()))
        beqs.append(b[id_rows])
    nb = len(blocks)
    x = [x0[ids_of[k]] for k in range(nb)]
    lam = [np.zeros(ids_of[k].shape) for k in range(nb)]
    alpha = 1.95  # :262
    start = time.perf_counter()
    i = 0
    while i <= nb_iter:
        for k in range(nb):  # :268-284
            y = np.hstack((gamma_ineq * xp[ids_of[k]] - lam[k], beqs[k]))
            xv = lus[k].solve(y)
            x[k] = alpha * xv[: x[k].shape[0]] + (1 - alpha) * xp[ids_of[k]]
        xp[nb_used > 0] = 0  # :290-299
        for k in range(nb):
            xp[ids_of[k]] += x[k] + lam[k] / gamma_ineq
        xp = xp - c / gamma_ineq
        xp = xp / np.maximum(nb_used, 1)
        xp = np.maximum(xp, lb)
        xp = np.minimum(xp, ub)
        for k in range(nb):  # :302-307
            lam[k] = lam[k] + gamma_ineq * (x[k] - xp[ids_of[k]])
        if i % nb_iter_plot == 0:  # :309-350
            elapsed = time.perf_counter() - start
            if max_time is not None and elapsed > max_time:
                break
            en = c.dot(xp)  # :246-253
            for k in range(nb):
                diff = x[k] - xp[ids_of[k]]
                en += 0.5 * gamma_ineq * np.sum(diff ** 2) + lam[k].dot(diff)
            if callback_func is not None:
                callback_func(i, xp[0:n], en, en, elapsed, 0, 0)
        i += 1
    return xp[0:n]


# ---------------------------------------------------------------------------------------------------
# The same block iteration with the per-block KKT solve done MATRIX-FREE (what exists at BASELINE config 5: a block of 5e5
# rows x 5e7 columns has no sparse LU).  The solve of ADMMBlocks.py:268-284 is the projection of
# v = xp[ids] - lambda / gamma onto {A_g x - s = 0} in the block's standard form [A_g, -I] (tools.py:88-127); eliminating it:
#   dual form    (A_g A_g^T + I) nu = A_g v - v_s ,  x = v - A_g^T nu , s = v_s + nu
#   primal form  (I + A_g^T A_g) x  = v + A_g^T v_s , s = A_g x                   (taken when a block has at least as many rows as columns)
# solved by conjugate gradients warm-started from the previous iteration's solution, stopping test |r|^2 <= tol^2 |rhs|^2
# looked at every `check_every` steps.  This is the CPU restatement of csrc/slp_blocks.hip's row-block form (same order of the
# elementwise operations); it agrees with lp_admm_block_decomposition's LU form to the CG tolerance
# (tests/test_oracle_golden.py) and is what bench.py times as the cpu_baseline of `--method admm_blocks`.
# All rows are inequalities  b_lower <= A_g x <= b_upper.
def lp_admm_blocks_cg(c, blocks, lb, ub, gamma=0.7, alpha=1.95, nb_iter=10, cg_tol=1e-13, cg_max_steps=500, check_every=10,
                      primal=None, iterate_hook=None):
    """``blocks``: list of ``(a_g, b_lower_g or None, b_upper_g)`` with ``a_g`` the block's rows over all n variables.
    Runs ``nb_iter`` iterations from x0 = 0; returns ``(xp, cg_steps)``; ``iterate_hook(i)`` is called after iteration i (a hook with
    the attribute ``wants_steps = True`` is called as ``iterate_hook(i, cg_steps_so_far)``)."""
    c, lb, ub = _f64(c), _f64(lb), _f64(ub)
    n = c.size
    st = []
    copies = np.zeros(n)
    for a_given, bl, bu in blocks:
        # a block may be handed over as a CALLABLE that returns its rows: it is then loaded for this set-up pass and again for each
        # of its updates, and dropped in between -- eight 5e5 x 5e7 blocks (BASELINE config 5: 240 GB of CSR, twice that with the
        # column-major copies the threaded products keep) never sit in host memory together (tools/c5_oracle_parity.py)
        a = as_csr(a_given() if callable(a_given) else a_given)
        m = a.shape[0]
        bu = _f64(bu)
        bl = np.full(m, -np.inf) if bl is None else _f64(bl)
        used = np.zeros(n, dtype=bool)
        used[a.indices] = True   # a copy exists only for the variables the block uses (:183-185)
        copies += used
        st.append({"a": a_given if callable(a_given) else a, "m": m, "slo": bl, "shi": bu, "used": used, "lam": np.zeros(n), "nu": np.zeros(m), "xsol": np.zeros(n),
                   "xps": np.minimum(np.maximum(0.0, bl), bu), "lams": np.zeros(m),
                   "primal": (m >= n and m > 0) if primal is None else bool(primal)})
    xp = np.minimum(np.maximum(0.0, lb), ub)   # :84-86 with x0 = 0
    steps = 0

    def cg(apply, rhs, r, sol):
        nonlocal steps
        rhs2, rs = float(rhs.dot(rhs)), float(r.dot(r))
        d = r.copy()
        it = 0
        while it < cg_max_steps:
            if not rs > cg_tol * cg_tol * rhs2:
                break
            chunk = min(check_every, cg_max_steps - it)
            for _ in range(chunk):
                q = apply(d)
                pq = float(d.dot(q))
                al = rs / pq if (rs > 0.0 and pq > 0.0) else 0.0
                sol += al * d
                r -= al * q
                rsn = float(r.dot(r))
                be = rsn / rs if rs > 0.0 else 0.0
                rs = rsn
                d = r + be * d
            it += chunk
            steps += chunk

    for i in range(nb_iter):
        total = np.zeros(n)
        xs_all = []
        for k, s in enumerate(st):
            a, m = s["a"], s["m"]
            if callable(a):
                a = as_csr(a())
            v = xp - s["lam"] / gamma
            vs = s["xps"] - s["lams"] / gamma
            if s["primal"]:
                u = rmatvec(a, vs)
                rhs = v + u
                r = rhs - (s["xsol"] + rmatvec(a, matvec(a, s["xsol"])))
                cg(lambda d: rmatvec(a, matvec(a, d)) + d, rhs, r, s["xsol"])
                w = matvec(a, s["xsol"])
                x = alpha * s["xsol"] + (1.0 - alpha) * xp
                xs = alpha * w + (1.0 - alpha) * s["xps"]
            elif m > 0:
                rhs = matvec(a, v) - vs
                r = rhs - (matvec(a, rmatvec(a, s["nu"])) + s["nu"])
                cg(lambda d: matvec(a, rmatvec(a, d)) + d, rhs, r, s["nu"])
                x = alpha * (v - rmatvec(a, s["nu"])) + (1.0 - alpha) * xp
                xs = alpha * (vs + s["nu"]) + (1.0 - alpha) * s["xps"]
            else:
                x = alpha * v + (1.0 - alpha) * xp
                xs = np.zeros(0)
            s["x"] = x
            acc = np.where(s["used"], x + s["lam"] / gamma, 0.0)
            total = acc if k == 0 else total + acc   # summands added in block order
            # the slack copies belong to this block alone: consensus, clamp and multiplier at once (:290-307 for one copy, cost 0)
            t = np.minimum(np.maximum(xs + s["lams"] / gamma, s["slo"]), s["shi"])
            s["lams"] = s["lams"] + gamma * (xs - t)
            s["xps"] = t
            a = None   # (a streamed block goes again)
        t = np.where(copies > 0, total, xp) - c / gamma
        t = t / np.maximum(copies, 1.0)
        xp = np.minimum(np.maximum(t, lb), ub)
        for s in st:
            s["lam"] = np.where(s["used"], s["lam"] + gamma * (s["x"] - xp), s["lam"])
        if iterate_hook is not None:
            if getattr(iterate_hook, "wants_steps", False):
                iterate_hook(i, steps)      # (cumulative conjugate-gradient steps so far)
            else:
                iterate_hook(i)
    return xp, steps
