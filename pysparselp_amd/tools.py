"""Host-side problem transforms of the ADMM path (setup, once per solve).

Mirrors ``pysparselp.tools.precondition_constraints`` (reference tools.py:272-290)
and ``convert_to_standard_form_with_bounds`` (tools.py:88-127) on raw CSR
arrays.  They run once per solve on the host, like in the reference; the loop
they feed runs on the GPU.  (Moving them onto the device is SURVEY.md
section 8f "next-2".)

Entry order matters for bit-level parity with the reference: its scaled matrix
``sigma * a`` comes out of scipy's sparse-sparse product, which lists every row
in the reverse of its input order, and its stacked standard-form matrix goes
through COO->CSR, which sorts every row by column.  Both orders are reproduced
here explicitly, so the device sums rows in the same order as scipy does.
"""
import numpy as np

import scipy.sparse


class CsrArrays:
    """Raw CSR triple + shape; the form in which matrices are handed to the C ABI."""

    __slots__ = ("indptr", "indices", "data", "shape", "blocks")

    def __init__(self, indptr, indices, data, shape, blocks=None):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.shape = (int(shape[0]), int(shape[1]))
        self.blocks = blocks if blocks is not None else []

    @classmethod
    def from_any(cls, a):
        if a is None or isinstance(a, cls):
            return a
        return cls(a.indptr, a.indices, a.data, a.shape, getattr(a, "blocks", None))

    @property
    def nnz(self):
        return int(self.indptr[-1])

    def row_of_entry(self):
        return np.repeat(np.arange(self.shape[0]), np.diff(self.indptr))

    def tocsr(self):
        m = scipy.sparse.csr_matrix((self.data, self.indices, self.indptr), shape=self.shape)
        m.__dict__["blocks"] = self.blocks
        return m


def row_norm_scaling(a, alpha=2):
    """1 / ||a_i||_alpha per row, rows of norm 0 scaled by 1 (tools.py:274-279)."""
    rows = a.row_of_entry()
    absd = np.abs(a.data)
    powd = absd * absd if alpha == 2 else absd ** alpha
    sums = np.zeros(a.shape[0])
    # storage-order accumulation, one add per entry (scipy csr_matvec against a vector of ones)
    np.add.at(sums, rows, powd)
    norms = np.sqrt(sums) if alpha == 2 else sums ** (1.0 / alpha)
    norms[norms == 0] = 1
    return 1 / norms


def precondition_constraints(a, b, b2=None, alpha=2):
    """Scale every constraint row to unit ``alpha``-norm (tools.py:272-290).

    Returns ``(a_scaled, b_scaled[, b2_scaled])`` with ``a_scaled`` a
    :class:`CsrArrays` whose rows are stored in reversed entry order (see the
    module docstring); infinite bounds stay infinite.
    """
    a = CsrArrays.from_any(a)
    inv = row_norm_scaling(a, alpha)
    rows = a.row_of_entry()
    mirrored = a.indptr[rows] + a.indptr[rows + 1] - 1 - np.arange(a.nnz)
    data = (inv[rows] * a.data)[mirrored]
    indices = a.indices[mirrored]
    indptr = a.indptr
    nz = data != 0
    if not nz.all():  # the sparse product drops entries that underflow to exactly 0
        counts = np.bincount(rows[nz], minlength=a.shape[0])
        indptr = np.concatenate(([0], np.cumsum(counts)))
        data, indices = data[nz], indices[nz]
    a_p = CsrArrays(indptr, indices, data, a.shape, a.blocks)
    bp = inv * np.asarray(b, dtype=np.float64) if b is not None else None
    if b2 is None:
        return a_p, bp
    return a_p, bp, inv * np.asarray(b2, dtype=np.float64)


def _rows_sorted_by_column(rows, cols, vals, shape):
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    if rows.size:
        first = np.ones(rows.size, dtype=bool)
        first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
        if not first.all():  # duplicate (row, col) pairs are summed, as COO->CSR does
            starts = np.nonzero(first)[0]
            # summed left to right in storage order, as scipy's csr_sum_duplicates does (np.add.reduceat is NOT
            # sequential on runs of three or more; np.add.at is)
            run = np.cumsum(first) - 1
            summed = np.zeros(starts.size)
            np.add.at(summed, run, vals)
            vals = summed
            rows, cols = rows[starts], cols[starts]
    indptr = np.concatenate(([0], np.cumsum(np.bincount(rows, minlength=shape[0]))))
    return CsrArrays(indptr, cols, vals, shape)


def convert_to_standard_form_with_bounds(c, a_eq, beq, a_ineq, b_lower, b_upper, lb, ub, x0):
    """Slack-variable standard form (tools.py:88-127)::

        A = [[A_eq, 0], [A_ineq, -I]],  b = [b_eq; 0],  c = [c; 0],
        lb = [lb; b_lower],  ub = [ub; b_upper],  x0 = [x0; A_ineq x0]

    ``a_ineq`` is required (the reference fails with an unbound name without it).
    """
    a_eq, a_ineq = CsrArrays.from_any(a_eq), CsrArrays.from_any(a_ineq)
    if a_ineq is None:
        raise UnboundLocalError("local variable 'a_eq2' referenced before assignment (no inequality constraints)")
    ni, n = a_ineq.shape
    me = a_eq.shape[0] if a_eq is not None else 0
    slack_rows = np.arange(ni)
    rows = [me + a_ineq.row_of_entry(), me + slack_rows]
    cols = [a_ineq.indices.astype(np.int64), n + slack_rows]
    vals = [a_ineq.data, -np.ones(ni)]
    if a_eq is not None:
        rows.insert(0, a_eq.row_of_entry())
        cols.insert(0, a_eq.indices.astype(np.int64))
        vals.insert(0, a_eq.data)
        b2 = np.hstack((np.asarray(beq, dtype=np.float64), np.zeros(ni)))
        blocks = list(a_eq.blocks) + [(lo + me, hi + me) for lo, hi in a_ineq.blocks]
    else:
        b2 = np.zeros(ni)
        blocks = list(a_ineq.blocks)
    a2 = _rows_sorted_by_column(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), (me + ni, n + ni))
    a2.blocks = blocks
    b_lower = np.full(ni, -np.inf) if b_lower is None else np.asarray(b_lower, dtype=np.float64)
    b_upper = np.full(ni, np.inf) if b_upper is None else np.asarray(b_upper, dtype=np.float64)
    x0 = np.asarray(x0, dtype=np.float64)
    slack0 = a_ineq.tocsr() @ x0
    return (np.hstack((c, np.zeros(ni))), a2, b2, np.hstack((lb, b_lower)), np.hstack((ub, b_upper)),
            np.hstack((x0, slack0)))


def normal_matrix(a, gamma_eq, gamma_ineq):
    """``M = gamma_eq A^T A + gamma_ineq I`` as CSR with sorted rows (ADMM.py:93-101).

    Host sparse-sparse product (scipy), as in the reference; it is a one-off
    setup cost and only exists for problems whose ``M`` is sparse (SURVEY.md
    section 7, hard part 2).
    """
    s = a.tocsr()
    n = a.shape[1]
    m = (gamma_eq * (s.T * s) + gamma_ineq * scipy.sparse.eye(n, n)).tocsr()
    return CsrArrays(m.indptr, m.indices, m.data, m.shape)
