"""Row partition of the constraint matrix over the GPUs of one node.

The reference is single-process; this layer is new.  Decomposition (DESIGN.md
"multi-GPU"): rank g owns a contiguous block of constraint rows K_g together
with everything indexed by rows (b_g, Sigma_g, y_g, lambda_g and, for ADMM, the
slack variables of those rows); everything indexed by original variables (x, z,
c, T, lb, ub) is replicated.  The only data-path exchange is the sum-all-reduce
of the n partial column sums ``K_g^T y_g`` (RCCL inside libslp_hip.so, on the
compute stream); dot products over [original | slack] unknowns add the
replicated part to the all-reduced slack part.

The control plane (exchange of the RCCL unique id, nothing else) goes through
``torch.distributed`` with the gloo backend when bench.py is launched by
``torch.distributed.run``.
"""
import ctypes

from . import _lib


def row_block(m, world, rank):
    """``(first_row, row_count)`` of rank ``rank``: contiguous blocks of ceil(m / world) rows.

    The synthetic rows all have the same expected number of entries, so equal
    row counts balance the stored entries to within their sampling noise."""
    assert 0 <= rank < world and m >= 0
    per = (m + world - 1) // world
    first = min(rank * per, m)
    return first, min(per, m - first)


def row_block_by_nnz(indptr, world, rank):
    """Contiguous row blocks with (nearly) equal stored entries, for matrices with ragged rows:
    block g ends at the first row where the cumulative count reaches (g+1)/world of the total."""
    import numpy as np

    indptr = np.asarray(indptr)
    m = indptr.size - 1
    total = int(indptr[-1])
    cuts = [0]
    for g in range(1, world):
        cuts.append(int(np.searchsorted(indptr, total * g / world, side="left")))
    cuts.append(m)
    cuts = [min(max(c, 0), m) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1] - cuts[rank]


def exchange_unique_id(dist, rank, make_id):
    """Rank 0 creates the 128-byte RCCL unique id, every rank receives it (gloo broadcast)."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = box[0]
    assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
    return bytes(uid)


def init_comm(dist, rank, world):
    """Create the RCCL communicator inside libslp_hip.so for this process's GPU."""
    lib = _lib.lib()

    def make_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.slp_comm_unique_id(buf))
        return buf.raw

    uid = exchange_unique_id(dist, rank, make_id)
    _lib.check(lib.slp_comm_init(int(world), int(rank), ctypes.create_string_buffer(uid, 128)))
