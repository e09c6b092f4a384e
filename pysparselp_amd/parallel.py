"""Row partition of the constraint matrix over the GPUs of one node.

The reference is single-process; this layer is new.  Decomposition (DESIGN.md
"multi-GPU"): rank g owns a contiguous block of constraint rows K_g together
with everything indexed by rows (b_g, Sigma_g, y_g, lambda_g and, for ADMM, the
slack variables of those rows); everything indexed by original variables (x, z,
c, T, lb, ub) is replicated.  The only data-path exchange is the sum-all-reduce
of the n partial column sums ``K_g^T y_g`` (RCCL inside libslp_hip.so, on the
compute stream); dot products over [original | slack] unknowns add the
replicated part to the all-reduced slack part.

The control plane is the exchange of the 128-byte RCCL unique id and nothing
else: a plain TCP rendezvous on MASTER_ADDR (``rendezvous_unique_id``) -- no
PyTorch anywhere in the product path.
"""
import ctypes
import os
import socket
import time

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


def _recv_exact(conn, nbytes):
    buf = b""
    while len(buf) < nbytes:
        chunk = conn.recv(nbytes - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return buf


def rendezvous_port():
    """The launcher's own store owns MASTER_PORT (torch.distributed.run keeps a TCPStore there), so the id exchange
    uses the next port; SLP_RDZV_PORT overrides."""
    if "SLP_RDZV_PORT" in os.environ:
        return int(os.environ["SLP_RDZV_PORT"])
    return int(os.environ.get("MASTER_PORT", "29511")) + 1


def rendezvous_unique_id(rank, world, make_id, addr=None, port=None, timeout=300.0):
    """Rank 0 creates the 128-byte RCCL unique id (``make_id()``) and serves it to the other ``world - 1`` ranks over TCP
    on ``addr:port`` (default MASTER_ADDR : MASTER_PORT + 1); every rank returns the same 128 bytes.  Each client sends
    its rank first, so a stray connection cannot take a rank's place."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = rendezvous_port() if port is None else int(port)
    if rank == 0:
        uid = make_id()
        assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
        if world == 1:
            return bytes(uid)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            served = set()
            while len(served) < world - 1:
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(timeout)
                    peer = int.from_bytes(_recv_exact(conn, 4), "little")
                    if 0 < peer < world and peer not in served:
                        conn.sendall(bytes(uid))
                        served.add(peer)
        return bytes(uid)
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.settimeout(timeout)
                conn.sendall(int(rank).to_bytes(4, "little"))
                return _recv_exact(conn, 128)
        except (ConnectionRefusedError, ConnectionResetError, socket.timeout, OSError):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.05)


def init_comm_from_env(rank=None, world=None):
    """Create the RCCL communicator inside libslp_hip.so for this process's GPU; rank / world size default to the
    launcher's RANK / WORLD_SIZE."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    lib = _lib.lib()

    def make_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.slp_comm_unique_id(buf))
        return buf.raw

    uid = rendezvous_unique_id(rank, world, make_id)
    _lib.check(lib.slp_comm_init(world, rank, ctypes.create_string_buffer(uid, 128)))
