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


def comm_world():
    """``(world, rank)`` of the communicator inside libslp_hip.so; ``(1, 0)`` without one (or before the library is loaded)."""
    if _lib._lib is None:
        return 1, 0
    w, r = ctypes.c_int(1), ctypes.c_int(0)
    _lib._lib.slp_comm_info(ctypes.byref(w), ctypes.byref(r))
    return int(w.value), int(r.value)


def collective_elapsed(elapsed):
    """The elapsed time every rank must compare with ``max_time``: the MAX over the ranks (one small all-reduce) under a
    communicator, ``elapsed`` itself without one.  The host-API solvers are collective under a communicator (all-reduces inside
    the x-step, the report and the multiplier step); a stop decision taken from each rank's own clock would let one rank leave
    the loop while the others wait for it in the next all-reduce forever -- or return different iterates."""
    if comm_world()[0] <= 1:
        return elapsed
    import numpy as np

    v = np.array([float(elapsed)])
    _lib.check(_lib.lib().slp_comm_allreduce_host(_lib.ptr(v), 1, 1))
    return float(v[0])


def local_rows(indptr, m_eq=0):
    """This rank's share of the stacked constraint rows of a host LP under the active communicator: ``(r0, r1, m_eq_local)`` --
    rows ``r0 .. r1`` (equal stored entries per rank, ``row_block_by_nnz``), of which the first ``m_eq_local`` are equalities
    (the equality rows come first in the stack).  ``(0, m, m_eq)`` without a communicator."""
    world, rank = comm_world()
    m = len(indptr) - 1
    if world <= 1:
        return 0, m, m_eq
    r0, rows = row_block_by_nnz(indptr, world, rank)
    return r0, r0 + rows, max(0, min(m_eq, r0 + rows) - r0)


def _recv_exact(conn, nbytes):
    buf = b""
    while len(buf) < nbytes:
        chunk = conn.recv(nbytes - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return buf


MAGIC = b"SLPRDZV2"
PORT_TRIES = 8


def job_token(world):
    """8 bytes that identify THIS job in the rendezvous hello and reply: two jobs on one host whose candidate ports overlap
    (or a stale rank-0 server of a killed run) then reject each other instead of cross-connecting.  Derived from what
    every rank of a job shares: MASTER_ADDR / MASTER_PORT, the world size and the launcher's run id (or SLP_JOB_TOKEN)."""
    import hashlib

    parts = [os.environ.get("SLP_JOB_TOKEN", ""), os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29511"),
             str(int(world)), os.environ.get("TORCHELASTIC_RUN_ID", "")]
    return hashlib.sha256("|".join(parts).encode()).digest()[:8]


def rendezvous_port():
    """The launcher's own store owns MASTER_PORT (torch.distributed.run keeps a TCPStore there), so the id exchange
    starts at the next port; SLP_RDZV_PORT overrides."""
    if "SLP_RDZV_PORT" in os.environ:
        return int(os.environ["SLP_RDZV_PORT"])
    return int(os.environ.get("MASTER_PORT", "29511")) + 1


def rendezvous_unique_id(rank, world, make_id, addr=None, port=None, timeout=300.0):
    """Rank 0 creates the 128-byte RCCL unique id (``make_id()``) and serves it to the other ``world - 1`` ranks over TCP
    on ``addr`` (default MASTER_ADDR), on the first free port of ``port .. port + 7`` (default MASTER_PORT + 1 ...); every
    rank returns the same 128 bytes.  A client sends a magic word and its rank and expects the magic word back, so a
    foreign listener on one of the candidate ports is skipped and a stray connection cannot take a rank's place."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = rendezvous_port() if port is None else int(port)
    hello_tag = MAGIC + job_token(world)
    if rank == 0:
        uid = make_id()
        assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
        if world == 1:
            return bytes(uid)
        srv = None
        for p in range(port, port + PORT_TRIES):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, p))
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise OSError(f"rendezvous: no free port in {port}..{port + PORT_TRIES - 1} on {addr}")
        def answer(conn, served):
            """One connection: hello -> id -> OK -> final byte.  The peer counts as served once it has acknowledged the id; the
            final byte tells the CLIENT that rank 0 has seen its acknowledgement -- a client that misses it connects again."""
            with conn:
                conn.settimeout(10.0)
                try:
                    hello = _recv_exact(conn, len(hello_tag) + 4)
                    peer = int.from_bytes(hello[len(hello_tag):], "little")
                    if hello[:len(hello_tag)] != hello_tag or not 0 < peer < world:
                        return
                    # a client that gave this connection up meanwhile (it retries on a new one) leaves a dead socket whose
                    # sendall may well "succeed": only the acknowledgement counts -- and its retry is answered too
                    conn.sendall(hello_tag + bytes(uid))
                    if _recv_exact(conn, 2) == b"OK":
                        served.add(peer)
                        conn.sendall(b"K")
                except (ConnectionError, socket.timeout, OSError):
                    return

        srv.listen(world)
        srv.settimeout(timeout)
        served = set()
        try:
            while len(served) < world - 1:
                conn, _ = srv.accept()
                answer(conn, served)
        except socket.timeout:
            srv.close()
            # the other ranks are about to sit in ncclCommInitRank waiting for this one: end the job loudly instead
            raise TimeoutError(f"rendezvous: only ranks {sorted(served)} of 1..{world - 1} fetched the unique id within {timeout} s")
        except BaseException:
            srv.close()
            raise

        # A peer whose acknowledgement arrived but whose final byte got lost connects again: keep answering for a while in the
        # background (the id is the same); rank 0 itself goes on to the communicator's initialisation at once.
        def linger():
            srv.settimeout(1.0)
            end = time.monotonic() + 30.0
            with srv:
                while time.monotonic() < end:
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue
                    except OSError:
                        return
                    answer(conn, set())

        import threading

        threading.Thread(target=linger, name="slp-rendezvous-linger", daemon=True).start()
        return bytes(uid)
    deadline = time.monotonic() + timeout
    while True:
        for p in range(port, port + PORT_TRIES):
            try:
                with socket.create_connection((addr, p), timeout=5.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(hello_tag + int(rank).to_bytes(4, "little"))
                    reply = _recv_exact(conn, len(hello_tag) + 128)
                    if reply[:len(hello_tag)] == hello_tag:
                        conn.sendall(b"OK")
                        if _recv_exact(conn, 1) == b"K":   # rank 0 has seen the acknowledgement (else: once more, it keeps answering)
                            return reply[len(hello_tag):]
            except (ConnectionError, socket.timeout, OSError):
                pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rendezvous: rank 0 did not answer on {addr}:{port}..{port + PORT_TRIES - 1}")
        time.sleep(0.05)


class HostTcpAllreduce:
    """All-reduce over TCP through rank 0 for ``slp_comm_init_host`` (SLP_COMM_TRANSPORT=host): every rank sends its buffer to
    rank 0, which adds them in rank order and sends the result back -- the same bits on every rank.  A debugging / test
    transport (several ranks of the partitioned device code on ONE GPU, where RCCL refuses to run); never the fast path."""

    def __init__(self, rank, world, addr=None, port=None, timeout=300.0):
        import numpy as np

        self._np = np
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = rendezvous_port() + PORT_TRIES if port is None else int(port)
        self.peers = {}
        if self.world == 1:
            return
        hello_tag = MAGIC + job_token(self.world)
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            # Connections are taken one after the other.  A client gives a connection up after 5 s without the tag echo and
            # reconnects, so a socket accepted late may already be dead -- and a sendall into it may still "succeed": a
            # connection is only kept once the client has ACKNOWLEDGED the echo (short timeouts on both reads); a newer
            # connection from a peer replaces the stored one.
            while len(self.peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.settimeout(5.0)
                try:
                    hello = _recv_exact(conn, len(hello_tag) + 4)
                    peer = int.from_bytes(hello[len(hello_tag):], "little")
                    if hello[:len(hello_tag)] != hello_tag or not 0 < peer < self.world:
                        raise ConnectionError("not a rank of this job")
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.sendall(hello_tag)  # the client checks that it reached ITS job's rank 0
                    if _recv_exact(conn, 2) != b"OK":
                        raise ConnectionError("no acknowledgement")
                    conn.sendall(b"K")       # the client only keeps a connection whose acknowledgement rank 0 has seen
                except (ConnectionError, socket.timeout, OSError):
                    conn.close()
                    continue
                if peer in self.peers:
                    self.peers[peer].close()
                conn.settimeout(timeout)
                self.peers[peer] = conn
            srv.close()
        else:
            # A listener of ANOTHER job on this port (or a stale rank 0) is treated like a refused connection: close, wait, try
            # again until the deadline -- as rendezvous_unique_id does -- with a short timeout on the tag read, so that a
            # listener that never answers does not hold this rank for the full timeout.
            deadline = time.monotonic() + timeout
            while True:
                conn = None
                try:
                    conn = socket.create_connection((addr, port), timeout=5.0)
                    conn.settimeout(5.0)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.sendall(hello_tag + self.rank.to_bytes(4, "little"))
                    if _recv_exact(conn, len(hello_tag)) == hello_tag:
                        conn.sendall(b"OK")
                        if _recv_exact(conn, 1) == b"K":
                            break
                except (OSError, ConnectionError, socket.timeout):
                    pass
                if conn is not None:
                    conn.close()
                if time.monotonic() > deadline:
                    raise ConnectionError("host transport: no rank 0 of this job answered on %s:%d" % (addr, port))
                time.sleep(0.05)
            conn.settimeout(timeout)
            self.peers[0] = conn
        self.callback = _lib.HOST_ALLREDUCE_FN(self._allreduce)

    def _allreduce(self, buf, count, op, user):
        np = self._np
        try:
            mine = np.ctypeslib.as_array(buf, shape=(count,))
            if self.rank == 0:
                acc = mine.copy()
                for peer in sorted(self.peers):
                    other = np.frombuffer(_recv_exact(self.peers[peer], 8 * count), dtype=np.float64)
                    acc = np.maximum(acc, other) if op == 1 else acc + other
                data = acc.tobytes()
                for peer in sorted(self.peers):
                    self.peers[peer].sendall(data)
                mine[:] = acc
            else:
                self.peers[0].sendall(mine.tobytes())
                mine[:] = np.frombuffer(_recv_exact(self.peers[0], 8 * count), dtype=np.float64)
            return 0
        except Exception:  # noqa: BLE001 -- the C side turns a non-zero return into an SlpError
            return 1

    def close(self):
        for c in self.peers.values():
            c.close()
        self.peers = {}


_host_transport = None


def init_comm_from_env(rank=None, world=None):
    """Create the communicator inside libslp_hip.so for this process's GPU; rank / world size default to the launcher's
    RANK / WORLD_SIZE.  RCCL unless SLP_COMM_TRANSPORT=host (``HostTcpAllreduce``)."""
    global _host_transport
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL's intra-node transport needs on this platform
    lib = _lib.lib()
    if os.environ.get("SLP_COMM_TRANSPORT") == "host":
        _host_transport = HostTcpAllreduce(rank, world)
        if world == 1:
            _host_transport.callback = _lib.HOST_ALLREDUCE_FN(lambda buf, count, op, user: 0)
        _lib.check(lib.slp_comm_init_host(world, rank, _host_transport.callback, None))
        return

    def make_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.slp_comm_unique_id(buf))
        return buf.raw

    uid = rendezvous_unique_id(rank, world, make_id)
    _lib.check(lib.slp_comm_init(world, rank, ctypes.create_string_buffer(uid, 128)))
