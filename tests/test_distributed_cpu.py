"""The N > 1 path on CPU: world_size-2 gloo processes run the row-partitioned
algorithms exactly as the device code decomposes them (pysparselp_amd/parallel.py,
csrc/slp_cp.hip "distributed", csrc/slp_admm_cg.hip), with the ORACLE's kernels
standing in for the HIP kernels of each rank (test only), and must reproduce the
single-process oracle.  Also covers the partition helpers and the id exchange."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse

from oracle import oracle
from pysparselp_amd.parallel import rendezvous_unique_id, row_block, row_block_by_nnz


def test_row_block_covers_all_rows():
    for m, world in ((10, 1), (10, 3), (7, 8), (2_000_000, 8), (0, 2)):
        blocks = [row_block(m, world, r) for r in range(world)]
        assert blocks[0][0] == 0 and sum(c for _, c in blocks) == m
        for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
            assert f1 == f0 + c0 or c0 == 0 or f1 == m


def test_row_block_by_nnz_balances_ragged_rows():
    rng = np.random.RandomState(0)
    lens = np.concatenate((rng.randint(1, 4, 900), rng.randint(200, 400, 100)))
    indptr = np.concatenate(([0], np.cumsum(lens)))
    world = 4
    blocks = [row_block_by_nnz(indptr, world, r) for r in range(world)]
    assert blocks[0][0] == 0 and sum(c for _, c in blocks) == 1000
    nnz = [indptr[f + c] - indptr[f] for f, c in blocks]
    assert max(nnz) - min(nnz) <= 2 * 400  # every cut is within one long row of the ideal split


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_lp(seed=0, n=40, m=70):
    rng = np.random.RandomState(seed)
    a = scipy.sparse.random(m, n, density=0.2, random_state=rng, format="csr")
    a.data = np.round(rng.randn(a.nnz) * 100) / 100
    a.sort_indices()
    xf = np.round(rng.randn(n) * 100) / 100
    b = np.ceil((a @ xf + 0.01) * 1000) / 1000
    c = np.round(rng.randn(n) * 100) / 100
    t = np.round(rng.randn(n) * 100) / 100
    return a, b, c, xf + np.minimum(0, t), xf + np.maximum(0, t)


def _rank_main(rank, world, port, out):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allreduce(v, op=dist.ReduceOp.SUM):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        dist.all_reduce(t, op=op)
        return t.numpy()

    # --- id exchange (the product's control plane: plain TCP on MASTER_ADDR : MASTER_PORT + 1, no torch)
    uid = rendezvous_unique_id(rank, world, lambda: bytes(range(128)))
    assert uid == bytes(range(128))

    a, b, c, lb, ub = _small_lp()
    m, n = a.shape
    r0, rows = row_block(m, world, rank)
    ag = oracle.as_csr(a[r0:r0 + rows])
    bg = b[r0:r0 + rows]

    # --- Chambolle-Pock, rows partitioned (slp_cp.hip: k_cp_colsum -> all-reduce -> invert; k_cp_colsum_y ->
    #     all-reduce -> k_cp_primal<FROM_PRE>; k_cp_dual local)
    colsum = oracle.rmatvec(oracle.Csr(ag.indptr, ag.indices, np.abs(ag.data), ag.shape), np.ones(rows))
    tsum = allreduce(colsum)
    tsum[tsum == 0] = 1
    diag_t = 1 / tsum
    rs = oracle.matvec(oracle.Csr(ag.indptr, ag.indices, np.abs(ag.data), ag.shape), np.ones(n))
    rs[rs == 0] = 1
    sigma = 1 / rs
    x, y = np.zeros(n), np.zeros(rows)
    for _ in range(60):
        d = c + allreduce(oracle.rmatvec(ag, y))
        x2 = np.minimum(np.maximum(x - diag_t * d, lb), ub)
        z = 2 * x2 - x
        x = x2
        y = np.maximum(y + sigma * (oracle.matvec(ag, z) - bg), 0)
    out["cp_x"] = x

    # --- matrix-free ADMM, rows + their slack variables partitioned (slp_admm_cg.hip)
    inv1 = np.empty(rows)
    oracle._lib().orc_row_scale_l2(rows, oracle._p(ag.indptr), oracle._p(ag.data), oracle._p(inv1))
    rowid = np.repeat(np.arange(rows), np.diff(ag.indptr))
    a1 = oracle.Csr(ag.indptr, ag.indices, inv1[rowid] * ag.data, ag.shape)
    bu = inv1 * bg
    s2 = np.sqrt(np.add.reduceat(np.append(a1.data ** 2, 0.0), a1.indptr[:-1])[:rows] * (np.diff(a1.indptr) > 0) + 1.0)
    inv2 = 1 / s2
    a2 = oracle.Csr(a1.indptr, a1.indices, inv2[rowid] * a1.data, a1.shape)
    sc = -inv2
    ge, gi, alpha = 2.0, 3.0, 1.4
    xo, xs = np.zeros(n), np.zeros(rows)            # original (replicated) | slack (local)
    lbo, ubo, lbs, ubs = lb, ub, np.full(rows, -np.inf), bu
    xpo, xps = np.maximum(xo, 0), np.maximum(xs, 0)
    lam, lio, lis = np.zeros(rows), np.zeros(n), np.zeros(rows)
    diro, dirs = np.zeros(n), np.zeros(rows)
    co = c

    def a_apply(vo, vs):
        return oracle.matvec(a2, vo) + sc * vs

    def at_apply(w):
        return allreduce(oracle.rmatvec(a2, w)), sc * w

    def dot(ao, as_, bo, bs):
        return ao.dot(bo) + float(allreduce(np.array([as_.dot(bs)]))[0])

    def m_apply(vo, vs):
        uo, us = at_apply(a_apply(vo, vs))
        return ge * uo + gi * vo, ge * us + gi * vs

    for _ in range(30):
        uo, us = at_apply(lam)
        yo, ys = (-co + gi * xpo) - uo - lio, (gi * xps) - us - lis  # A^T b = 0 (b = 0 for slack form)
        xprev_o, xprev_s = xo.copy(), xs.copy()
        mo, ms = m_apply(xo, xs)
        t = -dot(diro, dirs, mo - yo, ms - ys)
        if abs(t) > 0:
            mdo, mds = m_apply(diro, dirs)
            step = t / dot(diro, dirs, mdo, mds)
            xo, xs = xo + step * diro, xs + step * dirs
        mo, ms = m_apply(xo, xs)
        ro, rs_ = yo - mo, ys - ms
        rsold = dot(ro, rs_, ro, rs_)
        apo, aps = m_apply(ro, rs_)
        a_cg = rsold / dot(ro, rs_, apo, aps)
        xo, xs = xo + a_cg * ro, xs + a_cg * rs_
        diro, dirs = xo - xprev_o, xs - xprev_s
        xo, xs = alpha * xo + (1 - alpha) * xpo, alpha * xs + (1 - alpha) * xps
        xpo = np.minimum(np.maximum(xo + lio / gi, lbo), ubo)
        xps = np.minimum(np.maximum(xs + lis / gi, lbs), ubs)
        lio, lis = lio + gi * (xo - xpo), lis + gi * (xs - xps)
        lam = lam + ge * (a_apply(xo, xs) - 0.0)
    out["admm_x"] = xo

    # --- the same ADMM in the form the device runs at scale (slp_admm_cg.hip, reuse level 4, rows partitioned): fused
    #     gradient g = M x - y, M dir and A dir carried by recurrences, and the PACKED exchange -- the rank-local slack parts
    #     of all dot products ride behind the two vector all-reduces (5 doubles behind A^T v1, 1 behind A^T (A r)); r.r over
    #     the slack unknowns is rebuilt from three reduced coefficients once the step is known.  Exactly two collectives
    #     per iteration, counted here.
    counter = {"n": 0}

    def allreduce_counted(v):
        counter["n"] += 1
        return allreduce(v)

    xo, xs = np.zeros(n), np.zeros(rows)
    xpo, xps = np.maximum(xo, 0), np.maximum(xs, 0)
    lam, lio, lis = np.zeros(rows), np.zeros(n), np.zeros(rows)
    diro, dirs, mdo, mds = np.zeros(n), np.zeros(rows), np.zeros(n), np.zeros(rows)
    qo, qs = -co, np.zeros(rows)                       # q = -c + g_eq A^T b, b = 0
    v1 = ge * a_apply(xo, xs) + lam
    for _ in range(30):
        gs = (sc * v1 + gi * xs) - ((qs + gi * xps) - lis)         # rank-local pre-pass over the slack unknowns
        tail = np.array([dirs.dot(gs), dirs.dot(mds), gs.dot(gs), gs.dot(mds), mds.dot(mds)])
        red = allreduce_counted(np.concatenate((oracle.rmatvec(a2, v1), tail)))     # collective 1: n + 5
        go = (red[:n] + gi * xo) - ((qo + gi * xpo) - lio)
        tl = red[n:]
        xprev_o, xprev_s = xo.copy(), xs.copy()
        t = -(diro.dot(go) + tl[0])
        step = 0.0
        if abs(t) > 0:
            step = t / (diro.dot(mdo) + tl[1])
            xo, xs = xo + step * diro, xs + step * dirs
        ro, rs_ = -(go + step * mdo), -(gs + step * mds)
        rsold = ro.dot(ro) + ((tl[2] + 2.0 * step * tl[3]) + (step * step) * tl[4])
        w = a_apply(ro, rs_)
        aps = ge * (sc * w) + gi * rs_
        red = allreduce_counted(np.concatenate((oracle.rmatvec(a2, w), [rs_.dot(aps)])))  # collective 2: n + 1
        apo = ge * red[:n] + gi * ro
        a_cg = rsold / (ro.dot(apo) + red[n])
        xo, xs = xo + a_cg * ro, xs + a_cg * rs_
        mdo, mds = step * mdo + a_cg * apo, step * mds + a_cg * aps
        diro, dirs = xo - xprev_o, xs - xprev_s
        xo, xs = alpha * xo + (1 - alpha) * xpo, alpha * xs + (1 - alpha) * xps
        xpo = np.minimum(np.maximum(xo + lio / gi, lbo), ubo)
        xps = np.minimum(np.maximum(xs + lis / gi, lbs), ubs)
        lio, lis = lio + gi * (xo - xpo), lis + gi * (xs - xps)
        wx = a_apply(xo, xs)
        lam = lam + ge * wx
        v1 = ge * wx + lam
    assert counter["n"] == 2 * 30
    out["admm_packed_x"] = xo

    # --- block-splitting ADMM, one block per rank (slp_blocks.hip, rb_iteration): the rank's rows with their slacks kept
    #     implicit; per-block projection local (here: a direct solve of S nu = rhs), consensus sum = the only all-reduce
    gam, alf = 0.7, 1.95
    ad = a[r0:r0 + rows].toarray()
    smat = ad @ ad.T + np.eye(rows)
    used = (np.abs(ad).sum(axis=0) > 0).astype(np.float64)
    copies = allreduce(used)
    xp, xps = np.minimum(np.maximum(np.zeros(n), lb), ub), np.minimum(np.zeros(rows), bg)
    lam_b, lams_b = np.zeros(n), np.zeros(rows)
    for _ in range(40):
        v, vs = xp - lam_b / gam, xps - lams_b / gam
        nu = np.linalg.solve(smat, ad @ v - vs)
        xb = alf * (v - ad.T @ nu) + (1 - alf) * xp
        xsb = alf * (vs + nu) + (1 - alf) * xps
        acc = allreduce(np.where(used > 0, xb + lam_b / gam, 0.0))
        xp = np.minimum(np.maximum((np.where(copies > 0, acc, xp) - c / gam) / np.maximum(copies, 1), lb), ub)
        xps = np.minimum(xsb + lams_b / gam, bg)
        lam_b = np.where(used > 0, lam_b + gam * (xb - xp), lam_b)
        lams_b = lams_b + gam * (xsb - xps)
    out["blocks_x"] = xp
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q):
    out = {}
    _rank_main(rank, world, port, out)
    q.put((rank, out))


@pytest.mark.timeout(300)
def test_row_partitioned_solvers_match_single_process_oracle():
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b, c, lb, ub = _small_lp()
    x_ref, _ = oracle.chambolle_pock_ppd(c, None, None, a, None, b, lb, ub, nb_max_iter=60, nb_iter_plot=10 ** 9)
    for r in range(world):
        assert np.max(np.abs(results[r]["cp_x"] - x_ref)) < 1e-12          # partial sums add in another order
    assert np.array_equal(results[0]["cp_x"], results[1]["cp_x"])          # replicas stay bit-identical
    x_ref = oracle.lp_admm_cg(c, None, None, a, None, b, lb, ub, nb_iter=29, nb_iter_plot=10 ** 9)
    for r in range(world):
        assert np.max(np.abs(results[r]["admm_x"] - x_ref) / (1 + np.abs(x_ref))) < 1e-9
    assert np.array_equal(results[0]["admm_x"], results[1]["admm_x"])
    for r in range(world):  # the packed two-collective form the device runs at scale
        assert np.max(np.abs(results[r]["admm_packed_x"] - x_ref) / (1 + np.abs(x_ref))) < 1e-9
    assert np.array_equal(results[0]["admm_packed_x"], results[1]["admm_packed_x"])
    # the reference's block-splitting ADMM with the two row halves as its blocks (sparse LU per block)
    m = a.shape[0]
    cuts = [row_block(m, world, r) for r in range(world)]
    x_ref = oracle.lp_admm_block_decomposition(c, None, None, a, None, b, lb, ub, nb_iter=39, nb_iter_plot=10 ** 9,
                                               blocks_eq=[], blocks_ineq=[(f, f + cnt - 1) for f, cnt in cuts])
    for r in range(world):
        assert np.max(np.abs(results[r]["blocks_x"] - x_ref) / (1 + np.abs(x_ref))) < 1e-9
    assert np.array_equal(results[0]["blocks_x"], results[1]["blocks_x"])


def _tcp_rank(rank, world, port, q):
    import ctypes

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from pysparselp_amd.parallel import HostTcpAllreduce

    t = HostTcpAllreduce(rank, world)
    buf = np.arange(5, dtype=np.float64) * (rank + 1) + 0.25 * rank
    big = np.full(300_000, float(rank + 1))                      # larger than a socket buffer: no deadlock in the exchange
    out = []
    for arr, op in ((buf.copy(), 0), (buf.copy(), 1), (big, 0)):
        rc = t._allreduce(arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), arr.size, op, None)
        assert rc == 0
        out.append(arr)
    t.close()
    q.put((rank, out))


@pytest.mark.timeout(120)
def test_host_tcp_transport_reduces_identically_on_every_rank():
    """parallel.HostTcpAllreduce (the transport behind SLP_COMM_TRANSPORT=host): rank-ordered sums through rank 0, the same
    bits on every rank, sum and max, buffers larger than a socket buffer."""
    import multiprocessing as mp

    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tcp_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    base = np.arange(5, dtype=np.float64)
    want_sum = sum(base * (r + 1) + 0.25 * r for r in range(world))
    want_max = np.maximum.reduce([base * (r + 1) + 0.25 * r for r in range(world)])
    for r in range(world):
        assert np.array_equal(res[r][0], want_sum) and np.array_equal(res[r][1], want_max)
        assert np.array_equal(res[r][2], np.full(300_000, 6.0))


def test_rendezvous_rejects_a_rank_of_another_job():
    """the hello and the reply carry a job token (hash of MASTER_ADDR / MASTER_PORT / world size / run id or
    SLP_JOB_TOKEN): a rank of ANOTHER job that scans the same candidate ports neither receives this job's id nor takes a
    rank's place; the job's own rank still gets through afterwards."""
    import threading

    from pysparselp_amd import parallel

    port = _free_port()
    got = {}

    def serve():
        os.environ["SLP_JOB_TOKEN"] = "job-A"   # (threads share the environment: the server's token is fixed at its start)
        got["server"] = parallel.rendezvous_unique_id(0, 2, lambda: bytes(range(128)), addr="127.0.0.1", port=port, timeout=30.0)

    tag_a = None
    os.environ["SLP_JOB_TOKEN"] = "job-A"
    try:
        tag_a = parallel.MAGIC + parallel.job_token(2)
        th = threading.Thread(target=serve)
        th.start()
        import socket
        import time

        time.sleep(0.3)
        # a foreign rank 1 (other token): rank 0 never answers it with the id
        foreign = parallel.MAGIC + b"\x00" * 8
        reply = b""
        for _ in range(50):
            try:
                with socket.create_connection(("127.0.0.1", port), timeout=2.0) as conn:
                    conn.settimeout(2.0)
                    conn.sendall(foreign + (1).to_bytes(4, "little"))
                    try:
                        reply = conn.recv(200)
                    except socket.timeout:
                        reply = b""
                break
            except OSError:
                time.sleep(0.1)
        assert reply == b"" and th.is_alive()      # connection closed without the id; the server keeps waiting for ITS rank 1
        uid = parallel.rendezvous_unique_id(1, 2, None, addr="127.0.0.1", port=port, timeout=30.0)
        th.join(timeout=30)
        assert uid == bytes(range(128)) == got["server"] and tag_a is not None
    finally:
        del os.environ["SLP_JOB_TOKEN"]


@pytest.mark.timeout(120)
def test_rendezvous_closes_its_handshake_and_answers_retries():
    """the lost-acknowledgement hole.  A client only returns once rank 0's FINAL byte tells it that its ``OK`` arrived;
    a client that sent ``OK`` but lost the connection before that byte connects again -- and rank 0, which has counted it as served
    and gone on to the communicator's initialisation, still answers (a background thread keeps the listener for a while)."""
    import socket
    import threading
    import time

    from pysparselp_amd import parallel

    port = _free_port()
    os.environ["SLP_JOB_TOKEN"] = "job-handshake"
    uid = bytes(reversed(range(128)))
    got = {}
    try:
        tag = parallel.MAGIC + parallel.job_token(2)
        th = threading.Thread(target=lambda: got.setdefault("server", parallel.rendezvous_unique_id(0, 2, lambda: uid, addr="127.0.0.1", port=port,
                                                                                                    timeout=30.0)))
        th.start()

        def attempt(read_final):
            for _ in range(100):
                try:
                    with socket.create_connection(("127.0.0.1", port), timeout=2.0) as conn:
                        conn.settimeout(5.0)
                        conn.sendall(tag + (1).to_bytes(4, "little"))
                        reply = parallel._recv_exact(conn, len(tag) + 128)
                        conn.sendall(b"OK")
                        return reply, (parallel._recv_exact(conn, 1) if read_final else None)
                except OSError:
                    time.sleep(0.05)
            raise AssertionError("rank 0 did not answer")

        # the first connection acknowledges the id and drops before the final byte (as if that byte were lost)
        reply, _ = attempt(read_final=False)
        assert reply == tag + uid
        th.join(timeout=20.0)
        assert not th.is_alive() and got["server"] == uid        # rank 0 counted the peer as served and went on
        # the client's retry is still answered, final byte included
        reply, final = attempt(read_final=True)
        assert reply == tag + uid and final == b"K"
        # and the ordinary client path returns the id through the same listener
        assert parallel.rendezvous_unique_id(1, 2, None, addr="127.0.0.1", port=port, timeout=10.0) == uid
    finally:
        del os.environ["SLP_JOB_TOKEN"]


def test_host_transport_survives_a_connection_its_client_gave_up():
    """a client abandons a connection after 5 s without the tag echo and reconnects; rank 0, accepting strictly one
    after the other, may later pick up the ABANDONED socket (whose echo can still "succeed").  A connection only counts once
    the client has acknowledged the echo, so the dead one is dropped and the live retry is kept -- and rank 0 is not killed by
    a reset connection."""
    import ctypes
    import socket
    import threading

    from pysparselp_amd import parallel

    port = _free_port()
    os.environ["SLP_JOB_TOKEN"] = "abandoned-%d" % port
    try:
        tag = parallel.MAGIC + parallel.job_token(2)
        ends = {}

        def serve():
            ends[0] = parallel.HostTcpAllreduce(0, 2, addr="127.0.0.1", port=port, timeout=60.0)

        # the abandoned connection first: a valid hello of rank 1, then closed before rank 0 ever accepts it
        import time

        lst = threading.Thread(target=serve)
        lst.start()
        ghost = None
        for _ in range(100):
            try:
                ghost = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                break
            except OSError:
                time.sleep(0.05)
        assert ghost is not None
        ghost.sendall(tag + (1).to_bytes(4, "little"))
        ghost.close()
        ends[1] = parallel.HostTcpAllreduce(1, 2, addr="127.0.0.1", port=port, timeout=60.0)
        lst.join(timeout=60)
        assert 0 in ends and len(ends[0].peers) == 1
        out = {}

        def reduce(rank):
            arr = np.array([1.0 + rank, 10.0 * (rank + 1)])
            assert ends[rank]._allreduce(arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 2, 0, None) == 0
            out[rank] = arr

        th = threading.Thread(target=reduce, args=(0,))
        th.start()
        reduce(1)
        th.join(timeout=30)
        assert np.array_equal(out[0], [3.0, 30.0]) and np.array_equal(out[1], [3.0, 30.0])
        for e in ends.values():
            e.close()
    finally:
        del os.environ["SLP_JOB_TOKEN"]
