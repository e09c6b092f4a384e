"""world_size = 2 runs of the REAL row-partitioned device code on ONE GPU: two processes bind the same device and
exchange through the host-callback transport (slp_comm_init_host; here a pipe between the two processes) -- RCCL
itself refuses two ranks on one device, everything above it (partial sums, packed scalars, sequence and sizes of the
collectives, unequal and EMPTY row blocks) is the product's code.  The two replicas must stay bit-identical and match
the single-process run of the whole problem to summation-order tolerance.  -m gpu."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, conn, n, m, p, seed, iters, q):
    try:
        sys.path.insert(0, REPO)
        os.environ["SLP_STRIP_MIN_NNZ"] = "1"  # strip kernels, value dictionary, deferred row scaling: the at-scale configuration
        from pysparselp_amd import _lib
        from pysparselp_amd.admm_cg import DeviceADMM
        from pysparselp_amd.parallel import row_block
        from pysparselp_amd.problems import random_lp_on_device
        from pysparselp_amd.scale import DeviceCP

        lib = _lib.lib(0)
        sizes = []

        @_lib.HOST_ALLREDUCE_FN
        def allreduce(buf, count, op, user):
            mine = np.ctypeslib.as_array(buf, shape=(count,))
            sizes.append(int(count))
            if world > 2:  # a star: rank 0 adds the ranks' terms in rank order and hands the result back (conn: list / one pipe)
                if rank == 0:
                    res = mine.copy()
                    for c in conn:
                        other = np.frombuffer(c.recv_bytes(), dtype=np.float64)
                        assert other.size == count, (other.size, count)
                        res = np.maximum(res, other) if op == 1 else res + other
                    for c in conn:
                        c.send_bytes(res.tobytes())
                else:
                    conn.send_bytes(mine.tobytes())
                    res = np.frombuffer(conn.recv_bytes(), dtype=np.float64)
                    assert res.size == count, (res.size, count)
                mine[:] = res
                return 0
            if rank == 0:  # one side sends first, the other receives first: no deadlock on large buffers
                conn.send_bytes(mine.tobytes())
                other = np.frombuffer(conn.recv_bytes(), dtype=np.float64)
                res = np.maximum(mine, other) if op == 1 else mine + other
            else:
                other = np.frombuffer(conn.recv_bytes(), dtype=np.float64)
                conn.send_bytes(mine.tobytes())
                res = np.maximum(other, mine) if op == 1 else other + mine  # rank 0's term first on both sides
            assert other.size == count, (other.size, count)  # the ranks disagree on a collective's size
            mine[:] = res
            return 0

        if world > 1:
            _lib.check(lib.slp_comm_init_host(world, rank, allreduce, None))
        r0, rows = row_block(m, world, rank)
        a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=seed, row_offset=r0, rows=rows)
        out = {"rows": rows}
        cp = DeviceCP(a, b, c, lb, ub)
        cp.iterate(iters)
        out["cp_x"] = cp.x()
        cp.close()
        admm = DeviceADMM(a, b, c, lb, ub)
        assert admm.reuse == 4
        admm.iterate(iters)
        out["admm_x"] = admm.x(n)
        out["admm_report"] = admm.report()[:3]
        admm.close()
        a.close()
        out["sizes"] = sizes
        if world > 1:
            _lib.check(lib.slp_comm_finalize())
        q.put((rank, out))
    except BaseException as e:  # noqa: BLE001 -- report instead of leaving the peer blocked on the pipe
        import traceback

        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


def _run(world, n, m, p, seed, iters):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    if world > 2:
        pipes = [ctx.Pipe(duplex=True) for _ in range(world - 1)]
        conns = [[a for a, _ in pipes]] + [b for _, b in pipes]
    else:
        ends = ctx.Pipe(duplex=True) if world == 2 else (None, None)
        conns = [ends[r] if world == 2 else None for r in range(world)]
    procs = [ctx.Process(target=_rank, args=(r, world, conns[r], n, m, p, seed, iters, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in range(world):
        rank, out = q.get(timeout=900)
        assert "error" not in out, out["error"]
        res[rank] = out
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return res


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n,m,iters,p", [(30000, 40001, 70, 0.001), (3000, 1, 12, 0.001), (200000, 24001, 70, 1.5e-4)])
def test_two_ranks_on_one_gpu_match_the_single_process_run(n, m, iters, p):
    """m = 40001: unequal row blocks (20001 / 20000), 70 iterations cross the level-4 refresh at iteration 64.
    m = 1: rank 1's row block is EMPTY -- it must issue exactly the same collectives over zero-sized products.
    n = 200000 at density 1.5e-4: both orientations of both row blocks run on tall cells (the 1e7-variable shape's format)."""
    seed = 3
    one = _run(1, n, m, p, seed, iters)[0]
    two = _run(2, n, m, p, seed, iters)
    assert two[0]["rows"] + two[1]["rows"] == m and (m > 1 or two[1]["rows"] == 0)
    assert two[0]["sizes"] == two[1]["sizes"] and len(two[0]["sizes"]) > 2 * iters
    for key in ("cp_x", "admm_x"):
        assert np.array_equal(two[0][key], two[1][key]), key  # replicas never drift apart
        ref = one[key]
        err = float(np.max(np.abs(two[0][key] - ref) / (1 + np.abs(ref))))
        assert err < 1e-9, (key, err)
    assert np.allclose(two[0]["admm_report"], two[1]["admm_report"], rtol=0, atol=0)
    assert np.allclose(two[0]["admm_report"], one["admm_report"], rtol=1e-8, atol=1e-9)


@pytest.mark.timeout(1500)
def test_eight_ranks_on_one_gpu_match_the_single_process_run():
    """The partition the driver's 8-GPU run uses, on one GPU through the host transport: m = 40003 rows in eight unequal blocks
    (5001 x 3 + 5000 x 5), strip kernels and value dictionary on every block, 70 iterations across the level-4 refresh: the
    eight replicas stay bit-identical, issue the same collectives, and match the single-process run."""
    n, m, iters, p, seed = 30000, 40003, 70, 0.001, 3
    one = _run(1, n, m, p, seed, iters)[0]
    eight = _run(8, n, m, p, seed, iters)
    assert sum(eight[r]["rows"] for r in range(8)) == m and len({eight[r]["rows"] for r in range(8)}) == 2
    for r in range(1, 8):
        assert eight[r]["sizes"] == eight[0]["sizes"]
        for key in ("cp_x", "admm_x"):
            assert np.array_equal(eight[0][key], eight[r][key]), (key, r)
        assert np.array_equal(eight[0]["admm_report"], eight[r]["admm_report"])
    for key in ("cp_x", "admm_x"):
        err = float(np.max(np.abs(eight[0][key] - one[key]) / (1 + np.abs(one[key]))))
        assert err < 1e-9, (key, err)
    assert np.allclose(eight[0]["admm_report"], one["admm_report"], rtol=1e-8, atol=1e-9)


def _rank_groups(rank, world, port, n, m, p, seed, iters, async_mode, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                           "SLP_COMM_TRANSPORT": "host", "SLP_JOB_TOKEN": "groups-%d" % port, "SLP_STRIP_MIN_NNZ": "1",
                           "SLP_HOST_ASYNC": async_mode})
        from pysparselp_amd import _lib
        from pysparselp_amd.parallel import init_comm_from_env, row_block
        from pysparselp_amd.problems import random_lp_on_device
        from pysparselp_amd.scale import DeviceBlocksGroup

        lib = _lib.lib(0)
        if world > 1:
            init_comm_from_env(rank, world)
        r0, rows = row_block(m, world, rank)
        a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=seed, row_offset=r0, rows=rows)
        cuts = [0, rows // 4, (3 * rows) // 5, rows]    # three uneven blocks on every rank
        grp = DeviceBlocksGroup(a, cuts, b, c, lb, ub, cg_max_steps=40)
        c0 = int(lib.slp_comm_collectives())
        grp.iterate(iters)
        out = {"x": grp.x(), "collectives": int(lib.slp_comm_collectives()) - c0, "cg_steps": grp.cg_steps()}
        grp.close()
        a.close()
        if world > 1:
            _lib.check(lib.slp_comm_finalize())
        q.put((rank, out))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


@pytest.mark.timeout(900)
def test_block_groups_overlap_their_exchanges_with_two_ranks_on_one_gpu():
    """Block-splitting ADMM with three row blocks per rank and the rows split over two ranks (ADMMBlocks.py:264-307 at scale):
    a block's consensus summand is all-reduced while the NEXT block's projection computes (comm_allreduce_dev_async /
    comm_join).  Through the host transport's asynchronous form (a worker thread) the ordering is
    exercised with two real ranks: 3 all-reduces of n doubles per iteration in block order on both ranks, the same x as the
    transport's synchronous form (bit for bit: same sums, same order) and as the single-process group to the CG tolerance."""
    import socket

    n, m, p, seed, iters = 20000, 30000, 0.001, 4, 5

    def run(world, async_mode):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rank_groups, args=(r, world, port, n, m, p, seed, iters, async_mode, q)) for r in range(world)]
        for pr in procs:
            pr.start()
        res = {}
        for _ in range(world):
            rank, out = q.get(timeout=600)
            assert "error" not in out, out["error"]
            res[rank] = out
        for pr in procs:
            pr.join(timeout=60)
            assert pr.exitcode == 0
        return res

    asy, syn = run(2, "1"), run(2, "0")
    for res in (asy, syn):
        assert res[0]["collectives"] == res[1]["collectives"] == 3 * iters
        assert np.array_equal(res[0]["x"], res[1]["x"])
    assert np.array_equal(asy[0]["x"], syn[0]["x"])
    one = run(1, "1")[0]   # the same three-plus-three blocks cannot be formed on one rank: compare with ITS three blocks' consensus
    assert one["collectives"] == 0 and np.all(np.isfinite(one["x"]))


def _rank_chunked(rank, world, port, n, m, p, seed, iters, chunks, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                           "SLP_COMM_TRANSPORT": "host", "SLP_JOB_TOKEN": "chunked-%d" % port, "SLP_STRIP_MIN_NNZ": "1"})
        from pysparselp_amd import _lib
        from pysparselp_amd.admm_cg import DeviceADMM
        from pysparselp_amd.parallel import init_comm_from_env, row_block
        from pysparselp_amd.problems import random_lp_on_device
        from pysparselp_amd.scale import DeviceCP

        lib = _lib.lib(0)
        if world > 1:
            init_comm_from_env(rank, world)
        r0, rows = row_block(m, world, rank)
        a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=seed, row_offset=r0, rows=rows, chunks=chunks)
        out = {"chunks": getattr(a, "chunks", 1)}
        cp = DeviceCP(a, b, c, lb, ub)
        cp.iterate(iters)
        out["cp_x"] = cp.x()
        cp.close()
        admm = DeviceADMM(a, b, c, lb, ub)
        admm.iterate(iters)
        out["admm_x"] = admm.x(n)
        admm.close()
        a.close()
        if world > 1:
            _lib.check(lib.slp_comm_finalize())
        q.put((rank, out))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


@pytest.mark.timeout(900)
def test_two_ranks_with_chunked_row_blocks_match_the_single_process_run():
    """What ``bench.py --gpus 2`` does on the metric's LP: every rank's row block is itself a ChunkedDeviceMatrix (here 3 chunks
    of a tall-cell shape).  Replicas bit-identical, equal to the single-process unchunked run to the all-reduce's re-association."""
    import socket

    n, m, p, seed, iters = 200000, 48000, 1.5e-4, 7, 24

    def run(world, chunks):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rank_chunked, args=(r, world, port, n, m, p, seed, iters, chunks, q)) for r in range(world)]
        for pr in procs:
            pr.start()
        res = {}
        for _ in range(world):
            rank, out = q.get(timeout=600)
            assert "error" not in out, out["error"]
            res[rank] = out
        for pr in procs:
            pr.join(timeout=60)
            assert pr.exitcode == 0
        return res

    one = run(1, 1)[0]
    two = run(2, 3)
    assert two[0]["chunks"] == two[1]["chunks"] == 3
    for key in ("cp_x", "admm_x"):
        assert np.array_equal(two[0][key], two[1][key]), key
        err = float(np.max(np.abs(two[0][key] - one[key]) / (1 + np.abs(one[key]))))
        assert err < 1e-9, (key, err)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_updates_match_the_replicated_ones(world, monkeypatch):
    """VERDICT r03 item 4a, built behind ``SLP_SHARD_UPDATES=1``: in the level-4 steady state the two all-reduces of the variable
    vector become reduce-scatter ... all-gather pairs and every rank runs the elementwise passes over ITS n / N slice of the
    original variables (slice-partial dot products in two small all-reduces: 6 collectives per iteration instead of 2).  Through the
    host transport with 2 and 8 real ranks on one GPU, 70 iterations across the level-4 refresh at iteration 64 (which runs
    replicated, on gathered state): replicas bit-identical, equal to the single-process run and to the replicated partitioned run to
    the re-association of the sums, the same report."""
    n, m, iters, p, seed = 30000, 40003, 70, 0.001, 3
    one = _run(1, n, m, p, seed, iters)[0]
    rep = _run(world, n, m, p, seed, iters)
    monkeypatch.setenv("SLP_SHARD_UPDATES", "1")
    sh = _run(world, n, m, p, seed, iters)
    for r in range(1, world):
        assert sh[r]["sizes"] == sh[0]["sizes"]
        assert np.array_equal(sh[0]["admm_x"], sh[r]["admm_x"]) and np.array_equal(sh[0]["cp_x"], sh[r]["cp_x"])
        assert np.array_equal(sh[0]["admm_report"], sh[r]["admm_report"])
    assert len(sh[0]["sizes"]) > len(rep[0]["sizes"]) + 3 * (iters - 8)   # ~4 more collectives per steady-state iteration
    for ref in (one, rep[0]):
        err = float(np.max(np.abs(sh[0]["admm_x"] - ref["admm_x"]) / (1 + np.abs(ref["admm_x"]))))
        assert err < 1e-9, err
        assert np.allclose(sh[0]["admm_report"], ref["admm_report"], rtol=1e-8, atol=1e-9)
