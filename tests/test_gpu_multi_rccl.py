"""N > 1 ranks of the row-partitioned solvers under RCCL, one process per GPU -- runs only on a box with at least two
GPUs (the driver's 8-GPU node; a one-GPU box skips).  N = min(device count, 8).

Asserts: the replicas of x are bit-identical on every rank; x equals the single-process run of the whole LP to the
summation-order tolerance (1e-9 per entry) and the objective to 1e-9 relative; the collectives per iteration are the
documented ones (Chambolle-Pock 1, matrix-free ADMM at reuse level 4: 2); prints the per-iteration time of every N.
No reference counterpart (the reference is single-process): DESIGN.md section 5.  -m gpu."""
import multiprocessing as mp
import os
import socket
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_VARS, M_ROWS, DENSITY, SEED, ITERS = 200_000, 400_000, 1e-3, 4, 70


def _device_count():
    sys.path.insert(0, REPO)
    from pysparselp_amd import _lib

    return int(_lib.load().slp_device_count())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                           "LOCAL_RANK": str(rank), "HSA_ENABLE_IPC_MODE_LEGACY": "0", "SLP_STRIP_MIN_NNZ": "1"})
        from pysparselp_amd import _lib
        from pysparselp_amd.admm_cg import DeviceADMM
        from pysparselp_amd.parallel import init_comm_from_env, row_block
        from pysparselp_amd.problems import random_lp_on_device
        from pysparselp_amd.scale import DeviceCP

        lib = _lib.lib(rank)
        if world > 1:
            init_comm_from_env(rank, world)
        r0, rows = row_block(M_ROWS, world, rank)
        a, xf, c, lb, ub, b = random_lp_on_device(N_VARS, M_ROWS, DENSITY, seed=SEED, row_offset=r0, rows=rows)
        out = {}
        for name, make in (("cp", lambda: DeviceCP(a, b, c, lb, ub)), ("admm", lambda: DeviceADMM(a, b, c, lb, ub))):
            s = make()
            s.iterate(3)
            _lib.check(lib.slp_comm_barrier())
            c0 = int(lib.slp_comm_collectives())
            t0 = time.perf_counter()
            s.iterate(ITERS - 3)
            _lib.check(lib.slp_comm_barrier())
            dt = time.perf_counter() - t0
            out[name + "_collectives_per_iteration"] = (int(lib.slp_comm_collectives()) - c0 - 1) / (ITERS - 3) if world > 1 else 0
            out[name + "_ms_per_iteration"] = 1e3 * dt / (ITERS - 3)
            out[name + "_x"] = s.x() if name == "cp" else s.x(N_VARS)
            out[name + "_objective"] = float(c.dot(out[name + "_x"]))
            s.close()
        a.close()
        if world > 1:
            _lib.check(lib.slp_comm_finalize())
        q.put((rank, out))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in range(world):
        rank, out = q.get(timeout=900)
        assert "error" not in out, out["error"]
        res[rank] = out
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    return res


@pytest.mark.timeout(1800)
def test_partitioned_solvers_under_rccl_on_all_gpus_of_the_box():
    count = _device_count()
    if count < 2:
        pytest.skip(f"{count} GPU visible: the multi-rank RCCL run needs at least 2")
    one = _run(1)[0]
    worlds = sorted({2, min(count, 4), min(count, 8)})
    for world in worlds:
        res = _run(world)
        for key in ("cp_x", "admm_x"):
            for r in range(1, world):
                assert np.array_equal(res[0][key], res[r][key]), (world, key, r)   # replicas never drift apart
            err = float(np.max(np.abs(res[0][key] - one[key]) / (1 + np.abs(one[key]))))
            assert err < 1e-9, (world, key, err)
        for name in ("cp", "admm"):
            assert abs(res[0][name + "_objective"] - one[name + "_objective"]) <= 1e-9 * abs(one[name + "_objective"]), (world, name)
        # the refresh iteration (every 64th) sends one vector more in the SAME number of collectives
        assert res[0]["cp_collectives_per_iteration"] == 1.0, res[0]["cp_collectives_per_iteration"]
        assert res[0]["admm_collectives_per_iteration"] == 2.0, res[0]["admm_collectives_per_iteration"]
        print(f"N={world}: CP {res[0]['cp_ms_per_iteration']:.3f} ms/it (N=1: {one['cp_ms_per_iteration']:.3f}), "
              f"ADMM {res[0]['admm_ms_per_iteration']:.3f} ms/it (N=1: {one['admm_ms_per_iteration']:.3f})")


@pytest.mark.timeout(1800)
def test_sharded_updates_under_rccl_on_all_gpus_of_the_box():
    """The sharded form of the replicated updates (SLP_SHARD_UPDATES=1: reduce-scatter / slice updates / all-gather through RCCL's own
    ncclReduceScatter / ncclAllGather, in place): same answer, 6 collectives per steady-state iteration, and -- the number no
    one-GPU box can give -- its time per iteration next to the replicated form's.  Opt-in (SLP_TEST_SHARDED_RCCL=1) besides needing
    two GPUs: the form is off by default in the product and has only ever run through the host transport."""
    count = _device_count()
    if count < 2 or os.environ.get("SLP_TEST_SHARDED_RCCL") != "1":
        pytest.skip("needs at least 2 GPUs and SLP_TEST_SHARDED_RCCL=1")
    one = _run(1)[0]
    for world in sorted({2, min(count, 4), min(count, 8)}):
        if N_VARS % world:
            continue
        rep = _run(world)
        os.environ["SLP_SHARD_UPDATES"] = "1"
        try:
            sh = _run(world)
        finally:
            del os.environ["SLP_SHARD_UPDATES"]
        for r in range(1, world):
            assert np.array_equal(sh[0]["admm_x"], sh[r]["admm_x"]), (world, r)
        err = float(np.max(np.abs(sh[0]["admm_x"] - one["admm_x"]) / (1 + np.abs(one["admm_x"]))))
        assert err < 1e-9, (world, err)
        assert 5.5 < sh[0]["admm_collectives_per_iteration"] <= 6.5, sh[0]["admm_collectives_per_iteration"]
        print(f"N={world}: ADMM sharded updates {sh[0]['admm_ms_per_iteration']:.3f} ms/it, replicated {rep[0]['admm_ms_per_iteration']:.3f}")
