"""``SparseLP.solve`` under a communicator: the drop-in reaches the at-scale path.  Two processes share
ONE GPU through the host transport (``SLP_COMM_TRANSPORT=host``: RCCL refuses two ranks on one device); each holds the whole LP,
calls ``solve(method=...)`` exactly as a single process would, hands over only its block of constraint rows
(``parallel.local_rows``), and gets the same ``x`` and the same ten curve lists.  Reference boundary: SparseLP.py:990-1002
(``solve``), :1193-1208 / :1270-1288 (the solver calls).  Bar: replicas identical, equal to the single-process run to 1e-9
(the all-reduce re-associates the column sums).  Also ``xstep="auto"`` and ``xstep="cg"`` through ``solve`` on one process:
``cg`` against the oracle's restatement of the reference's use_cg branch.  -m gpu."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

from conftest import lp_from_golden, load_golden, solver_args

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURVES = ("itrn_curve", "pobj_curve", "dobj_curve", "max_violated_constraint", "max_violated_equality", "max_violated_inequality",
          "distance_to_ground_truth")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, fixture, method, xstep, nb_iter, q, max_time=None, slow_clock_rank=-1):
    try:
        if rank == slow_clock_rank:   # this rank's clock runs at a quarter of the speed: it would cross max_time four reports later
            import time

            real, t0 = time.perf_counter, time.perf_counter()
            time.perf_counter = lambda: t0 + 0.25 * (real() - t0)
        sys.path.insert(0, REPO)
        sys.path.insert(0, os.path.join(REPO, "tests"))
        os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                           "SLP_COMM_TRANSPORT": "host", "SLP_JOB_TOKEN": "solve-ranks-%d" % port})
        from conftest import lp_from_golden as from_golden, load_golden as load

        from pysparselp_amd import _lib
        from pysparselp_amd.SparseLP import SparseLP
        from pysparselp_amd.parallel import comm_world, init_comm_from_env

        lib = _lib.lib(0)
        if world > 1:
            init_comm_from_env(rank, world)
            assert comm_world() == (world, rank)
        d = load(fixture)
        lp = from_golden(d, SparseLP)
        gt = d["c"] * 0.0
        x, _ = lp.solve(method=method, nb_iter=nb_iter, nb_iter_plot=10, ground_truth=gt, ground_truth_indices=np.arange(gt.size),
                        max_time=max_time, **({"xstep": xstep} if method == "admm" else {}))
        out = {"x": np.asarray(x), "collectives": int(lib.slp_comm_collectives())}
        for name in CURVES:
            out[name] = np.asarray(getattr(lp, name), dtype=np.float64)
        if world > 1:
            _lib.check(lib.slp_comm_finalize())
        q.put((rank, out))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


def _run(world, fixture, method, xstep, nb_iter, max_time=None, slow_clock_rank=-1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, fixture, method, xstep, nb_iter, q, max_time, slow_clock_rank))
             for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, out = q.get(timeout=600)
        assert "error" not in out, out["error"]
        res[rank] = out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fixture,method,xstep", [("lp_sc105", "chambolle_pock_ppd", None), ("lp_sc105", "admm", "cg"),
                                                  ("lp_potts50", "chambolle_pock_ppd", None), ("lp_potts50", "admm", "auto")])
def test_solve_with_two_ranks_on_one_gpu_equals_the_single_process_run(fixture, method, xstep):
    nb_iter = 120
    one = _run(1, fixture, method, "cg" if xstep == "auto" else xstep, nb_iter)[0]   # (auto picks cg only under a communicator here)
    two = _run(2, fixture, method, xstep, nb_iter)
    assert two[0]["collectives"] == two[1]["collectives"] > nb_iter     # a partitioned run, not two replicas
    for name in ("x",) + CURVES:
        assert np.array_equal(two[0][name], two[1][name]), name           # the ranks return the same answer
        ref, got = one[name], two[0][name]
        assert ref.shape == got.shape and ref.size > 0, name
        err = float(np.max(np.abs(got - ref) / (1 + np.abs(ref))))
        assert err <= 1e-9, (name, err)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("method,xstep", [("chambolle_pock_ppd", None), ("admm", "cg")])
def test_max_time_stops_every_rank_at_the_same_report(method, xstep):
    """under a communicator the solvers are collective (all-reduces inside the x-step, the report, the multiplier
    step); ``max_time`` compared with each rank's OWN clock would let one rank leave the loop while the other waits for it in the
    next all-reduce forever.  The decision is the max over the ranks' clocks (``parallel.collective_elapsed``): with rank 1's
    clock running at a quarter of the speed both ranks stop at the same report and return the same x (the old form deadlocks
    here -- the test's timeout would catch it)."""
    two = _run(2, "lp_sc105", method, xstep, 10 ** 7, max_time=0.5, slow_clock_rank=1)
    assert np.array_equal(two[0]["itrn_curve"], two[1]["itrn_curve"]) and 1 <= two[0]["itrn_curve"].size < 10 ** 5
    assert np.array_equal(two[0]["x"], two[1]["x"]) and two[0]["collectives"] == two[1]["collectives"]


def test_solve_reaches_the_matrix_free_admm():
    """``solve(method="admm", xstep="cg")`` = the oracle's restatement of the reference's use_cg branch (ADMM.py:182-201) to
    1e-9 over the first iterations; ``xstep="auto"`` keeps the exact Gauss-Seidel sweep on a small LP (M has 1e4 entries) and
    switches once the estimate of nnz(M) passes the bound."""
    from oracle import oracle

    import pysparselp_amd.SparseLP as mod
    from pysparselp_amd.SparseLP import SparseLP

    d = load_golden("lp_sc105")
    args = solver_args(d)
    lp = lp_from_golden(d, SparseLP)
    x_cg, _ = lp.solve(method="admm", nb_iter=60, nb_iter_plot=20, xstep="cg")
    ref = oracle.lp_admm_cg(*args, nb_iter=60, nb_iter_plot=20)
    assert float(np.max(np.abs(x_cg - ref) / (1 + np.abs(ref)))) <= 1e-9
    x_gs, _ = lp_from_golden(d, SparseLP).solve(method="admm", nb_iter=60, nb_iter_plot=20)
    assert np.array_equal(x_gs, oracle.lp_admm(*args, nb_iter=60, nb_iter_plot=20))           # the default: the reference as shipped
    x_auto, _ = lp_from_golden(d, SparseLP).solve(method="admm", nb_iter=60, nb_iter_plot=20, xstep="auto")
    assert np.array_equal(x_auto, x_gs)
    old = mod.ADMM_AUTO_M_ENTRIES
    try:
        mod.ADMM_AUTO_M_ENTRIES = 10.0
        x_auto, _ = lp_from_golden(d, SparseLP).solve(method="admm", nb_iter=60, nb_iter_plot=20, xstep="auto")
        assert np.array_equal(x_auto, x_cg)
    finally:
        mod.ADMM_AUTO_M_ENTRIES = old
