"""BASELINE config 5 as a WHOLE LP: block-splitting ADMM (ADMMBlocks.py:178-243 per-block solve, :264-307
loop, :290-299 consensus average) on 5e7 variables x eight row blocks of 5e5 rows at density 1e-4 -- 2e10 stored entries, resident
on ONE GPU because every block is generated from its own row range, converted into its tall-cell copies and its CSR released
before the next one is generated (``DeviceBlocksGroup.from_generator``: no resident matrix to cut the blocks out of).

Full size (needs ~295 GB of free device memory; skips with the reason below that):
  * every block's products against the oracle on regenerated row slices: ``A^T y`` for a slice-supported ``y`` bit for bit;
    ``A x`` to 5e-12 (the strip-range split ``bench.py --method admm_blocks`` switches on adds the partial sums of a 5000-entry
    row in range order -- re-association only, tests/test_gpu_c5_shape.py pins split against unsplit against the oracle);
  * after EVERY block update of three iterations the projection's TRUE residual ``|| rhs - S nu ||`` within 10 x the
    conjugate-gradient bar (in this dual form that is the constraint residual ``A z - z_s`` of the projected point);
  * two runs from the same start bit-identical (x and CG step counts).
Reduced shape (8 blocks of 6000 rows x 2e5 variables, tall cells in both orientations), the partition the driver's runs use:
  * 8 blocks on 1 rank = 2 ranks x 4 blocks = 8 ranks x 1 block through the host transport (asynchronous form where a rank
    holds several blocks): replicas bit-identical, x equal to 1e-9, G all-reduces per iteration;
  * the one-rank group against the oracle's matrix-free form (``oracle.lp_admm_blocks_cg``, pinned against the reference's LU
    iterates in tests/test_admm_blocks.py) to 1e-9.
-m gpu."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M, P, SEED, G = 50_000_000, 4_000_000, 1e-4, 0, 8
RB = M // G


def _free_gb():
    from pysparselp_amd import _lib

    lib = _lib.lib()
    free, total = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
    _lib.check(lib.slp_device_memory(_lib.ptr(free), _lib.ptr(total)))
    return float(free[0]) / 1e9


@pytest.fixture(scope="module")
def c5():
    from pysparselp_amd import _lib
    from pysparselp_amd.scale import DeviceBlocksGroup

    _lib.check(_lib.lib().slp_trim())
    if _free_gb() < 295:
        pytest.skip("less than 295 GB of device memory free: BASELINE config 5 needs the whole GPU")
    os.environ["SLP_TALL_SPLIT"] = "-1"   # what bench.py --method admm_blocks runs with
    try:
        grp, xf, c, lb, ub, b = DeviceBlocksGroup.from_generator(N, M, P, SEED, [RB * g for g in range(G + 1)])
    finally:
        os.environ.pop("SLP_TALL_SPLIT", None)
    yield grp, xf, c, lb, ub, b
    grp.close()


@pytest.mark.timeout(1800)
def test_every_block_is_resident_and_multiplies_like_the_oracle(c5):
    from pysparselp_amd.device import DeviceMatrix

    grp, xf, c, lb, ub, b = c5
    assert len(grp.blocks) == G and 1.99e10 < grp.nnz < 2.01e10
    rng = np.random.RandomState(11)
    x = rng.randn(N)
    rows = 384
    for g in range(G):
        mat = grp._mats[g]
        assert mat.shape == (RB, N) and mat.chunks == 1 and mat.spmv_kernel(False) == 6 and mat.spmv_kernel(True) == 6
        ax = mat.matvec(x)
        starts = (0, (RB // 2 + 977 * g) & ~1, RB - rows)     # first rows, a block-dependent middle, last rows
        ysub, slices, picks = np.zeros(RB), [], []
        y = rng.randn(RB)
        for r0 in starts:
            sl = DeviceMatrix.random(rows, N, P, SEED, RB * g + r0)   # the same rows again from the counter-based generator
            host = sl.download()
            sl.close()
            want = oracle.matvec(oracle.as_csr(host), x)
            assert float(np.max(np.abs(ax[r0:r0 + rows] - want) / (1 + np.abs(want)))) <= 5e-12, (g, r0)
            # b_upper of the block = ceil((A x_f + ...) 1000) / 1000 >= A x_f: the generator's own product over these rows
            assert np.all(host @ xf <= b[RB * g + r0:RB * g + r0 + rows] + 1e-9)
            slices.append(host)
            picks.append(y[r0:r0 + rows])
            ysub[r0:r0 + rows] = y[r0:r0 + rows]
        want = oracle.rmatvec(oracle.as_csr(scipy.sparse.vstack(slices, format="csr")), np.concatenate(picks))
        assert np.array_equal(mat.rmatvec(ysub), want), g          # csc_matvec order, zero terms left out: bit for bit
    # the adjoint identity on one block with dense vectors
    mat = grp._mats[G - 1]
    y = rng.randn(RB)
    ax, aty = mat.matvec(x), mat.rmatvec(y)
    lhs, rhs = float(ax.dot(y)), float(x.dot(aty))
    assert abs(lhs - rhs) <= 1e-10 * (abs(lhs) + np.linalg.norm(ax) * np.linalg.norm(y))


@pytest.mark.timeout(2400)
def test_block_solver_properties_on_the_whole_lp_not_an_oracle_comparison(c5):
    """Properties only (projection residuals, determinism, bounds): the solver against the ORACLE at this size is
    tools/c5_oracle_parity.py (profiles/r06_c5_oracle_parity.json), and on the reduced shape below."""
    grp, xf, c, lb, ub, b = c5
    runs = []
    for rep in range(2):
        grp.restart()
        for it in range(3):
            grp.iterate(1)
            for g, (res, rhs) in enumerate(grp.projection_residuals()):
                assert rhs > 0 and res <= 1e-12 * rhs, (rep, it, g, res, rhs)   # the CG bar is 1e-13 on the recurrence
        runs.append((grp.x(), grp.cg_steps()))
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    # once more with the blocks' projections side by side on streams of their own (round 6, opt-in: a run's first iteration goes
    # block after block, the others side by side)
    os.environ["SLP_BLOCKS_STREAMS"] = "1"
    try:
        grp.restart()
        grp.iterate(3)
    finally:
        os.environ.pop("SLP_BLOCKS_STREAMS", None)
    assert np.array_equal(runs[0][0], grp.x()) and runs[0][1] == grp.cg_steps()
    x = runs[0][0]
    assert np.all(np.isfinite(x)) and np.all(x >= lb - 1e-12) and np.all(x <= ub + 1e-12)
    assert runs[0][1] >= 3 * G * 10    # every block ran its conjugate gradients


# ---- the partition of the driver's runs on a reduced shape, through the host transport ---------------------------------
RN, RROWS, RP, RSEED, RIT = 200_000, 6000, 1.5e-4, 5, 6


def _rank(rank, world, port, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                           "SLP_COMM_TRANSPORT": "host", "SLP_JOB_TOKEN": "c5-%d" % port, "SLP_STRIP_MIN_NNZ": "1"})
        from pysparselp_amd import _lib
        from pysparselp_amd.parallel import init_comm_from_env, row_block
        from pysparselp_amd.scale import DeviceBlocksGroup

        lib = _lib.lib(0)
        if world > 1:
            init_comm_from_env(rank, world)
        m = G * RROWS
        r0, rows = row_block(m, world, rank)
        per = G // world
        grp, xf, c, lb, ub, b = DeviceBlocksGroup.from_generator(RN, m, RP, RSEED, [RROWS * g for g in range(per + 1)], row_offset=r0)
        assert all(mat.spmv_kernel(False) == 6 and mat.spmv_kernel(True) == 6 for mat in grp._mats)   # tall cells
        c0 = int(lib.slp_comm_collectives())
        _lib.check(lib.slp_comm_timing(1))
        grp.iterate(RIT)
        _lib.check(lib.slp_comm_timing(0))
        t = np.zeros(6)
        _lib.check(lib.slp_comm_timing_read(_lib.ptr(t)))
        out = {"x": grp.x(), "collectives": int(lib.slp_comm_collectives()) - c0, "cg_steps": grp.cg_steps(), "timed": t,
               "residuals": grp.projection_residuals()}
        grp.close()
        if world > 1:
            _lib.check(lib.slp_comm_finalize())
        q.put((rank, out))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


def _run(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
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


@pytest.mark.timeout(2400)
def test_eight_blocks_on_one_rank_equal_two_by_four_and_eight_by_one():
    from pysparselp_amd.device import DeviceMatrix

    one = _run(1)[0]
    assert one["collectives"] == 0 and np.all(np.isfinite(one["x"]))
    for res, rhs in one["residuals"]:
        assert rhs > 0 and res <= 1e-12 * rhs
    for world in (2, 8):
        got = _run(world)
        per = G // world
        for r in range(world):
            assert got[r]["collectives"] == per * RIT                    # one all-reduce of n doubles per block and iteration
            assert got[r]["timed"][0] == per * RIT and got[r]["timed"][1] > 0 and got[r]["timed"][2] == per * RIT * 8.0 * RN
            assert np.array_equal(got[r]["x"], got[0]["x"]), (world, r)   # replicas never drift apart
        err = float(np.max(np.abs(got[0]["x"] - one["x"]) / (1 + np.abs(one["x"]))))
        assert err <= 1e-9, (world, err)
    # the one-rank group against the oracle's matrix-free form on the regenerated blocks
    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        blocks, lp = [], None
        for g in range(G):
            a = DeviceMatrix.random(RROWS, RN, RP, RSEED, RROWS * g)
            xf, c, lb, ub, b = a.random_lp_vectors(RP, RSEED, RROWS * g)
            blocks.append((oracle.as_csr(a.download()), None, b))
            a.close()
            lp = (c, lb, ub)
    finally:
        os.environ.pop("SLP_STRIP_MIN_NNZ", None)
    want, steps = oracle.lp_admm_blocks_cg(lp[0], blocks, lp[1], lp[2], nb_iter=RIT)
    err = float(np.max(np.abs(one["x"] - want) / (1 + np.abs(want))))
    assert err <= 1e-9, err


def test_blocks_side_by_side_on_streams_change_nothing():
    """Round 6: with SLP_BLOCKS_STREAMS=1 the blocks' projections run side by side from a group's second iteration
    on, every block on a stream of its own, taking their rounds of conjugate-gradient steps in turn (csrc/slp_blocks.hip
    blocks_project_side_by_side; opt-in: measured 2.3 % slower than block after block at config 5).  Every block's arithmetic is
    what it was: x, the step counts and the projection residuals bit for bit those of block after block."""
    from pysparselp_amd.scale import DeviceBlocksGroup

    os.environ["SLP_STRIP_MIN_NNZ"] = "1"
    try:
        grp, xf, c, lb, ub, b = DeviceBlocksGroup.from_generator(RN, G * RROWS, RP, RSEED, [RROWS * g for g in range(G + 1)])
        outs = []
        for mode in ("0", "1", "1"):
            os.environ["SLP_BLOCKS_STREAMS"] = mode
            grp.restart()
            grp.iterate(1)
            grp.iterate(RIT - 1)
            outs.append((grp.x(), grp.cg_steps(), np.array(grp.projection_residuals())))
        grp.close()
    finally:
        os.environ.pop("SLP_STRIP_MIN_NNZ", None)
        os.environ.pop("SLP_BLOCKS_STREAMS", None)
    assert outs[0][1] > 0
    for other in outs[1:]:
        assert np.array_equal(outs[0][0], other[0]) and outs[0][1] == other[1] and np.array_equal(outs[0][2], other[2])
