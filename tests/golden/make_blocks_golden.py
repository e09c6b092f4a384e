"""Fixtures for the block-splitting ADMM (reference ADMMBlocks.py:45-352): iterates of the
reference's lp_admm_block_decomposition on the LPs of the other fixtures, plus the block structure the reference's
modelling layer records for them.  Build container only:

    python tests/golden/make_blocks_golden.py      -> tests/golden/admm_blocks.npz

Also requires the oracle's restatement (SuperLU through scipy, like the reference) to reproduce those iterates.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden  # noqa: E402

KEEP = [0, 1, 2, 3, 5, 10, 20, 50, 100, 200]


def main():
    make_golden.build_reference()
    make_golden.install_shims()
    from oracle import oracle
    from pysparselp.ADMMBlocks import lp_admm_block_decomposition
    from pysparselp import randomLP
    from pysparselp.examples.example_pott_segmentation import build_linear_program

    sink = io.StringIO()
    cases = {}
    for name in ("SC50A", "SC105"):
        cases[name.lower()] = make_golden.netlib_lp(name)[0]
    with contextlib.redirect_stdout(sink):
        cases["potts8"] = build_linear_program(8, 0.5, 500)[0]
        cases["potts50"] = build_linear_program(50, 0.5, 500)[0]
    for seed in (0, 1):
        np.random.seed(seed)
        cases[f"random{seed}"] = randomLP.generate_random_lp(nbvar=60, n_eq=10, n_ineq=80, sparsity=0.2)[0]
    out = {}
    for name, lp in cases.items():
        d = np.load(os.path.join(HERE, f"lp_{name}.npz"))
        assert np.array_equal(d["c"], lp.costsvector) and np.array_equal(d["Ai_data"], lp.a_inequalities.tocsr().data)
        a_eq = lp.a_equalities if lp.a_equalities.shape[0] > 0 else None
        beq = lp.b_equalities if a_eq is not None else None
        args = (lp.costsvector, a_eq, beq, lp.a_inequalities, lp.b_lower, lp.b_upper, lp.lower_bounds, lp.upper_bounds)
        blocks_eq = [] if a_eq is None else list(a_eq.blocks)
        blocks_ineq = list(lp.a_inequalities.blocks)
        with contextlib.redirect_stdout(sink):
            rec = make_golden.capture(lambda cb: lp_admm_block_decomposition(*args, nb_iter=200, nb_iter_plot=1, callback_func=cb,
                                                                             max_time=1e9), KEEP)
        orc = make_golden.capture(lambda cb: oracle.lp_admm_block_decomposition(*args, nb_iter=200, nb_iter_plot=1, callback_func=cb,
                                                                                blocks_eq=blocks_eq, blocks_ineq=blocks_ineq), KEEP)
        worst = max(np.max(np.abs(a - b) / (1 + np.abs(a))) for a, b in zip(rec["x"], orc["x"]))
        print(f"{name}: {len(blocks_eq)} + {len(blocks_ineq)} blocks, oracle vs reference max rel diff {worst:.2e}")
        assert rec["it"] == orc["it"] and worst < 1e-9
        out[f"{name}_blocks_eq"] = np.array(blocks_eq, dtype=np.int64).reshape(-1, 2)
        out[f"{name}_blocks_ineq"] = np.array(blocks_ineq, dtype=np.int64).reshape(-1, 2)
        out[f"{name}_it"] = np.array(rec["it"])
        out[f"{name}_x"] = np.array(rec["x"])
        out[f"{name}_e1"] = np.array(rec["e1"])
        out[f"{name}_oracle_vs_reference"] = np.array(worst)
    np.savez_compressed(os.path.join(HERE, "admm_blocks.npz"), **out)


if __name__ == "__main__":
    main()
