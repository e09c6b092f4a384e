"""Compute side of ONE rank of N = 8 on one GPU, replicated updates against sharded updates (SLP_SHARD_UPDATES=1): the process
poses as rank 0 of 8 through the host transport with the exchange switched off (SLP_COMM_NULL=1: collectives return at once --
WRONG iterates, right kernel sequence and sizes), holds rank 0's row block of the LP and times ADMM steps.  What the sharding of
the elementwise passes saves per step, before any exchange cost (the sharded form issues 6 collectives per iteration instead of 2:
that side needs an 8-GPU node).
    python tools/shard_compute_only.py [--config c4|c3] [--ranks 8] > profiles/rNN_shard_compute_only.json"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"c4": (10_000_000, 20_000_000, 1e-4), "c3": (1_000_000, 2_000_000, 1e-3)}


def child(config, ranks, steps):
    sys.path.insert(0, REPO)
    import numpy as np

    from pysparselp_amd import _lib
    from pysparselp_amd.admm_cg import DeviceADMM
    from pysparselp_amd.parallel import row_block
    from pysparselp_amd.problems import random_lp_on_device

    lib = _lib.lib(0)
    cb = _lib.HOST_ALLREDUCE_FN(lambda buf, count, op, user: 0)
    _lib.check(lib.slp_comm_init_host(ranks, 0, cb, None))
    n, m, dens = SHAPES[config]
    r0, rows = row_block(m, ranks, 0)
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, dens, seed=0, row_offset=r0, rows=rows)
    s = DeviceADMM(a, b, c, lb, ub)
    s.iterate(3)
    _lib.check(lib.slp_synchronize())
    c0 = int(lib.slp_comm_collectives())
    t0 = time.perf_counter()
    s.iterate(steps)
    _lib.check(lib.slp_synchronize())
    dt = time.perf_counter() - t0
    out = {"ms_per_step": 1e3 * dt / steps, "collectives_per_step": (int(lib.slp_comm_collectives()) - c0) / steps, "rows": rows}
    s.close()
    a.close()
    print(json.dumps(out))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="c4", choices=sorted(SHAPES))
    p.add_argument("--ranks", type=int, default=8)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--child", action="store_true")
    args = p.parse_args()
    if args.child:
        child(args.config, args.ranks, args.steps)
        return
    res = {"what": __doc__.split("\n    python")[0], "config": args.config, "ranks": args.ranks}
    for name, shard in (("replicated_updates", "0"), ("sharded_updates", "1"), ("replicated_updates_again", "0"), ("sharded_updates_again", "1")):
        env = dict(os.environ, SLP_COMM_NULL="1", SLP_SHARD_UPDATES=shard)
        out = subprocess.run([sys.executable, __file__, "--child", "--config", args.config, "--ranks", str(args.ranks), "--steps", str(args.steps)],
                             capture_output=True, text=True, check=True, env=env).stdout.strip().splitlines()[-1]
        res[name] = json.loads(out)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
