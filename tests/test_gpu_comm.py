"""RCCL path on one GPU: the communicator is created with a single rank and
SLP_FORCE_DISTRIBUTED=1 routes the solvers through the partitioned code path
(partial column sums -> all-reduce -> update; scalar all-reduces for dot
products and report terms).  With one rank the all-reduce is the identity, so the
results must equal the non-partitioned run.  Runs in a subprocess because the
communicator is process-wide.  -m gpu."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
os.environ["SLP_FORCE_DISTRIBUTED"] = "1"
os.environ["SLP_STRIP_MIN_NNZ"] = %(min_nnz)r
from pysparselp_amd import _lib
from pysparselp_amd.problems import random_lp_on_device
from pysparselp_amd.scale import DeviceBlocks, DeviceBlocksGroup, DeviceCP
from pysparselp_amd.admm_cg import DeviceADMM
lib = _lib.lib(0)
n, m, p = 30000, 40000, 0.001

def run():
    a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=3)
    cp = DeviceCP(a, b, c, lb, ub); cp.iterate(30); x_cp = cp.x(); cp.close()
    blk = DeviceBlocks(a, b, c, lb, ub, cg_max_steps=40); blk.iterate(4); x_blk = blk.x(); e_blk = blk.report()[0]; blk.close()
    # three row blocks on this rank: under the communicator every block's summand is all-reduced on the second stream while
    # the next block's projection computes (asynchronous block updates)
    grp = DeviceBlocksGroup(a, [0, 15000, 27000, m], b, c, lb, ub, cg_max_steps=40); grp.iterate(4); x_grp = grp.x(); grp.close()
    admm = DeviceADMM(a, b, c, lb, ub); admm.iterate(15); x_admm = admm.x(n); rep = admm.report(); admm.close()
    a.close()
    return x_cp, x_admm, rep, x_blk, e_blk, x_grp

plain = run()                                   # no communicator yet: single-GPU code path
uid = ctypes.create_string_buffer(128)
_lib.check(lib.slp_comm_unique_id(uid))
_lib.check(lib.slp_comm_init(1, 0, uid))
v = np.array([1.5, -2.0]); _lib.check(lib.slp_comm_allreduce_host(_lib.ptr(v), 2, 0)); assert np.array_equal(v, [1.5, -2.0])
_lib.check(lib.slp_comm_barrier())
part = run()                                    # partitioned code path, one rank
_lib.check(lib.slp_comm_finalize())
if os.environ["SLP_STRIP_MIN_NNZ"] == "1":      # strip kernels: sequential-order sums on both paths
    assert np.array_equal(plain[0], part[0]), np.max(np.abs(plain[0] - part[0]))
else:                                           # lane-parallel sums: the two paths unroll differently
    assert np.max(np.abs(plain[0] - part[0]) / (1 + np.abs(plain[0]))) < 1e-12
assert np.max(np.abs(plain[1] - part[1]) / (1 + np.abs(plain[1]))) < 1e-12
assert np.allclose(plain[2], part[2], rtol=1e-10)
assert np.max(np.abs(plain[3] - part[3]) / (1 + np.abs(plain[3]))) < 1e-12   # block-splitting ADMM, consensus all-reduce
assert abs(plain[4] - part[4]) <= 1e-10 * (1 + abs(plain[4]))
assert np.array_equal(plain[5], part[5])                                     # one rank: the per-block all-reduces are identities
print("COMM-OK")
"""


@pytest.mark.parametrize("min_nnz", ["1", "1000000000"])  # strip kernels / generic kernels
def test_partitioned_path_single_rank(min_nnz):
    code = SCRIPT % {"repo": REPO, "min_nnz": min_nnz}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "COMM-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------------------------------------------------------
# The exchange steps themselves, recorded: slp_comm_init_host routes every all-reduce of the partitioned device
# code through a host callback (here: one rank, the callback records (count, op) and leaves the buffer alone).
RECORD_SCRIPT = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
os.environ["SLP_FORCE_DISTRIBUTED"] = "1"
os.environ["SLP_STRIP_MIN_NNZ"] = "1"
from pysparselp_amd import _lib
from pysparselp_amd.problems import random_lp_on_device
from pysparselp_amd.scale import DeviceBlocks, DeviceCP
from pysparselp_amd.admm_cg import DeviceADMM
lib = _lib.lib(0)
calls = []
@_lib.HOST_ALLREDUCE_FN
def record(buf, count, op, user):
    calls.append((int(count), int(op)))
    return 0
_lib.check(lib.slp_comm_init_host(1, 0, record, None))
n, m, p = 30000, 40000, 0.001
a, xf, c, lb, ub, b = random_lp_on_device(n, m, p, seed=3)

def per_iteration(solver, warm, k):
    solver.iterate(warm)
    del calls[:]
    c0 = lib.slp_comm_collectives()
    solver.iterate(k)
    assert lib.slp_comm_collectives() - c0 == len(calls)
    return list(calls)

cp = DeviceCP(a, b, c, lb, ub)
got = per_iteration(cp, 2, 10)
assert got == [(n, 0)] * 10, got[:4]                      # Chambolle-Pock: ONE all-reduce of the n partial column sums
cp.close()

admm = DeviceADMM(a, b, c, lb, ub)                        # reuse level 4 (the default at scale)
got = per_iteration(admm, 3, 10)                          # iterations 4..13: no refresh inside
assert got == [(n + 5, 0), (n + 1, 0)] * 10, got[:6]      # TWO: A^T v1 (+5 packed scalars), A^T (A r) (+1); no scalar all-reduces
got = per_iteration(admm, 40, 20)                         # iterations 54..73: the refresh of iteration 64 falls inside
assert len(got) == 40 and got.count((2 * n + 5, 0)) == 1 and got.count((n + 5, 0)) == 19 and got.count((n + 1, 0)) == 20, got
admm.close()

blk = DeviceBlocks(a, b, c, lb, ub, cg_max_steps=30)
got = per_iteration(blk, 1, 3)
assert got == [(n, 0)] * 3, got                           # block splitting: ONE consensus all-reduce, none inside the block CG
blk.close()
from pysparselp_amd.scale import DeviceBlocksGroup
grp = DeviceBlocksGroup(a, [0, 10000, 25000, m], b, c, lb, ub, cg_max_steps=30)
got = per_iteration(grp, 1, 3)
assert got == [(n, 0)] * 9, got                           # three blocks on the rank: one (overlapped) all-reduce per block
grp.close()
a.close()
_lib.check(lib.slp_comm_finalize())
print("RECORD-OK")
"""


def test_collectives_per_iteration_are_counted_and_minimal():
    r = subprocess.run([sys.executable, "-c", RECORD_SCRIPT % {"repo": REPO}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RECORD-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
