"""Profiling target: both SpMV orientations of the 2.5e6 x 1e7, density-1e-4 slice on tall cells, nothing else.
    python tools/tall_only.py [reps] [rows] [cols] [density]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2_500_000
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
dens = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
lib = _lib.lib(0)
a = DeviceMatrix.random(rows, cols, dens, 1)
out = {"nnz": a.nnz}
for t, name in ((False, "Ax"), (True, "ATy")):
    out[name + "_kernel"] = a.spmv_kernel(t)
    out[name + "_ms"] = a.bench_spmv(t, reps=reps)
    out[name + "_copy_bytes"] = int(lib.slp_matrix_format_bytes(a._h, int(t)))
if os.environ.get("TALL_ONLY_HASH"):   # the products themselves, for comparing two builds of the copy on one box
    import hashlib

    import numpy as np
    rng = np.random.RandomState(1)
    x, y = rng.randn(cols), rng.randn(rows)
    out["Ax_sha"] = hashlib.sha256(a.matvec(x).tobytes()).hexdigest()[:16]
    out["ATy_sha"] = hashlib.sha256(a.rmatvec(y).tobytes()).hexdigest()[:16]
print(json.dumps(out))
