"""Kernel lab: HIP-event times of the SpMV kernels on BASELINE config 3 (both orientations; value-dictionary strips and,
with the dictionary ruled out, fp64 strips), for before / after comparisons of kernel changes.

    python tools/spmv_lab.py [reps]
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = _lib.lib(0)
a = DeviceMatrix.random(2_000_000, 1_000_000, 1e-3, 0)
out = {}
for policy, tag in ((0, "dict"), (1, "fp64")):
    a.set_format(policy)
    for t, name in ((False, "Ax"), (True, "ATy")):
        a.bench_spmv(t, reps=2)
        ms = min(a.bench_spmv(t, reps=reps) for _ in range(3))
        out[f"{tag}_{name}_ms"] = round(ms, 4)
        out[f"{tag}_{name}_kernel"] = a.spmv_kernel(t)
        out[f"{tag}_{name}_GBps"] = round((lib.slp_matrix_format_bytes(a._h, int(t)) + 24e6) / ms / 1e6, 1)
print(json.dumps(out))
