"""Kernel lab for the 1e7-variable shape: HIP-event times of both SpMV orientations on the 1/8 row slice of a
1e7 x 2e7, density-1e-4 LP (2.5e6 x 1e7, 2.5e9 stored entries), tall cells against the wide strips they replace
(SLP_TALL=0) and the CSR kernel, with the bit-for-bit comparison between them.

    python tools/tall_lab.py [reps] [rows] [cols] [density]
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2_500_000
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
dens = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
lib = _lib.lib(0)
out = {"rows": rows, "cols": cols, "density": dens}
rng = np.random.RandomState(0)
x, y = rng.randn(cols), rng.randn(rows)
res = {}
for tag, env in (("tall", {}), ("tall_fp64", {"SLP_VALUE_DICT": "0"}), ("wide", {"SLP_TALL": "0"})):
    os.environ.pop("SLP_TALL", None)
    os.environ.pop("SLP_VALUE_DICT", None)
    os.environ.update(env)
    if tag == "wide" and os.environ.get("TALL_LAB_WIDE", "1") != "1":
        res["wide"] = res["tall"]
        continue
    a = DeviceMatrix.random(rows, cols, dens, 1)
    out["nnz"] = a.nnz
    for t, name in ((False, "Ax"), (True, "ATy")):
        t0 = time.perf_counter()
        k = a.spmv_kernel(t)
        lib.slp_synchronize()
        out[f"{tag}_{name}_build_s"] = round(time.perf_counter() - t0, 3)
        ms = min(a.bench_spmv(t, reps=reps) for _ in range(2))
        nbytes = lib.slp_matrix_format_bytes(a._h, int(t)) + 8 * (rows + cols)
        out[f"{tag}_{name}_kernel"] = k
        out[f"{tag}_{name}_ms"] = round(ms, 4)
        out[f"{tag}_{name}_copy_GB"] = round(lib.slp_matrix_format_bytes(a._h, int(t)) / 1e9, 3)
        out[f"{tag}_{name}_frac_of_8TBps"] = round(nbytes / ms / 1e6 / 8000.0, 4)
    res[tag] = (a.matvec(x), a.rmatvec(y))
    if tag == "tall" and os.environ.get("TALL_LAB_CSR", "1") == "1":
        a.set_format(2)
        res["csr"] = (a.matvec(x, order=1), a.rmatvec(y, order=1))
    a.close()
out["tall_fp64_equals_tall_bitwise"] = bool(np.array_equal(res["tall"][0], res["tall_fp64"][0]) and np.array_equal(res["tall"][1], res["tall_fp64"][1]))
out["tall_equals_wide_bitwise"] = bool(np.array_equal(res["tall"][0], res["wide"][0]) and np.array_equal(res["tall"][1], res["wide"][1]))
if "csr" in res:
    out["tall_equals_csr_bitwise"] = bool(np.array_equal(res["tall"][0], res["csr"][0]) and np.array_equal(res["tall"][1], res["csr"][1]))
print(json.dumps(out))
