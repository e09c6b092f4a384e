"""Where does the strip SpMV start to pay?  Times A.x and A^T.y with the strip format forced on and off over
a range of sizes of the synthetic matrix (density 1e-3), and the host->device hand-over of a CSR matrix
(PCIe-inclusive cost of slp_matrix_create).  python tools/strip_threshold.py"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402

lib = _lib.lib()
rows = []
for n, m in ((20_000, 40_000), (50_000, 100_000), (100_000, 200_000), (200_000, 400_000), (400_000, 800_000)):
    rec = {"n": n, "m": m}
    for tag, min_nnz in (("csr", "1000000000000"), ("strip", "1")):
        os.environ["SLP_STRIP_MIN_NNZ"] = min_nnz
        a = DeviceMatrix.random(m, n, 1e-3, 0)
        rec["nnz"] = a.nnz
        rec[tag + "_Ax_ms"] = a.bench_spmv(False, reps=20)
        rec[tag + "_ATy_ms"] = a.bench_spmv(True, reps=20)
        rec[tag + "_uses_strip"] = lib.slp_matrix_spmv_kernel(a._h, 0)
        if tag == "csr" and a.nnz <= 2e8:
            s = a.download()
            t0 = time.perf_counter()
            b = DeviceMatrix.from_csr(s)
            lib.slp_synchronize()
            rec["host_to_device_s"] = time.perf_counter() - t0
            rec["host_to_device_GBps"] = 12 * s.nnz / rec["host_to_device_s"] / 1e9
            b.close()
        a.close()
    rows.append(rec)
    print(json.dumps(rec), flush=True)
