"""Profiling target: ONLY the two products of the metric's LP (config 4: 1e7 variables x 2e7 rows at 1e-4, chunked) -- every launch of
k_tall_spmv<true, false, false> in the trace is then a whole product (all chunks in one launch), so a PMC summary per launch is a
summary per product.  (In a solver run the same kernel name also covers chunk-sized launches of the set-up.)
    python tools/c4_products_only.py [reps] [chunks]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysparselp_amd import _lib  # noqa: E402
from pysparselp_amd.device import ChunkedDeviceMatrix, DeviceMatrix  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n, m, p, seed = 10_000_000, 20_000_000, 1e-4, 0
lib = _lib.lib(0)
cuts = ChunkedDeviceMatrix.cuts(m, chunks)
a = ChunkedDeviceMatrix(n, expect_chunks=len(cuts) - 1, expect_rows=m)
for r0, r1 in zip(cuts, cuts[1:]):
    a.append(DeviceMatrix.random(r1 - r0, n, p, seed, r0))   # (no LP vectors: nothing but the generator and the conversions runs before)
out = {"nnz": a.nnz, "chunks": a.chunks}
for t, name in ((False, "Ax"), (True, "ATy")):
    out[name + "_kernel"] = a.spmv_kernel(t)
    out[name + "_launches_per_product"] = int(lib.slp_matrix_product_launches(a._h, int(t)))
    out[name + "_ms"] = a.bench_spmv(t, reps=reps)
    out[name + "_copy_bytes"] = int(lib.slp_matrix_format_bytes(a._h, int(t)))
print(json.dumps(out))
