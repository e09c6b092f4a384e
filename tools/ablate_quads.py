"""Which resource bounds the value-dictionary quad kernel (k_qstrip_spmv<1>) at BASELINE config 3?  Timing experiments in
the -DSLP_ABLATION build (WRONG results by design): 1 = no value-table lookup (one LDS gather per entry instead of two),
2 = no LDS gathers, 3 = no entry loads (synthetic entries: LDS gathers and arithmetic only).

    make -C pysparselp_amd/csrc ablation && python tools/ablate_quads.py
"""
import json
import os
import sys

os.environ["SLP_LIB_VARIANT"] = "ablation"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysparselp_amd.device import DeviceMatrix  # noqa: E402

a = DeviceMatrix.random(2_000_000, 1_000_000, 1e-3, 0)
out = {}
for mode, name in ((0, "full"), (1, "no_value_table_lookup"), (2, "no_lds_gathers"), (3, "no_entry_loads")):
    os.environ["SLP_QSTRIP_ABLATE"] = str(mode)
    a.bench_spmv(False, reps=2)
    out[name + "_ms"] = round(min(a.bench_spmv(False, reps=10) for _ in range(3)), 4)
print(json.dumps(out))
