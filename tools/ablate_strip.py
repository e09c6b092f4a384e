"""Timing experiments on the strip SpMV at BASELINE config 3 (wrong results in ablated modes):
which part of k_strip_spmv costs what.  make -C pysparselp_amd/csrc ablation && python tools/ablate_strip.py"""
import os, sys, json
os.environ["SLP_LIB_VARIANT"] = "ablation"  # make -C pysparselp_amd/csrc ablation ; the shipped library has no ablation switch
os.environ.setdefault("SLP_VALUE_DICT", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysparselp_amd.device import DeviceMatrix

n, m, p = 1_000_000, 2_000_000, 1e-3
a = DeviceMatrix.random(m, n, p, 0)
out = {}
for mode, name in ((0, "full"), (1, "no_tile_staging"), (2, "no_entry_streaming")):
    os.environ["SLP_STRIP_ABLATE"] = str(mode)
    out[name + "_Ax_ms"] = a.bench_spmv(False, reps=10)
    out[name + "_ATy_ms"] = a.bench_spmv(True, reps=10)
os.environ["SLP_STRIP_ABLATE"] = "0"
print(json.dumps(out))
