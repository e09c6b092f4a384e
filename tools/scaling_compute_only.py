"""Compute side of the strong-scaling curve, measured on ONE GPU: the row block a rank holds when BASELINE
config 3 (1e6 x 2e6, density 1e-3) is partitioned over N = 1, 2, 4, 8 GPUs, without the all-reduces.
Also the three seeds of the full problem.  python tools/scaling_compute_only.py > profiles/rNN_scaling_compute_only.json"""
import json
import subprocess
import sys
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--steps", "20", "--warmup", "3"] + extra,
                         capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    return {"it_per_s": d["value"], "ms_per_step": d["ms_per_step"], "Ax_ms": d["roofline"]["ms_per_launch"],
            "ATy_ms": d["roofline"]["spmv_transposed"]["ms_per_launch"], "objective": d["objective_after_run"]}


res = {"local_row_block": {}, "seeds_full_problem": {}}
for n_gpus in (1, 2, 4, 8):
    rows = 2_000_000 // n_gpus
    res["local_row_block"][f"N={n_gpus} ({rows} rows)"] = {
        "admm": run(["--m", str(rows)]), "chambolle_pock_ppd": run(["--m", str(rows), "--method", "chambolle_pock_ppd"])}
for seed in (0, 1, 2):
    res["seeds_full_problem"][f"seed={seed}"] = {"admm": run(["--seed", str(seed)]),
                                                 "chambolle_pock_ppd": run(["--seed", str(seed), "--method", "chambolle_pock_ppd"])}
print(json.dumps(res, indent=1))
