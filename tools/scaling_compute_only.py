"""Compute side of the strong-scaling curve, measured on ONE GPU: the row block a rank holds when the metric's LP
(BASELINE config 4 at density 1e-4: 1e7 variables x 2e7 rows, bench.py's default) is partitioned over N = 1, 2, 4, 8 GPUs,
WITHOUT the all-reduces -- what bounds the scaling from the compute side (the replicated elementwise work over the n
variables does not shrink with N).  `--config c3` does the same for config 3.
    python tools/scaling_compute_only.py [--config c4] > profiles/rNN_scaling_compute_only.json"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"c4": (10_000_000, 20_000_000, 1e-4), "c3": (1_000_000, 2_000_000, 1e-3)}


def run(extra):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-general", "--no-secondary",
                          "--steps", "10", "--warmup", "2"] + extra, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    return {"it_per_s": d["value"], "ms_per_step": d["ms_per_step"], "Ax_ms": d["roofline"]["ms_per_product"],
            "ATy_ms": d["roofline"]["spmv_transposed"]["ms_per_product"], "chunks": d["config"]["chunks_per_rank"],
            "products_ms": 2 * (d["roofline"]["ms_per_product"] + d["roofline"]["spmv_transposed"]["ms_per_product"]),
            "setup_seconds": d["setup_seconds"]}


p = argparse.ArgumentParser()
p.add_argument("--config", default="c4", choices=sorted(SHAPES))
args = p.parse_args()
n, m, dens = SHAPES[args.config]
res = {"what": __doc__.split("\n    python")[0], "config": args.config, "local_row_block": {}}
base = None
for n_gpus in (1, 2, 4, 8):
    rows = m // n_gpus
    r = run(["--vars", str(n), "--rows", str(rows), "--density", str(dens)])
    base = base or r["ms_per_step"]
    r["compute_only_speedup_vs_N1"] = base / r["ms_per_step"]
    r["not_in_the_products_ms"] = r["ms_per_step"] - r["products_ms"]   # elementwise passes over n + rows unknowns, dot products, launches
    res["local_row_block"][f"N={n_gpus} ({rows} rows)"] = r
print(json.dumps(res, indent=1))
