"""Pairs or quads?  ADMM step time and the two SpMV times for the row block a rank holds at N = 1, 2, 4, 8 (config 3 split by
rows), with the dictionary geometry forced to pairs (SLP_DICT_VARIANT=1), to quads (=2) and chosen by the library's rule.
    python tools/variant_rule.py > gpurun_out/variant_rule.json"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(rows, variant):
    env = dict(os.environ)
    env.pop("SLP_DICT_VARIANT", None)
    if variant:
        env["SLP_DICT_VARIANT"] = str(variant)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-general", "--steps", "20", "--warmup", "3",
                          "--m", str(rows)], capture_output=True, text=True, check=True, env=env).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    return {"ms_per_step": round(d["ms_per_step"], 4), "Ax_ms": round(d["roofline"]["ms_per_product"], 4),
            "ATy_ms": round(d["roofline"]["spmv_transposed"]["ms_per_product"], 4), "objective": d["objective_after_run"]}


res = {}
for n_gpus in (1, 2, 4, 8):
    rows = 2_000_000 // n_gpus
    res[f"N={n_gpus} ({rows} rows)"] = {name: run(rows, v) for name, v in (("rule", 0), ("pairs", 1), ("quads", 2))}
print(json.dumps(res, indent=1))
