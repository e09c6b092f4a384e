#!/bin/bash
# Round-4 bench lines for profiles/ (run on the GPU box from the repo root): the default line (config 4 + secondary config 3),
# Chambolle-Pock on config 4, both methods on the 1/8 slice and on config 3.
O=gpurun_out/final
mkdir -p $O
SLP_TRACE=1 timeout 900 python bench.py > $O/bench_admm_c4_1gpu.json 2> $O/setup_trace_c4_1gpu.txt; echo rc=$?
timeout 900 python bench.py --method chambolle_pock_ppd --no-secondary > $O/bench_cp_c4_1gpu.json 2> /dev/null; echo rc=$?
timeout 600 python bench.py --config c4slice > $O/bench_admm_c4slice.json 2> /dev/null; echo rc=$?
timeout 600 python bench.py --config c4slice --method chambolle_pock_ppd > $O/bench_cp_c4slice.json 2> /dev/null; echo rc=$?
SLP_TRACE=1 timeout 600 python bench.py --config c3 > $O/bench_admm_c3.json 2> $O/setup_trace_c3.txt; echo rc=$?
timeout 600 python bench.py --config c3 --method chambolle_pock_ppd > $O/bench_cp_c3.json 2> /dev/null; echo rc=$?
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/bench_*.json")):
    r = json.load(open(f))
    print(f.split("/")[-1], round(r["value"], 3), "it/s", round(r["ms_per_step"], 3), "ms  frac", round(r["roofline"]["frac"], 4),
          " Ax", round(r["roofline"]["ms_per_launch"], 3), " ATy", round(r["roofline"]["spmv_transposed"]["ms_per_launch"], 3),
          " setup", round(r["setup_seconds"], 2), " cpu", r.get("cpu_baseline", {}).get("value"))
PY
