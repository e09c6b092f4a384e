#!/bin/bash
# Round-5 bench lines for profiles/ (run on the GPU box from the repo root): the default line (config 4 + general share + CPU
# baseline + secondary config 3), Chambolle-Pock on config 4, config 5 (block-splitting ADMM, 8 blocks on one GPU), the 1/8 slice,
# config 3.
O=gpurun_out/final_r05
mkdir -p $O
SLP_TRACE=1 timeout 900 python bench.py > $O/bench_admm_c4_1gpu.json 2> $O/setup_trace_c4_1gpu.txt; echo rc=$?
timeout 900 python bench.py --method chambolle_pock_ppd --no-secondary > $O/bench_cp_c4_1gpu.json 2> /dev/null; echo rc=$?
timeout 900 python bench.py --config c5 > $O/bench_blocks_c5_1gpu.json 2> $O/bench_blocks_c5_1gpu.err; echo rc=$?
timeout 600 python bench.py --config c4slice > $O/bench_admm_c4slice.json 2> /dev/null; echo rc=$?
timeout 600 python bench.py --config c4slice --method chambolle_pock_ppd > $O/bench_cp_c4slice.json 2> /dev/null; echo rc=$?
timeout 600 python bench.py --config c3 > $O/bench_admm_c3.json 2> /dev/null; echo rc=$?
timeout 600 python bench.py --config c3 --method chambolle_pock_ppd > $O/bench_cp_c3.json 2> /dev/null; echo rc=$?
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final_r05/bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    ro = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 3), "it/s", round(r["ms_per_step"], 3), "ms  frac", round(ro["frac"], 4),
          " Ax", round(ro["ms_per_product"], 3), " ATy", round(ro["spmv_transposed"]["ms_per_product"], 3),
          " setup", round(r["setup_seconds"], 2), " peak", round(r["setup_breakdown"]["peak_device_gb"], 1), " cpu", r.get("cpu_baseline", {}).get("value"))
PY
