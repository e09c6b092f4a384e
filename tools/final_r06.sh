#!/bin/bash
# Round-6 bench lines for profiles/ (run on the GPU box from the repo root): the default line (config 4, matrix-free ADMM + the
# Chambolle-Pock figure + general share + CPU baseline with its Chambolle-Pock partner + secondary config 3), Chambolle-Pock on
# config 4, SURVEY 8(d)'s 10 %-equality variant at config 3 and config 4 (both methods), config 5, config 3.
O=gpurun_out/final_r06
mkdir -p $O
SLP_TRACE=1 timeout 900 python bench.py > $O/bench_admm_c4_1gpu.json 2> $O/setup_trace_c4_1gpu.txt; echo rc=$?
timeout 900 python bench.py --method chambolle_pock_ppd --no-secondary --no-general > $O/bench_cp_c4_1gpu.json 2> /dev/null; echo rc=$?
for m in admm chambolle_pock_ppd; do
  s=$([ $m = admm ] && echo admm || echo cp)
  timeout 900 python bench.py --method $m --eq-frac 0.1 --no-secondary --no-general --no-cpu-baseline > $O/bench_${s}_c4_eq10.json 2> /dev/null; echo rc=$?
  timeout 600 python bench.py --config c3 --method $m --eq-frac 0.1 --no-general --no-cpu-baseline > $O/bench_${s}_c3_eq10.json 2> /dev/null; echo rc=$?
  timeout 600 python bench.py --config c3 --method $m --no-general > $O/bench_${s}_c3.json 2> /dev/null; echo rc=$?
done
timeout 900 python bench.py --config c5 > $O/bench_blocks_c5_1gpu.json 2> $O/bench_blocks_c5_1gpu.err; echo rc=$?
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final_r06/bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    ro = r["roofline"]
    print(f.split("/")[-1], round(r["value"], 3), "it/s", round(r["ms_per_step"], 3), "ms  frac", round(ro["frac"], 4),
          " Ax", round(ro["ms_per_product"], 3), " ATy", round(ro["spmv_transposed"]["ms_per_product"], 3),
          " setup", round(r["setup_seconds"], 2), " peak", round(r["setup_breakdown"]["peak_device_gb"], 1), " cpu", r.get("cpu_baseline", {}).get("value"),
          " cp", (r.get("chambolle_pock") or {}).get("value"), " eq", r["config"].get("eq_frac"))
PY
