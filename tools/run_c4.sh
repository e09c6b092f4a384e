#!/bin/bash
# BASELINE config 4 at full size on one GPU (gpurun box): chunked tests, ADMM and Chambolle-Pock bench lines, full-size checks.
mkdir -p gpurun_out/c4
timeout 600 python -m pytest tests/test_gpu_chunked.py tests/test_gpu_tall.py -x -q -m gpu 2>&1 | tail -5
SLP_TRACE=1 timeout 900 python bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/c4/bench_admm_c4.json 2> gpurun_out/c4/bench_admm_c4.err; echo rc=$?
python - <<'PY'
import json
r = json.load(open("gpurun_out/c4/bench_admm_c4.json"))
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["ms_per_launch"], r["roofline"]["spmv_transposed"]["ms_per_launch"],
      r["setup_seconds"], r["setup_breakdown"], r["device_memory"], r.get("cpu_baseline"))
PY
grep -v "hipMalloc" gpurun_out/c4/bench_admm_c4.err | tail -12
timeout 900 python bench.py --config c4 --method chambolle_pock_ppd --steps 10 --warmup 2 > gpurun_out/c4/bench_cp_c4.json 2> gpurun_out/c4/bench_cp_c4.err; echo rc=$?
python - <<'PY'
import json
r = json.load(open("gpurun_out/c4/bench_cp_c4.json"))
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["setup_seconds"], r["objective_after_run"], r.get("cpu_baseline"))
PY
timeout 1200 python tools/c4_full.py > gpurun_out/c4/c4_full.log 2>&1; echo rc=$?
tail -c 1500 gpurun_out/c4/c4_full.log
