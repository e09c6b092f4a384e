#!/bin/bash
# full GPU suite + the default bench line (what the driver runs at round end)
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/full/pytest_gpu.log 2>&1; echo pytest rc=$?
tail -5 gpurun_out/full/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/full/bench_default.json 2> gpurun_out/full/bench_default.err; echo bench rc=$?
python - <<'PY'
import json
r = json.load(open("gpurun_out/full/bench_default.json"))
print(r["metric"], r["value"], r["ms_per_step"], r["roofline"]["frac"], r["setup_seconds"], r["setup_breakdown"], r.get("cpu_baseline", {}).get("value"))
s = r.get("secondary", {}).get("c3")
if s:
    print("c3:", s["value"], s["ms_per_step"], s["roofline"]["frac"], s["setup_seconds"], s["setup_breakdown"], s["device_memory"])
PY
tail -3 gpurun_out/full/bench_default.err
