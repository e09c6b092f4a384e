#!/bin/bash
# Same-box comparison of library variants on config 3 (SLP_LIB_VARIANT names as arguments; "" = the shipped library)
O=gpurun_out/ab_c3
mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
    SLP_LIB_VARIANT=$v timeout 600 python bench.py --config c3 --no-cpu-baseline --no-general > $O/c3_${v:-new}_$rep.json 2>> $O/err.log
    python - "$O/c3_${v:-new}_$rep.json" "${v:-new}" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    ms = r.get("ms_per_product", r.get("ms_per_launch")); mst = r["spmv_transposed"].get("ms_per_product", r["spmv_transposed"].get("ms_per_launch"))
    print(f"{sys.argv[2]:12s} {d['value']:8.3f} it/s  step {d['ms_per_step']:7.3f} ms  Ax {ms:6.3f} ms frac {r['frac']:.4f}  ATy {mst:6.3f} ms  obj {d['objective_after_run']:.9f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
done
