#!/bin/bash
# Same-box A/B of variant libraries at config 5 (bench.py --config c5, block-splitting ADMM): every variant twice, interleaved.
#   bash tools/lab/c5_variants.sh out.log variant1 variant2 ...      ("" = the shipped library)
OUT=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "variant=[${v:-shipped}] $(SLP_LIB_VARIANT=$v timeout 600 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('it/s', round(d['value'],4), 'step', round(d['ms_per_step'],1), 'Ax', round(r['ms_per_product'],3), 'ATy', round(r['spmv_transposed']['ms_per_product'],3), 'frac', round(r['frac'],4), 'obj', d['objective_after_run'])
except Exception as e:
    print('failed', e)
")" >> $OUT
  done
done
