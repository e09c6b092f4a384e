#!/bin/bash
# Same-box A/B of kernel-lab builds at config 4 (bench.py, ADMM): every variant `reps` times, interleaved.
#   bash tools/lab/c4_variants.sh out.log variant1 variant2 ...      ("" = the shipped library)
OUT=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "variant=[${v:-shipped}] $(SLP_LIB_VARIANT=$v timeout 400 python bench.py --no-cpu-baseline --no-secondary --no-general 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('it/s', round(d['value'],3), 'step', round(d['ms_per_step'],2), 'Ax', round(r['ms_per_product'],3), 'ATy', round(r['spmv_transposed']['ms_per_product'],3), 'GB per product', round(r['bytes_per_product']/1e9,2), 'frac', round(r['frac'],4))
except Exception as e:
    print('failed', e)
")" >> $OUT
  done
done
