#!/bin/bash
# Same-box attribution of the round's two late changes at config 4: libslp_hip_r5mid.so (-DSLP_TALL_FULL_ISSUE on slp_tall_spmv.hip,
# -DSLP_TALL_DEAL_CEIL on slp_tall.hip: the kernel and the dealing as they stood mid-round), libslp_hip_ceil.so (-DSLP_TALL_DEAL_CEIL
# only: the per-wave issue of slots 4-7 with the old dealing) and the current library.  One box, back to back, twice.
for rep in 1 2; do
  for v in r5mid ceil ""; do
    echo "variant=[${v:-final}] $(SLP_LIB_VARIANT=$v timeout 280 python bench.py --no-cpu-baseline --no-secondary --no-general 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('it/s', round(d['value'],3), 'step', round(d['ms_per_step'],2), 'Ax', round(r['ms_per_product'],3), 'ATy', round(r['spmv_transposed']['ms_per_product'],3), 'GB per product', round(r['bytes_per_product']/1e9,2), 'frac', round(r['frac'],4))
")"
  done
done
