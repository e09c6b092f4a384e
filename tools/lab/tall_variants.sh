#!/bin/bash
# Same-box A/B of kernel-lab builds of the tall-cell product kernel on the 2.5e6 x 1e7 slice (tools/tall_lab.py: HIP-event times of
# both orientations, bit-for-bit comparison of the dictionary and the fp64-entry copies): every variant twice, interleaved.
#   bash tools/lab/tall_variants.sh out.log variant1 variant2 ...      ("" = the shipped library)
OUT=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "variant=[${v:-shipped}] $(SLP_LIB_VARIANT=$v TALL_LAB_WIDE=0 TALL_LAB_CSR=0 timeout 600 python tools/tall_lab.py 10 2>&1 | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('Ax', d['tall_Ax_ms'], 'ATy', d['tall_ATy_ms'], 'fp64 Ax', d['tall_fp64_Ax_ms'], 'ATy', d['tall_fp64_ATy_ms'], 'dict == fp64 bitwise', d['tall_fp64_equals_tall_bitwise'], 'frac', d['tall_Ax_frac_of_8TBps'])
except Exception as e:
    print('failed', e)
")" >> $OUT
  done
done
