#!/bin/bash
# Round 6: config 5 (bench.py --config c5, block-splitting ADMM, 8 blocks on one GPU) with the blocks' projections block after
# block (SLP_BLOCKS_STREAMS=0, the default) and side by side on the blocks' streams (=1), interleaved on ONE box; the objective after the
# run must not change by a bit.      bash tools/lab/c5_streams_ab.sh out.log [reps]
OUT=${1:-gpurun_out/c5_streams_ab.log}
REPS=${2:-2}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for rep in $(seq 1 $REPS); do
  for mode in ${MODES:-0 1}; do
    export SLP_BLOCKS_STREAMS=$mode
    echo "streams=[$mode] $(timeout 900 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('it/s', round(d['value'],4), 'step', round(d['ms_per_step'],1), 'Ax', round(r['ms_per_product'],3), 'frac', round(r['frac'],4), 'cg', d['config'].get('cg_steps_per_iteration'), 'obj', repr(d['objective_after_run']))
except Exception as e:
    print('failed', e)
")" >> "$OUT"
  done
done
cat "$OUT"
