#!/bin/bash
# Gauss-Seidel bands: the fuzz over every plan (bit-exact against the oracle), then Potts 256^2 ADMM with 0 / auto / 4 / 8 / 16 bands
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
timeout 600 python3 tools/fuzz_gs.py --cases ${1:-150} --seed ${2:-5} 2>&1 | tail -4
for b in 0 auto 4 8 16; do
  if [ $b = auto ]; then unset SLP_GS_BANDS; else export SLP_GS_BANDS=$b; fi
  echo "bands=$b"
  SLP_GS_VERBOSE=1 timeout 300 python3 tools/bench_small.py --admm-iters 1000 --cp-iters 100 --cpu-iters 3 2>gpurun_out/bands_$b.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('admm_gpu_it_per_s','admm_bit_exact_vs_oracle','admm_setup_s')})
except Exception as e: print('failed', e)
"
  grep "gauss-seidel bands" gpurun_out/bands_$b.err | sort | uniq | head -5
done
