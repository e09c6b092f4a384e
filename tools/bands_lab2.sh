#!/bin/bash
# fetch distance 2 / 4 / 8 levels, 4 / 8 / 16 bands, Potts 256^2 ADMM (lab builds libslp_hip_d2.so / _d8.so: make variant NAME=d8 EXTRA=-DSLP_GS_FETCH_D=8)
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
for v in "" d2 d8; do for b in 4 8 16; do
  echo "variant=$v bands=$b"
  SLP_LIB_VARIANT=$v SLP_GS_BANDS=$b timeout 300 python3 tools/bench_small.py --admm-iters 2000 --cp-iters 100 --cpu-iters 3 2>gpurun_out/lab2.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('admm_gpu_it_per_s','admm_bit_exact_vs_oracle')})
except Exception as e: print("failed", e, open("gpurun_out/lab2.err").read()[-600:])
"
done; done
