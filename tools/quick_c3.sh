#!/bin/bash
# quick C3 timing: ADMM, CP and the two-vector pass (one line each)
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "it/s", round(d["ms_per_step"],3), "ms; spmv", round(d["roofline"]["ms_per_launch"],3), round(d["roofline"]["spmv_transposed"]["ms_per_launch"],3), "obj", d["objective_after_run"], "copyGB", round(d["roofline"]["matrix_copy_bytes_per_launch"]/1e9,3))'
timeout 300 python bench.py --steps 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P" admm
timeout 300 python bench.py --steps 20 --no-cpu-baseline --method chambolle_pock_ppd 2>&1 | tail -1 | python -c "$P" cp
SLP_BENCH_TWO_VECTORS=1 timeout 300 python bench.py --steps 3 --no-cpu-baseline --method chambolle_pock_ppd 2>&1 | tail -1 | python -c "$P" two-vector
