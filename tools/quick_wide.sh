#!/bin/bash
# the 1/8 row slice of a 10^7-variable, density-10^-4 LP (2.5e6 x 1e7, 2.5e9 stored entries): wide-strip regime
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "it/s", round(d["ms_per_step"],3), "ms; spmv", round(d["roofline"]["ms_per_launch"],3), round(d["roofline"]["spmv_transposed"]["ms_per_launch"],3), "obj", d["objective_after_run"], "copyGB", round(d["roofline"]["matrix_copy_bytes_per_launch"]/1e9,3), d["roofline"]["kernel"][:16])'
A="--n 10000000 --m 2500000 --density 1e-4 --no-cpu-baseline --steps 10"
timeout 600 python bench.py $A --method chambolle_pock_ppd 2>&1 | tail -1 | python -c "$P" cp
timeout 600 python bench.py $A 2>&1 | tail -1 | python -c "$P" admm
timeout 600 python bench.py $A 2>&1 | tail -1 | python -c "$P" admm-again
SLP_VALUE_DICT=0 timeout 600 python bench.py $A 2>&1 | tail -1 | python -c "$P" admm-fp64
SLP_VALUE_DICT=0 timeout 600 python bench.py $A 2>&1 | tail -1 | python -c "$P" admm-fp64-again
