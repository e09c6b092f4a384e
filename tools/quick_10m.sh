#!/bin/bash
# a 10^7-variable / 2*10^7-row random LP that fits one GPU: density 2e-5 (4e9 stored entries, 200 per row, 400 per column)
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "it/s", round(d["ms_per_step"],3), "ms; spmv", round(d["roofline"]["ms_per_launch"],3), round(d["roofline"]["spmv_transposed"]["ms_per_launch"],3), "obj", d["objective_after_run"], "copyGB", round(d["roofline"]["matrix_copy_bytes_per_launch"]/1e9,3), d["roofline"]["kernel"][:16], "setup_s", round(d["setup_seconds"],1))'
A="--n 10000000 --m 20000000 --density 2e-5 --no-cpu-baseline --steps 5 --warmup 1"
timeout 900 python bench.py $A 2>&1 | tail -1 > gpurun_out/bench_admm_10m.json; python -c "$P" admm < gpurun_out/bench_admm_10m.json
timeout 900 python bench.py $A --method chambolle_pock_ppd 2>&1 | tail -1 > gpurun_out/bench_cp_10m.json; python -c "$P" cp < gpurun_out/bench_cp_10m.json
