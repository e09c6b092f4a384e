#!/bin/bash
# Round evidence for profiles/: final bench lines, rocprofv3 kernel stats and the separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ_*), all on BASELINE config 3.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python3 bench.py > gpurun_out/final_admm.json 2> gpurun_out/final_admm.err
python3 bench.py --method chambolle_pock_ppd > gpurun_out/final_cp.json 2> gpurun_out/final_cp.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_admm -o admm -- python3 bench.py --steps 10 --no-cpu-baseline > gpurun_out/prof_admm_bench.json 2> gpurun_out/prof_admm.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cp -o cp -- python3 bench.py --steps 10 --no-cpu-baseline --method chambolle_pock_ppd > gpurun_out/prof_cp_bench.json 2> gpurun_out/prof_cp.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_admm -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_fetch_admm.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_admm -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_write_admm.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_sq -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_sq.err
python3 tools/scaling_compute_only.py > gpurun_out/scaling.json 2> gpurun_out/scaling.err
find gpurun_out -name "*.csv" -size +20M -delete
cut -c1-400 gpurun_out/final_admm.json; cut -c1-300 gpurun_out/final_cp.json
timeout 300 python3 bench.py --method admm_blocks --steps 3 --warmup 1 > gpurun_out/final_admm_blocks.json 2> gpurun_out/final_admm_blocks.err
cut -c1-300 gpurun_out/final_admm_blocks.json
