#!/bin/bash
# PMC (HBM bytes) and kernel stats of the wide-strip kernel on the 2.5e6 x 1e7 slice
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
A="--n 10000000 --m 2500000 --density 1e-4 --no-cpu-baseline --steps 4 --warmup 1 --method chambolle_pock_ppd"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_wide -o wide -- python3 bench.py $A > gpurun_out/prof_wide_bench.json 2> gpurun_out/prof_wide.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_wide -o f -- python3 bench.py $A > /dev/null 2> gpurun_out/pmc_fetch_wide.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_wide -o w -- python3 bench.py $A > /dev/null 2> gpurun_out/pmc_write_wide.err
ls gpurun_out/prof_wide gpurun_out/pmc_fetch_wide gpurun_out/pmc_write_wide
