#!/bin/bash
# Round evidence for profiles/ on the metric's named LP (BASELINE config 4 at density 1e-4, chunked, ONE GPU): rocprofv3 kernel
# stats of the ADMM bench and the separate PMC passes (FETCH_SIZE / WRITE_SIZE) -- one tall-cell launch per chunk and product,
# so a product's traffic = 8 launches.  Run on the GPU box from the repo root:  bash tools/profile_c4.sh [tag]
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/${1:-prof_c4}
mkdir -p $O
A="--config c4 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $R/$O/stats_admm -o admm -- python3 bench.py $A --steps 5 --warmup 2 > $O/stats_admm_bench.json 2> $O/stats_admm.err
rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
for d in stats_admm pmc_fetch pmc_write; do find $O/$d -name "*.db" | head -3; done
python3 tools/summarize_rocprof.py db-stats $(find $O/stats_admm -name "*.db" | head -1) > $O/kernel_stats.csv 2>> $O/summ.err
python3 tools/summarize_rocprof.py db-pmc $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > $O/pmc_hbm.json 2>> $O/summ.err
find $O -name "*.db" -delete
find $O -name "*.csv" -size +20M -delete
for d in stats_admm pmc_fetch pmc_write; do rm -rf $O/$d; done
head -8 $O/kernel_stats.csv; grep -A8 tall_spmv $O/pmc_hbm.json | head -24; cut -c1-400 $O/stats_admm_bench.json
