#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and kernel stats of the config-5-shaped block run (5e5 x 5e7 at 1e-4,
# block-splitting ADMM, strip-range split on): summaries under gpurun_out/prof_c5/
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/prof_c5
mkdir -p $O
A="--method admm_blocks --vars 50000000 --rows 500000 --density 1e-4 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s -- python3 bench.py $A --steps 3 --warmup 1 > $O/bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o f -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o w -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/write.err
python3 tools/summarize_rocprof.py db-stats $(find $O/stats -name "*.db" | head -1) > $O/kernel_stats.csv
python3 tools/summarize_rocprof.py db-pmc $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) > $O/pmc_hbm.json
for d in stats fetch write; do rm -rf $O/$d; done
head -8 $O/kernel_stats.csv; grep -A7 "tall_spmv" $O/pmc_hbm.json | head -30; cut -c1-300 $O/bench.json
