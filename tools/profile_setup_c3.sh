#!/bin/bash
# kernel stats of a short config-3 run: what the setup kernels (generator, dictionary, transposition, strip conversion) cost
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/prof_setup
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general > $O/bench.json 2> $O/stats.err
S=$(find $O/stats -name "*.db" | head -1); python3 tools/summarize_rocprof.py db-stats $S > $O/kernel_stats.csv
rm -rf $O/stats
head -30 $O/kernel_stats.csv
