#!/bin/bash
# kernel stats of the config-5-shaped block run (5e5 x 5e7 at 1e-4, one block): where setup and an iteration spend their time
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/prof_c5
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s -- python3 bench.py --method admm_blocks --n 50000000 --m 500000 --density 1e-4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/stats.err
S=$(find $O/stats -name "*.db" | head -1); python3 tools/summarize_rocprof.py db-stats $S > $O/kernel_stats.csv
rm -rf $O/stats
head -25 $O/kernel_stats.csv
