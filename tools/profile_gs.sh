cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
SLP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gs -o gs -- python3 tools/bench_small.py --cp-iters 200 --admm-iters 200 --cpu-iters 2 > gpurun_out/prof_gs.json 2> gpurun_out/prof_gs.err
tail -1 gpurun_out/prof_gs.json | cut -c1-600
S=$(find $R/gpurun_out/prof_gs -name "*.db" | head -1); python3 tools/summarize_rocprof.py db-stats $S > gpurun_out/prof_gs_kernel_stats.csv; rm -rf $R/gpurun_out/prof_gs
head -16 gpurun_out/prof_gs_kernel_stats.csv
