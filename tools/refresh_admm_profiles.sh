cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
timeout 200 python3 bench.py > gpurun_out/final_admm.json 2> gpurun_out/final_admm.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_admm -o admm -- python3 bench.py --steps 10 --no-cpu-baseline > gpurun_out/prof_admm_bench.json 2> gpurun_out/prof_admm.err
timeout 600 python3 tools/scaling_compute_only.py > gpurun_out/scaling.json 2> gpurun_out/scaling.err
cut -c1-200 gpurun_out/final_admm.json
