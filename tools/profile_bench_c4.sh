cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench/stats -o s -- python3 $R/bench.py --steps 10 --no-cpu-baseline --no-secondary --no-general > $R/gpurun_out/prof_bench_line.json 2> /dev/null
cd $R
f=$(find gpurun_out/prof_bench/stats -name "*.db" | head -1)
if [ -n "$f" ]; then timeout 120 python3 tools/summarize_rocprof.py db-stats $f > gpurun_out/bench_admm_c4_kernel_stats.csv; fi
rm -rf gpurun_out/prof_bench
head -12 gpurun_out/bench_admm_c4_kernel_stats.csv | cut -c1-150
python3 -c "
import json; d=json.loads(open('gpurun_out/prof_bench_line.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['ms_per_product'], r['frac'])"
