# rocprofv3 --kernel-trace --stats of the config-5 bench command (block-splitting ADMM, 8 blocks on one GPU): the average duration
# of k_tall_spmv<...> over the WHOLE run (iterations included) beside the bench line's roofline.ms_per_product
#   -> gpurun_out/bench_blocks_c5_kernel_stats.csv, gpurun_out/prof_bench_c5_line.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench_c5/stats -o s -- python3 $R/bench.py --config c5 --no-cpu-baseline > $R/gpurun_out/prof_bench_c5_line.json 2> /dev/null
cd $R
f=$(find gpurun_out/prof_bench_c5/stats -name "*.db" | head -1)
if [ -n "$f" ]; then timeout 300 python3 tools/summarize_rocprof.py db-stats $f > gpurun_out/bench_blocks_c5_kernel_stats.csv; fi
rm -rf gpurun_out/prof_bench_c5
head -8 gpurun_out/bench_blocks_c5_kernel_stats.csv | cut -c1-150
python3 -c "
import json; d=json.loads(open('gpurun_out/prof_bench_c5_line.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['ms_per_product'], r['frac'], r.get('products_timed'))"
