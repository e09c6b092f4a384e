# config-5-shaped block run (5e5 x 5e7 at 1e-4 on one GPU), 1 and 4 blocks per rank; bench.py turns SLP_TALL_SPLIT on for admm_blocks
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
for g in 1 4; do
  python bench.py --method admm_blocks --blocks-per-rank $g --n 50000000 --m 500000 --density 1e-4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_blocks_c5shape_split_g$g.json 2> gpurun_out/r03_bench_blocks_c5shape_split_g$g.err
  tail -3 gpurun_out/r03_bench_blocks_c5shape_split_g$g.err
  python -c "
import json
d=json.load(open('gpurun_out/r03_bench_blocks_c5shape_split_g$g.json')); r=d['roofline']; print(d['metric'], d['value'], d['ms_per_step'], r['kernel'][:14], r['frac'], r['ms_per_product'], r['spmv_transposed']['ms_per_product'], d['setup_seconds'], d['config'].get('cg_steps_per_iteration'), d['device_memory'])
"
done
