#!/bin/bash
# Round-6 evidence beyond the bench lines (GPU box, repo root): rocprofv3 of the two products of config 4 (kernel stats, FETCH /
# WRITE in separate passes, both SQ passes), rocprofv3 --kernel-trace --stats of the bench command itself, every row of the
# resident copies against the oracle, the streamed-oracle whole-solver parity on the final kernel, fresh-seed fuzz.
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/final_r06
mkdir -p $O
bash tools/profile_c4_products.sh > $O/profile_c4_products.log 2>&1
bash tools/profile_bench_c4.sh > $O/profile_bench_c4.log 2>&1
cd "$R"
timeout 1200 python tools/c4_all_rows.py --chunks 16 --out $O/c4_all_rows.json > $O/c4_all_rows.log 2>&1; echo all_rows rc=$?
timeout 1500 python tools/c4_streamed_parity.py --out $O/c4_streamed_oracle_parity.json > $O/c4_streamed.log 2>&1; echo streamed rc=$?
timeout 1500 python tools/c4_streamed_parity.py --eq-frac 0.1 --out $O/c4_eq10_streamed_oracle_parity.json > $O/c4_eq10_streamed.log 2>&1; echo streamed eq rc=$?
timeout 900 python tools/fuzz_spmv.py --cases 300 --seed 606 > $O/fuzz_spmv.log 2>&1; echo fuzz_spmv rc=$?
timeout 900 python tools/fuzz_chunked.py --cases 200 --seed 607 > $O/fuzz_chunked.log 2>&1; echo fuzz_chunked rc=$?
for m in admm chambolle_pock_ppd; do
  s=$([ $m = admm ] && echo admm || echo cp)
  timeout 900 python bench.py --method $m --eq-frac 0.1 --no-secondary --no-general --no-cpu-baseline > $O/bench_${s}_c4_eq10.json 2> /dev/null; echo rc=$?
  timeout 900 python bench.py --method $m --no-secondary --no-general --no-cpu-baseline > $O/bench_${s}_c4_same_box.json 2> /dev/null; echo rc=$?
done
