cd $GRAFT_REPO_ROOT
bash tools/final_r05.sh > gpurun_out/final_r05_summary.log 2>&1
for c in auto 4096; do if [ $c = auto ]; then unset SLP_TALL_C; else export SLP_TALL_C=$c; fi; echo "2.5e6 x 1e7 at 2e-4, strips $c: $(timeout 400 python3 tools/tall_only.py 5 2500000 10000000 2e-4 2>&1 | tail -1)"; done > gpurun_out/tall_2e-4_full_width.log 2>&1
unset SLP_TALL_C
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4/stats -o s -- python3 $GRAFT_REPO_ROOT/tools/c4_products_only.py 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_c4_stats.json 2> /dev/null
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_c4/stats -name "*.db" | head -1)
if [ -n "$f" ]; then timeout 120 python3 tools/summarize_rocprof.py db-stats $f > gpurun_out/c4_products_kernel_stats.csv; fi
rm -rf gpurun_out/prof_c4
cat gpurun_out/final_r05_summary.log | tail -9; cat gpurun_out/tall_2e-4_full_width.log | cut -c1-250; head -8 gpurun_out/c4_products_kernel_stats.csv | cut -c1-200
