#!/bin/bash
# Round 6, VERDICT r05 #7: k_tall_build before (libslp_hip_oldbuild.so: HEAD~'s slp_tall.hip, 42 vector + 130 scalar registers
# spilled) and after (no scratch) on ONE box: set-up trace, products' SHA-256 (the copies must give the same bits), kernel stats.
#   bash tools/lab/ab_builder.sh   -> gpurun_out/ab_builder.log, gpurun_out/ab_builder_{old,new}_stats.csv
set -u
mkdir -p gpurun_out
L=gpurun_out/ab_builder.log
: > $L
for round in 1 2; do
  for v in oldbuild ""; do
    echo "variant=[$v] round $round" >> $L
    SLP_LIB_VARIANT=$v SLP_TRACE=1 TALL_ONLY_HASH=1 python tools/tall_only.py 5 >> $L 2>&1
  done
done
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/ab_builder_prof
for v in oldbuild new; do
  if [ $v = new ]; then unset SLP_LIB_VARIANT; else export SLP_LIB_VARIANT=$v; fi
  mkdir -p $O/$v
  rocprofv3 --kernel-trace --stats -d $R/$O/$v/stats -o s -- python3 tools/tall_only.py 2 > /dev/null 2> $O/$v/stats.err
  rocprofv3 --pmc FETCH_SIZE -d $R/$O/$v/fetch -o f -- python3 tools/tall_only.py 2 > /dev/null 2> $O/$v/fetch.err
  rocprofv3 --pmc WRITE_SIZE -d $R/$O/$v/write -o w -- python3 tools/tall_only.py 2 > /dev/null 2> $O/$v/write.err
  python3 tools/summarize_rocprof.py db-stats $(find $O/$v/stats -name "*.db" | head -1) > gpurun_out/ab_builder_${v}_stats.csv
  python3 tools/summarize_rocprof.py db-pmc $(find $O/$v/fetch -name "*.db" | head -1) $(find $O/$v/write -name "*.db" | head -1) > gpurun_out/ab_builder_${v}_pmc_hbm.json
  rm -rf $O/$v
done
