#!/bin/bash
# Kernel stats, HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the two products of config 4 alone
# (tools/c4_products_only.py: every k_tall_spmv launch is a whole product) -> gpurun_out/prof_c4_products/
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/prof_c4_products
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s -- python3 tools/c4_products_only.py 5 > $O/stats.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o f -- python3 tools/c4_products_only.py 2 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o w -- python3 tools/c4_products_only.py 2 > /dev/null 2> $O/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/$O/sq1 -o s -- python3 tools/c4_products_only.py 2 > /dev/null 2> $O/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $R/$O/sq2 -o s -- python3 tools/c4_products_only.py 2 > /dev/null 2> $O/sq2.err
python3 tools/summarize_rocprof.py db-stats $(find $O/stats -name "*.db" | head -1) > $O/kernel_stats.csv
python3 tools/summarize_rocprof.py db-pmc $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) > $O/pmc_hbm.json
python3 tools/summarize_rocprof.py db-sq $(find $O/sq1 -name "*.db" | head -1) > $O/sq1.json
python3 tools/summarize_rocprof.py db-sq $(find $O/sq2 -name "*.db" | head -1) > $O/sq2.json
for d in stats fetch write sq1 sq2; do rm -rf $O/$d; done
head -6 $O/kernel_stats.csv; grep -A6 "k_tall_spmv" $O/pmc_hbm.json; grep -A12 "k_tall_spmv" $O/sq1.json; grep -A10 "k_tall_spmv" $O/sq2.json; cat $O/stats.json
