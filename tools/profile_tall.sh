#!/bin/bash
# kernel stats, SQ counters and HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) of the tall-cell SpMV on the
# 2.5e6 x 1e7 slice; summaries: tools/summarize_rocprof.py db-stats / db-sq / db-pmc
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/prof_tall
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s -- python3 tools/tall_only.py 5 > $O/stats.json 2> $O/stats.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/$O/sq1 -o s -- python3 tools/tall_only.py 3 > /dev/null 2> $O/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $R/$O/sq2 -o s -- python3 tools/tall_only.py 3 > /dev/null 2> $O/sq2.err
rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o f -- python3 tools/tall_only.py 3 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o w -- python3 tools/tall_only.py 3 > /dev/null 2> $O/write.err
for d in stats sq1 sq2 fetch write; do find $O/$d -name "*.db" | head -2; done
S=$(find $O/stats -name "*.db" | head -1); python3 tools/summarize_rocprof.py db-stats $S > $O/kernel_stats.csv
python3 tools/summarize_rocprof.py db-sq $(find $O/sq1 -name "*.db" | head -1) > $O/sq1.json
python3 tools/summarize_rocprof.py db-sq $(find $O/sq2 -name "*.db" | head -1) > $O/sq2.json
python3 tools/summarize_rocprof.py db-pmc $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) > $O/pmc_hbm.json
for d in stats sq1 sq2 fetch write; do rm -rf $O/$d; done  # raw rocpd files are hundreds of MB: only the summaries travel back
head -12 $O/kernel_stats.csv; cat $O/sq1.json $O/sq2.json; grep -A 6 tall_spmv $O/pmc_hbm.json
