#!/bin/bash
# Round evidence for profiles/: bench lines (ADMM = headline, CP), the setup timeline (SLP_TRACE), rocprofv3 kernel
# stats and the separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_*), all on BASELINE config 3.  The ADMM bench run
# also times the general fp64-strip path (roofline.general_fp64), so one profile covers the dictionary kernels and
# k_strip_spmv<0>.  Run on the GPU box from the repo root:  bash tools/profile_c3.sh [tag]
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/${1:-prof}
mkdir -p $O
SLP_TRACE=1 python3 bench.py > $O/bench_admm.json 2> $O/bench_admm.err
python3 bench.py --method chambolle_pock_ppd > $O/bench_cp.json 2> $O/bench_cp.err
rocprofv3 --kernel-trace --stats -d $R/$O/stats_admm -o admm -- python3 bench.py --steps 10 --no-cpu-baseline > $O/stats_admm_bench.json 2> $O/stats_admm.err
rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/$O/pmc_sq -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
for d in stats_admm pmc_fetch pmc_write pmc_sq; do find $O/$d -name "*.db" | head -3; done
python3 tools/summarize_rocprof.py db-stats $(find $O/stats_admm -name "*.db" | head -1) > $O/kernel_stats.csv 2>> $O/summ.err
python3 tools/summarize_rocprof.py db-pmc $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > $O/pmc_hbm.json 2>> $O/summ.err
python3 tools/summarize_rocprof.py db-sq $(find $O/pmc_sq -name "*.db" | head -1) > $O/pmc_sq.json 2>> $O/summ.err
find $O -name "*.db" -delete   # raw rocpd files: only the summaries travel back (gpurun merges at most 64 MiB)
find $O -name "*.csv" -size +20M -delete
for d in stats_admm pmc_fetch pmc_write pmc_sq; do rm -rf $O/$d; done
cut -c1-600 $O/bench_admm.json; echo; cut -c1-300 $O/bench_cp.json; echo; grep "slp trace" $O/bench_admm.err
