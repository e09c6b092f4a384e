cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_sq_dict -o s -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --method chambolle_pock_ppd > /dev/null 2> gpurun_out/pmc_sq_dict.err
tail -3 gpurun_out/pmc_sq_dict.err
