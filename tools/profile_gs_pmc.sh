# PMC passes over the Gauss-Seidel ADMM on the Potts 256^2 LP: what bounds the single-workgroup sweep?
# (separate --pmc runs, kernel-trace only; see tools/profile_gs.sh for the timing pass)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  SLP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_gs_pmc$i -o gs -- python3 tools/bench_small.py --cp-iters 50 --admm-iters 60 --cpu-iters 2 > gpurun_out/prof_gs_pmc$i.json 2> gpurun_out/prof_gs_pmc$i.err || tail -3 gpurun_out/prof_gs_pmc$i.err
done
python3 - <<'PY'
import sqlite3, collections, glob, json
out = collections.defaultdict(dict)
for db in sorted(glob.glob("gpurun_out/prof_gs_pmc[0-9]/*results.db")):
    cur = sqlite3.connect(db).cursor()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, counter, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if "k_gs_sweep" in name:
            agg[name.split("(")[0]][counter].append(float(value))
    for k, d in agg.items():
        for c, v in d.items():
            out[k][c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/prof_gs_pmc_summary.json", "w"), indent=1)
PY
