#!/bin/bash
# SQ counters and HBM bytes (separate passes) of the setup kernels on a short config-3 run
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
cd "$R" || exit 1
O=gpurun_out/prof_setup
mkdir -p $O
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/$O/sq1 -o s -- $CMD > /dev/null 2> $O/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $R/$O/sq2 -o s -- $CMD > /dev/null 2> $O/sq2.err
rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o f -- $CMD > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o w -- $CMD > /dev/null 2> $O/write.err
python3 tools/summarize_rocprof.py db-sq $(find $O/sq1 -name "*.db" | head -1) > $O/sq1.json
python3 tools/summarize_rocprof.py db-sq $(find $O/sq2 -name "*.db" | head -1) > $O/sq2.json
python3 tools/summarize_rocprof.py db-pmc $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) > $O/pmc_hbm.json
for d in sq1 sq2 fetch write; do rm -rf $O/$d; done
python3 - <<'PY'
import json
for f in ("sq1", "sq2", "pmc_hbm"):
    d = json.load(open(f"gpurun_out/prof_setup/{f}.json"))
    ks = d.get("kernels", d)
    for k, v in ks.items():
        if any(s in k for s in ("strip_fill", "strip_count", "random_rows", "value_set")):
            print(f, k[:60], json.dumps(v)[:420])
PY
