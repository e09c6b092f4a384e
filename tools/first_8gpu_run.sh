#!/bin/bash
# The first run on a node with 8 MI355X -- self-diagnosing: every bench line carries `exchange`
# (ms per iteration inside collectives, bytes and algbw per collective, max over the ranks) and `compute_ms_per_step` next to
# `ms_per_step`, so a scaling shortfall can be read off as exchange or as compute without a second run.
#
#   0. tools/rccl_preflight.py      (dlopen + symbols, device count, IPC mode, one-rank and N-rank 80 MB all-reduce timing)
#   1. tests/test_gpu_multi_rccl.py   (N = 2 / 4 / 8 processes, one per GPU, RCCL: replicas identical, x = the one-process run)
#   2. bench.py --gpus 1 / 2 / 4 / 8 on config 4 (matrix-free ADMM; replicated updates, then SLP_SHARD_UPDATES=1: 6 collectives
#      per iteration instead of 2) and on config 5 (block-splitting ADMM, 8 / N blocks per rank, per-block all-reduce overlapped
#      with the next block's projection)
# Every JSON line is appended to ONE file (default gpurun_out/first_8gpu_run.jsonl) with the launch parameters in front.
# No number from this script is a measured scaling result until a driver SCALE_r*.json carries it.
#
# Usage (repo root):  bash tools/first_8gpu_run.sh [out.jsonl] [steps]
set -u
OUT=${1:-gpurun_out/first_8gpu_run.jsonl}
STEPS=${2:-20}
mkdir -p "$(dirname "$OUT")"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "from pysparselp_amd import _lib; print(_lib.load().slp_device_count())")
echo "{\"what\": \"first_8gpu_run\", \"gpus_visible\": $NGPU, \"steps\": $STEPS}" >> "$OUT"

echo "== preflight (librccl symbols, device count, HSA_ENABLE_IPC_MODE_LEGACY, a one-rank and an N-rank 80 MB all-reduce)"
python tools/rccl_preflight.py --ranks "$NGPU" 2>> "${OUT%.jsonl}.err" | tail -1 | tee -a "$OUT"

echo "== RCCL tests (skip below 2 GPUs)"
python -m pytest tests/test_gpu_multi_rccl.py -m gpu -q -x 2>&1 | tail -5 | tee -a "${OUT%.jsonl}.tests.log"

port=29700
run() {   # run <config> <gpus> <shard 0|1>
    local cfg=$1 n=$2 shard=$3 line
    port=$((port + 11))
    if [ "$n" -gt "$NGPU" ]; then
        echo "{\"config\": \"$cfg\", \"gpus\": $n, \"skipped\": \"only $NGPU GPU(s) visible\"}" >> "$OUT"
        return
    fi
    # bench.py starts its own ranks (self_launch: child processes + the TCP id exchange of pysparselp_amd/parallel.py, no torch);
    # the driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` gives the same line
    line=$(SLP_SHARD_UPDATES=$shard MASTER_PORT=$port python bench.py --config "$cfg" --gpus "$n" --steps "$STEPS" --warmup 3 --no-cpu-baseline \
           --no-general --no-secondary 2>> "${OUT%.jsonl}.err" | grep '^{' | tail -1)
    if [ -z "$line" ]; then line="{\"error\": \"no bench line (see ${OUT%.jsonl}.err)\"}"; fi
    echo "{\"config\": \"$cfg\", \"gpus\": $n, \"shard_updates\": $shard, \"line\": $line}" >> "$OUT"
    python - "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[1])
if "error" in d:
    print("   ", d["error"])
else:
    e = d["exchange"]
    print(f"    {d['metric']} n_gpus={d['n_gpus']} {d['value']:.3f} it/s  step {d['ms_per_step']:.2f} ms = compute {e['compute_ms_per_step']:.2f} "
          f"+ exchange {e['ms_per_iteration'] - e['ms_overlapped_per_iteration']:.2f} (overlapped {e['ms_overlapped_per_iteration']:.2f}) ms, "
          f"{e['collectives_per_iteration']:.0f} collectives of {e['bytes_per_collective'] / 1e6:.1f} MB at {e['algbw_gbps'] or 0:.1f} GB/s")
PY
}

for cfg in c4 c5; do
    for n in 1 2 4 8; do
        echo "== $cfg on $n GPU(s)"
        run "$cfg" "$n" 0
        if [ "$cfg" = c4 ] && [ "$n" -gt 1 ]; then
            echo "== $cfg on $n GPU(s), sharded updates"
            run "$cfg" "$n" 1
        fi
    done
done
echo "lines in $OUT"
