#!/bin/bash
# The N = 2 / 4 / 8 launches of bench.py exactly as the driver issues them, but with all ranks on ONE GPU and the host TCP
# transport instead of RCCL (which refuses two ranks on one device): the row partition, the per-rank kernel choices
# (strip splits for few row blocks), the packed exchange and rank 0's output at FULL BASELINE config 3 size.  Rates are
# meaningless (ranks time-share one GPU, all-reduces go through Python); the objective after the run must equal the
# single-process value.   bash tools/multirank_one_gpu.sh > gpurun_out/multirank.txt
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export SLP_DEVICE=0 SLP_COMM_TRANSPORT=host
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-general | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1', d['objective_after_run'], d['value'])"
for N in 2 4 8; do
  MASTER_PORT=$((29700 + N * 20)) timeout 900 python3 bench.py --gpus $N --steps 6 --warmup 2 --no-cpu-baseline --no-general 2> gpurun_out/multirank_$N.err | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('N=$N', d['objective_after_run'], d['config']['collectives_per_iteration'], d['config']['nnz'], d['device_memory'])"
  tail -3 gpurun_out/multirank_$N.err
done
