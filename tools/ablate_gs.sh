# What bounds the windowed Gauss-Seidel sweep (k_gs_sweep_windowed) on the Potts 256^2 LP: lab builds of slp_admm.hip with
# -DSLP_GS_ABLATE=n (1: no lane / term / row-record loads, 2: no barriers, 3: one chain round, 4: no LDS reads; wrong results by
# design), timed as ADMM iterations per second on the same box.  Build the variants first:
#   for n in 1 2 3 4; do hipcc ... -DSLP_GS_ABLATE=$n -c slp_admm.hip -o /tmp/a$n.o; hipcc -shared -o ../libslp_hip_gsabl$n.so /tmp/a$n.o <other .o>; done
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for v in "" gsabl1 gsabl2 gsabl3 gsabl4 ""; do
  SLP_LIB_VARIANT=$v timeout 200 python3 tools/bench_small.py --admm-iters 1000 --cp-iters 100 --cpu-iters 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'variant': '$v' or 'product', 'admm_it_per_s': round(d['admm_gpu_it_per_s'], 1), 'bit_exact': d['admm_bit_exact_vs_oracle']}))"
done
