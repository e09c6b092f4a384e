for rep in 1 2; do for v in 0 1; do echo "SLP_TALL_ARITH=$v $(SLP_TALL_ARITH=$v timeout 400 python bench.py --no-cpu-baseline --no-secondary --no-general 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('it/s', round(d['value'],3), 'step', round(d['ms_per_step'],2), 'Ax', round(r['ms_per_product'],3), 'ATy', round(r['spmv_transposed']['ms_per_product'],3), 'frac', round(r['frac'],4), 'obj', d['objective_after_run'])")"; done; done
