#!/bin/bash
# Same-box A/B of the round-4 tall-cell kernel (libslp_hip_r4kernel.so: `make variant NAME=r4kernel` at the round's first commit)
# against the current library: boxes differ by ~6 % among themselves, only back-to-back runs on ONE box compare kernels.
# (Since the generator takes b_upper from the chunked matrix's own product -- late in round 5 -- the round-4 library can no longer run
# bench.py's chunked workloads: its lines come out FAILED.  tools/tall_only.py still runs under SLP_LIB_VARIANT=r4kernel: the slice-level
# A/B of profiles/r05_tall_half_packets.log.)
O=gpurun_out/ab_r04
mkdir -p $O
run() {  # run <tag> <variant or ""> <bench args...>
    local tag=$1 v=$2; shift 2
    SLP_LIB_VARIANT=$v timeout 600 python bench.py --no-cpu-baseline --no-general --no-secondary "$@" > $O/$tag.json 2>> $O/err.log
    python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    ms = r.get("ms_per_product", r.get("ms_per_launch")); mst = r["spmv_transposed"].get("ms_per_product", r["spmv_transposed"].get("ms_per_launch"))
    print(f"{sys.argv[2]:28s} {d['value']:8.3f} it/s  step {d['ms_per_step']:8.3f} ms  Ax {ms:7.3f} ms frac {r['frac']:.4f}  ATy {mst:7.3f} ms  setup {d['setup_seconds']:.2f} s  peak {d['setup_breakdown']['peak_device_gb']:.1f} GB")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
    [ -n "$AB_C3" ] && run c3_r4_$rep r4kernel --config c3
    [ -n "$AB_C3" ] && run c3_new_$rep "" --config c3
    run c4_r4_k8_$rep r4kernel --chunks 8
    run c4_new_k8_$rep "" --chunks 8
    run c4_new_k16_$rep ""
done
