#!/bin/bash
# Tall cells off the metric's density: 2.5e6 rows, ~2.5e9 stored entries, columns = 1000 / density; 4096-column strips forced
# 2048 and 1024 forced, and the width the build picks by its cost model (slp_tall.hip, tall_build).  One box, back to back.
#   tools/tall_density_sweep.sh > profiles/rNN_tall_density_sweep.log
cd "$(dirname "$0")/.."
for d in 1e-4 1.5e-4 2e-4 3e-4 4.5e-4; do
  cols=$(python3 -c "print(int(round(1000/$d)))")
  for c in 4096 2048 1024 auto; do
    if [ $c = auto ]; then unset SLP_TALL_C; else export SLP_TALL_C=$c; fi
    echo "density $d columns $cols strips $c: $(timeout 300 python3 tools/tall_only.py 10 2500000 $cols $d 2>&1 | tail -1)"
  done
done
