#!/bin/bash
# Lab library libslp_hip_tbprof.so: the shipped objects + slp_tall.hip with tools/lab/patches/slp_tall_lab_switches.patch and
# -DSLP_TALL_BUILD_PROF (k_tall_build sums the time between its marks over all cells; tall_build prints us per cell and mark).
#   bash tools/lab/build_tbprof.sh && SLP_LIB_VARIANT=tbprof python tools/tall_only.py 2
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/pysparselp_amd/csrc
make -C $C > /dev/null
T=$(mktemp -d)
cp $C/slp_tall.hip $T/slp_tall.hip
(cd $T && patch -p3 < $R/tools/lab/patches/slp_tall_lab_switches.patch > /dev/null)
cp $T/slp_tall.hip $C/slp_tall_tbprof_tmp.hip
trap "rm -f $C/slp_tall_tbprof_tmp.hip; rm -rf $T" EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I/opt/rocm/include -DSLP_LAB_VARIANT -DSLP_TALL_BUILD_PROF \
    -c $C/slp_tall_tbprof_tmp.hip -o $T/slp_tall.o
objs=$(ls $C/slp_*.o | grep -v "abl\|/slp_tall.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/pysparselp_amd/libslp_hip_tbprof.so $objs $T/slp_tall.o -ldl -Wl,-rpath,/opt/rocm/lib
ls -la $R/pysparselp_amd/libslp_hip_tbprof.so
