#!/bin/bash
# A variant build of libola_gpu.so for same-box A/B runs (tools/gpu.sh lib_ab / sbox_ab / quot_ab): the main translation unit compiled with
# extra flags, linked with the objects of the current build.   tools/build_variant.sh <name> [-DFLAG ...]  ->  ab_tmp/libola_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
mkdir -p $R/ab_tmp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread"
hipcc $F "$@" -c -o $R/ab_tmp/ola_gpu_$name.o $R/olavm_amd/csrc/ola_gpu.hip
objs=$(ls $R/olavm_amd/lib/obj/*.o | grep -v '/ola_gpu.o$')
hipcc $F -shared -o $R/ab_tmp/libola_$name.so $R/ab_tmp/ola_gpu_$name.o $objs
rm -f $R/ab_tmp/ola_gpu_$name.o
echo "built ab_tmp/libola_$name.so"
