#!/bin/bash
# A variant build of libola_gpu.so whose GENERATED quotient kernels are compiled with extra flags and/or printed with other generator
# settings (environment variables read by olavm_amd/air/codegen.py, e.g. OLA_AIRQ_LIMB_LOADS_AHEAD=2), the rest taken from the current
# build.   [ENV=...] tools/build_variant_gen.sh <name> [-DFLAG ...]  ->  ab_tmp/libola_<name>.so   (for tools/gpu.sh quot_ab)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
G=$R/ab_tmp/gen_$name/gen; mkdir -p $G $R/ab_tmp/obj_$name
cp $R/olavm_amd/csrc/airq.cuh $R/olavm_amd/csrc/gl.cuh $R/ab_tmp/gen_$name/
(cd $R && python -c "
import sys; sys.path.insert(0, '.')
from olavm_amd.air import codegen
codegen.write_default('$G')")
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread"
pids=()
for u in $G/airq_*.hip; do
  hipcc $F "$@" -c -o $R/ab_tmp/obj_$name/$(basename ${u%.hip}).o $u & pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
objs=$(ls $R/olavm_amd/lib/obj/*.o | grep -v '/airq_')
hipcc $F -shared -o $R/ab_tmp/libola_$name.so $objs $R/ab_tmp/obj_$name/*.o
rm -rf $R/ab_tmp/obj_$name $R/ab_tmp/gen_$name
echo "built ab_tmp/libola_$name.so"
