#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2g; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/ubench/valu_rates.hip > /dev/null 2>&1
for w in 8 2 1; do /tmp/valu_rates $w; done 2>&1 | tee $O/valu_rates.log
