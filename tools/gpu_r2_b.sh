#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
timeout 300 tests/gpu_ntt3_selftest > $O/selftest.log 2>&1
cat $O/selftest.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --steps 5 --warmup 1 --no-prove --no-cpu-baseline > $O/prof_bench.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12
hipcc --offload-arch=gfx950 -O3 -o /tmp/segment_copy tools/ubench/segment_copy.hip > /dev/null 2>&1 && timeout 120 /tmp/segment_copy > $O/segment_copy.log 2>&1
cat $O/segment_copy.log
