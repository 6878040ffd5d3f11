#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r2d}
mkdir -p $O
cd $R
timeout 300 tests/gpu_ntt3_selftest > $O/selftest.log 2>&1
grep -c ": ok" $O/selftest.log; grep -v ": ok" $O/selftest.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 2 --out $O/ntt_matrix_prof.json > $O/prof.log 2>&1)
grep "\"op\"" $O/prof.log | cut -c1-110
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-130 | grep "ntt3_pass\|Name"
