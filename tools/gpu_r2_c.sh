#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
timeout 300 tests/gpu_ntt3_selftest > $O/selftest.log 2>&1
grep -c ": ok" $O/selftest.log; grep -v ": ok" $O/selftest.log
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or coset or commit" 2>&1 | tail -5) > $O/pytest_ntt.log 2>&1
tail -3 $O/pytest_ntt.log
timeout 600 python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 3 --out $O/ntt_matrix.json > $O/ntt_matrix.log 2>&1
grep -v "^{\"prop" $O/ntt_matrix.log | tail -8
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 2 --out $O/ntt_matrix_prof.json > $O/prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-150 | head -14
