#!/bin/bash
# round 2, first GPU call: correctness of the new NTT path, its speed under both grid layouts next to the old path, access-pattern ubench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or coset or commit" 2>&1 | tail -15) > $O/pytest_ntt.log 2>&1
for cm in 1 0; do
  OLA_NTT3_COL_MAJOR=$cm timeout 300 python bench.py --steps 10 --warmup 2 --no-prove --no-cpu-baseline > $O/bench_ntt3_cm$cm.log 2>&1
done
OLA_NTT3=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-prove --no-cpu-baseline > $O/bench_ntt2.log 2>&1
timeout 600 python tools/bench_ntt_matrix.py --log-n 20 22 --cols 94 --reps 3 --out $O/ntt_matrix.json > $O/ntt_matrix.log 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/segment_copy tools/ubench/segment_copy.hip > /dev/null 2>&1 && timeout 120 /tmp/segment_copy > $O/segment_copy.log 2>&1
tail -3 $O/pytest_ntt.log; tail -c 600 $O/bench_ntt3_cm1.log; echo; tail -c 400 $O/bench_ntt3_cm0.log; echo; tail -c 400 $O/bench_ntt2.log; echo; grep -v "^{\"prop" $O/ntt_matrix.log | tail -12; cat $O/segment_copy.log
