#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2o; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -30) > $O/pytest_all.log 2>&1
tail -32 $O/pytest_all.log
bash tools/pmc_ntt.sh r02 2>&1 | tail -12
cp profiles/r02_ntt_pmc.json $O/ 2>/dev/null
bash tools/prof.sh r02_bench_default --steps 10 --warmup 2 2>&1 | tail -3
