#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2t; mkdir -p $O
bash tools/pmc_ntt.sh r02 2>&1 | tail -8
cp profiles/r02_ntt_pmc.json $O/ 2>/dev/null
bash tools/prof.sh r02_bench_default --steps 10 --warmup 2 2>&1 | tail -3
ls gpurun_out/r02_bench_default/ | head
