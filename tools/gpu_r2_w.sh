#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2w; mkdir -p $O
for w in 0 1 0 1; do echo "PRE_W3=$w"; OLA_NTT2_PRE_W3=$w timeout 200 python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 3 --out $O/m$w.json 2>&1 | grep -E "coset_lde8_leaf|all checks|ok\": false" | cut -c1-140; done
for w in 0 1; do echo "PRE_W3=$w"; OLA_NTT2_PRE_W3=$w timeout 200 python tools/bench_prove.py 22 3 2>&1 | grep prove_with | tail -2; done
