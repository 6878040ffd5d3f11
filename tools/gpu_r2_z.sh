#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in cb8 cb16 cb8 cb16; do cp ab_tmp/libola_$v.so olavm_amd/lib/libola_gpu.so; echo $v; timeout 300 python bench.py --steps 30 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ntt', d['ms_per_step'], d['roofline']['frac'])"; timeout 200 python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 3 --out /tmp/m.json 2>&1 | grep -E "coset_lde8_leaf|intt" | cut -c1-100; done
