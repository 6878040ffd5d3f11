#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ntt', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
timeout 200 python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 3 --out /tmp/m.json 2>&1 | grep -E "\"op\"|all checks|false" | cut -c1-120
timeout 200 python tools/bench_prove.py 22 3 2>&1 | grep prove_with | tail -2
