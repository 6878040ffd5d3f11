#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2r; mkdir -p $O
for w in 0 1 0 1; do OLA_NTT2_W4=$w timeout 300 python bench.py --steps 20 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W4=$w', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done | tee $O/w4.log
OLA_NTT2_W4=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "ntt or lde or selftest or carry" 2>&1 | tail -3
