#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2t2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "ntt or lde or commit or carry" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_stark.py -x -q -k "twelve_table_all or memory_lean or device_resident or long_fib" 2>&1 | tail -2
bash tools/pmc_ntt.sh r02 2>&1 | tail -8
cp profiles/r02_ntt_pmc.json $O/ 2>/dev/null
bash tools/prof.sh r02_bench_default --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-300
