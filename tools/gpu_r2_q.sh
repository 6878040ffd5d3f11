#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2q; mkdir -p $O
(timeout 60 ./tests/gpu_glasm_selftest 2000 > $O/glasm.log 2>&1); cat $O/glasm.log
(time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lookup.py -x -q 2>&1 | tail -5) > $O/pytest_parity.log 2>&1; tail -8 $O/pytest_parity.log
(timeout 600 python bench.py --steps 10 --warmup 2 --no-2p24 --no-cpu-baseline > $O/bench.log 2> $O/bench.err); cut -c1-2600 $O/bench.log
