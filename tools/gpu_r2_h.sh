#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2h; mkdir -p $O
timeout 300 python tools/cold_phases.py 22 > $O/cold.out 2> $O/cold.err; grep "cold\]" $O/cold.err
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -s -k "fullsize or ntt3 or config or 2p16" 2>&1 | tail -25) > $O/pytest_full.log 2>&1
tail -25 $O/pytest_full.log
