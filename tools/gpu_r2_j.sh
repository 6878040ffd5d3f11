#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2j; mkdir -p $O
(OLA_TIMING=1 timeout 2400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s --durations=12 2>&1 | grep -v "ola-timing\]     \|amdgpu.ids" | tail -60) | tee $O/pytest_full.log
