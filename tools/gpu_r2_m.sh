#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2m; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q --durations=6 2>&1 | tail -25) | tee $O/pytest_dist.log
