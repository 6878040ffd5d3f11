#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2k; mkdir -p $O
timeout 1500 python bench.py > $O/bench.log 2> $O/bench.err; tail -c 6000 $O/bench.log; tail -5 $O/bench.err
