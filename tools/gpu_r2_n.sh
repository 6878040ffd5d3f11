#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2n; mkdir -p $O
OLA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 20 --no-cpu-baseline > $O/bench2.log 2> $O/bench2.err
tail -c 3500 $O/bench2.log; tail -5 $O/bench2.err
