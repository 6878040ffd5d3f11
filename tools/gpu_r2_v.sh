#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2v; mkdir -p $O
(time timeout 1000 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -16) > $O/pytest_all.log 2>&1
tail -18 $O/pytest_all.log
(OLA_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 20 > $O/bench2.log 2> $O/bench2.err); python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r2v/bench2.log').read().splitlines() if l.startswith('{')][-1])
    print('N=2 dry run:', d['value'], d['n_gpus'], {k: (v.get('seconds') or v.get('ms')) if isinstance(v, dict) else v for k, v in d.items() if 'sharded' in k})
    print({k: v.get('verified') for k, v in d.items() if isinstance(v, dict) and 'verified' in v})
except Exception as e:
    print('bench2 failed', e); print(open('gpurun_out/r2v/bench2.err').read()[-1500:])
PY
