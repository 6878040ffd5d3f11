#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2u; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_stark.py -x -q -k "device_resident" 2>&1 | tail -5)
(timeout 400 python -m pytest tests/test_gpu_distributed.py -x -q -k "blake3" 2>&1 | tail -5)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3)
(timeout 600 python bench.py --steps 10 --warmup 2 --no-2p24 --no-cpu-baseline > $O/bench.log 2> $O/bench.err); python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2u/bench.log').read().strip().splitlines()[-1])
print('ntt', d['ms_per_step'], d['roofline']['frac'], d['roofline']['valu'] and d['roofline']['valu']['insts_per_element'])
p=d['prove']; print('prove', p['seconds'], p['verified'], p.get('tables_resident_in_hbm'), p['blake3_config']['seconds'], p['cold_process']['cold_over_warm'], p['cold_process_without_reserve']['cold_over_warm'])
r=d['prove_real_execution']; print('real', r['seconds'], r['verified'], r['blake3_config']['seconds'])
PY
tail -3 $O/bench.err
