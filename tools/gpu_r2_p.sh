#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2p; mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_blake3.py -x -q -s --durations=8 2>&1 | tail -40) > $O/pytest_b3.log 2>&1
tail -42 $O/pytest_b3.log
(timeout 600 python tools/bench_prove_real.py 70000 3 --hasher blake3 --phases --json $O/prove_real_2p20_blake3.json > $O/prove_b3.log 2> $O/prove_b3_phases.log); tail -8 $O/prove_b3.log
(OLA_HASHER=blake3 OLA_VERIFY=1 timeout 600 python tools/bench_prove.py 22 3 > $O/prove22_blake3.log 2>&1); tail -4 $O/prove22_blake3.log
bash tools/pmc_ntt.sh r02 2>&1 | tail -12
cp profiles/r02_ntt_pmc.json $O/ 2>/dev/null
