#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "carry or poseidon or merkle or ntt_evaluate" 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
