#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2l; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_stark.py tests/test_gpu_parity.py tests/test_abi.py -m gpu -x -q -k "phase_entry or lean or poseidon_kats or ntt3_sizes or config or abi" 2>&1 | tail -15) | tee $O/pytest.log
OLA_TIMING=1 python tools/cold_phases.py 20 2>&1 | grep -v "table [1-9]\|table 1[01]" | head -60 | tee $O/timing_names.log
