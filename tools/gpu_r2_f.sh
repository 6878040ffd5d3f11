#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2f; mkdir -p $O
for w in 2 3; do
  echo "== WPS=$w selftest"; OLA_NTT3_WPS=$w timeout 300 tests/gpu_ntt3_selftest | grep -v ": ok"
  for cm in 1 0; do echo "== WPS=$w COL_MAJOR=$cm"; OLA_NTT3_WPS=$w OLA_NTT3_COL_MAJOR=$cm timeout 300 tests/gpu_ntt3_bench; done
done 2>&1 | tee $O/bench.log
