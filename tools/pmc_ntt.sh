#!/bin/bash
# usage (GPU box, repo root): tools/pmc_ntt.sh <tag>      e.g. tools/pmc_ntt.sh r02
# HBM traffic and VALU instruction counts of the NTT pass kernels, measured as MI355X_MICROARCH.md prescribes (one counter group per
# pass: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; --kernel-trace only) over tools/pmc_workload.py (3 forward NTTs of 94 x 2^22 + one
# 94 x 2^20 commitment whose kernels have known byte counts: calibration).  Writes profiles/<tag>_ntt_pmc.json, which bench.py reads
# (it carries the hash of the kernel sources; bench.py reports the figures only while the sources are unchanged).
set -u
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU TCC_HIT_sum TCC_MISS_sum; do
  (cd $R && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o p -- python tools/pmc_workload.py > $out/$ctr.log 2>&1)
done
# the Infinity-Cache blocking experiment of round 3 (ntt2_run, OLA_NTT2_GROUP_MB): the same transforms in column groups of 96 MB;
# only with OLA_PMC_G96=1 (it doubles the box time and its answer is on record in profiles/r03_ntt_pmc.json)
[ "${OLA_PMC_G96:-0}" = 1 ] && for ctr in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  (cd $R && OLA_NTT2_GROUP_MB=96 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/G96_$ctr -o p -- python tools/pmc_workload.py > $out/G96_$ctr.log 2>&1)
done
cd $R && python3 tools/pmc_ntt_json.py $out profiles/${tag}_ntt_pmc.json
cp profiles/${tag}_ntt_pmc.json $out/ 2>/dev/null || true
