#!/bin/bash
# usage (GPU box, repo root): tools/pmc_proof.sh <tag> [hasher]     e.g. tools/pmc_proof.sh r04 blake3
# HBM traffic and VALU instruction counts of EVERY kernel of one 2^22-row proof (prove_with_traces, 12 tables), collected as
# MI355X_MICROARCH.md prescribes: one counter per pass (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU), --kernel-trace only, over
# tools/bench_prove.py 22 1 (one proof; the first call of a context, so buffers are fresh: traffic does not depend on that).
# Writes profiles/<tag>_proof_pmc_<hasher>.txt: per kernel family, dispatches, traffic per proof (FETCH x 2 + WRITE, the gfx950
# correction of the guide), VALU wave-instructions per proof.
set -u
tag=${1:-r04}; hasher=${2:-blake3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_proof_${tag}_$hasher
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  (cd $R && OLA_HASHER=$hasher timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o p -- python tools/bench_prove.py 22 1 > $out/$ctr.log 2>&1)
done
cd $R && python3 tools/pmc_proof_table.py $out $hasher > $out/table.txt && cp $out/table.txt profiles/${tag}_proof_pmc_$hasher.txt && cat $out/table.txt
