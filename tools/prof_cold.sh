#!/bin/bash
# usage (GPU box, repo root): tools/prof_cold.sh <name> [count]
# Where does the first proof on a fresh context lose its time (DESIGN.md "Open item")?  HIP API + kernel trace of
# tools/ubench/fresh_buffers.py (first call / same arrays / fresh copies), no counters; summaries land in gpurun_out/<name>/.
set -u
name=$1; count=${2:-599185}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
(cd $R && OLA_TIMING=1 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $out -o p -- python tools/ubench/fresh_buffers.py $count > $out/cmd.log 2>&1)
grep -v "^\[ola-timing\]    " $out/cmd.log | tail -70
for f in $(find $out -name "*hip_api_stats.csv" -o -name "*domain_stats.csv" | head -2); do echo "== $f"; head -15 $f; done
