#!/bin/bash
# usage (on the GPU box, from repo root): tools/prof.sh <name> <bench args...>
# rocprofv3 kernel-trace + stats of bench.py; CSVs land in gpurun_out/<name>/
set -u
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $R/bench.py "$@" > $out/bench.log 2>&1
tail -2 $out/bench.log
find $out -name "*kernel_stats.csv" | head -1 | xargs -r head -25
