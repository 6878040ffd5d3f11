#!/bin/bash
# usage (on the GPU box, from repo root): tools/prof_cmd.sh <name> <command...>
# rocprofv3 kernel-trace + stats of an arbitrary command; CSVs land in gpurun_out/<name>/
set -u
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- "$@" > $out/cmd.log 2>&1)
tail -4 $out/cmd.log
find $out -name "*kernel_stats.csv" | head -1 | xargs -r head -40
