#!/bin/bash
# usage (GPU box, repo root): tools/kt_small.sh <hasher>   -- kernel trace of one README-shape proof, compacted to name,start,end
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/kt
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $R/tools/small_tables.py ${1:-blake3} 1 > $out/run.log 2>&1 < /dev/null
cd $R
f=$(find $out -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no trace"; tail -5 $out/run.log; exit 1; }
python3 - "$f" <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with gzip.open("gpurun_out/kt/trace_compact.csv.gz", "wt") as f:
    for r in rows:
        f.write("%s,%s,%s\n" % (r["Kernel_Name"].split("(")[0][:60].replace(",", ";"), r["Start_Timestamp"], r["End_Timestamp"]))
print(len(rows), "kernel records")
PY
rm -f "$f"
grep -v amdgpu $out/run.log | head -16
