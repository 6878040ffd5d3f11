#!/bin/bash
# usage (on the GPU box, from repo root): tools/pmc.sh <name>
# HBM traffic of the NTT kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in separate passes, kernel-trace only (no sys/hip/hsa tracing next to --pmc).  The same process
# (tools/pmc_workload.py) commits a 94 x 2^20 table: its canonicalize_kernel reads and writes every element once with this code base's 8 B/lane access
# pattern, a known byte count that calibrates the two counters.
set -u
name=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd $R && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o p -- \
      python tools/pmc_workload.py > $out/$ctr.log 2>&1)
  f=$(find $out/$ctr -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$ctr" <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != ctr: continue
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{ctr:10s} {k:62s} dispatches {n:5d}  total {v:16.1f}  per-dispatch {v/n:14.1f}")
PY
done
