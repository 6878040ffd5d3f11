#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq.sh <name> <python script and args...>
# SQ issue/stall counters per kernel (one pass, 8 SQ slots): where do the wave cycles go?
set -u
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $out -o p -- "$@" > $out/run.log 2>&1)
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-70:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:8]
for k, c in rows:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:70s} n={cnt[k]:4d} wave_cyc={wc:.3e} wait_any={c.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
          f"active_any={c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} active_valu={c.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} "
          f"valu_insts={c.get('SQ_INSTS_VALU',0):.3e} lds_conf={c.get('SQ_LDS_BANK_CONFLICT',0)/wc:5.3f} busy={c.get('SQ_BUSY_CYCLES',0):.3e}")
PY
