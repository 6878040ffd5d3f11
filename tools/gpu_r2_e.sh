#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/pmc_sq.sh r2e_sq python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 1 --out gpurun_out/r2e_sq/m.json
# second counter set: LDS / memory instruction counts and waits
out=$R/gpurun_out/r2e_sq2; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && cd $R && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $out -o p -- python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 1 --out $out/m.json > $out/run.log 2>&1)
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:60s} n={cnt[k]:3d} wave_cyc={wc:.3e} waves={c.get('SQ_WAVES',0):.3e} lds_insts={c.get('SQ_INSTS_LDS',0):.3e} vmem_rd={c.get('SQ_INSTS_VMEM_RD',0):.3e} vmem_wr={c.get('SQ_INSTS_VMEM_WR',0):.3e} salu={c.get('SQ_INSTS_SALU',0):.3e} wait_lds={c.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} active_lds={c.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f}")
PY
