#!/bin/bash
# First run on a box with MORE THAN ONE MI355X.  Nothing in this repository has executed on two physical
# GPUs: xGMI pulls, cross-device event waits and a multi-rank RCCL communicator are exercised by construction only (ranks aliased
# onto one GPU, gloo on CPU).  This script is the order in which to find out, cheapest and most diagnostic step first; every step
# writes under gpurun_out/multi_first_run/ and the last one puts the MEASURED exchange latency and link rate next to the ASSUMED ones
# the projections in DESIGN.md section 5 / tools/proof_stats.py use (31 / 59 / 158 us per exchange at 2 / 4 / 8 ranks, 53.8 GB/s per
# link and direction).        usage (repo root, after `python -c 'import __graft_entry__ as g; g.build()'`):  bash tools/first_multi_gpu_run.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/multi_first_run; mkdir -p $O
NG=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null | tail -1)
echo "GPUs visible: $NG" | tee $O/00_gpus.txt
# OLA_KIT_DRY=1: rehearse steps 1 and 3 with the ranks aliased onto GPU 0 (checks the script, measures nothing about xGMI)
if [ "${OLA_KIT_DRY:-0}" = 1 ]; then NG=2; export OLA_KIT_DRY; O=$O/dry; mkdir -p $O
elif [ "${NG:-0}" -lt 2 ]; then echo "needs at least two GPUs; nothing run" | tee -a $O/00_gpus.txt; exit 3; fi
rocm-smi --showtopo > $O/00_topology.txt 2>&1 || true
GS=""; for g in 2 4 8; do [ "$NG" -ge $g ] && GS="$GS $g"; done

echo "=== 1. the library's all-gather by itself on distinct devices (no prover): peer pulls, then RCCL"
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/01_all_gather.txt
import json, sys, torch
sys.path.insert(0, ".")
from olavm_amd.backend import Backend
ASSUMED_US = {2: 31.0, 4: 59.0, 8: 158.0}; ASSUMED_LINK_GBS = 53.8
import os
dry = os.environ.get("OLA_KIT_DRY") == "1"
ng = 2 if dry else torch.cuda.device_count(); out = {}
for g in [g for g in (2, 4, 8) if g <= ng]:
    for carrier in ("peer", "rccl"):
        be = Backend(devices=[0] * g if dry else list(range(g)), collective=carrier)
        c = be.collective(); row = {"got": c["carrier"], "ranks": c["ranks"], "note": c["note"]}
        if c["carrier"] == carrier:
            ms_small, bad_small = be.all_gather_check(carrier, 512, reps=50)
            ms_big, bad_big = be.all_gather_check(carrier, 64 << 20, reps=5)
            recv = (64 << 20) * (g - 1)                      # bytes a rank receives, over g - 1 links at once
            row.update({"exchange_us_512B": round(ms_small * 1e3, 1), "assumed_exchange_us": ASSUMED_US[g], "wrong_bytes": bad_small + bad_big,
                        "per_link_GBs_64MB": round(recv / (g - 1) / (ms_big * 1e-3) / 1e9, 1), "assumed_per_link_GBs": ASSUMED_LINK_GBS})
        be.close(); out["%d_%s" % (g, carrier)] = row; print(g, carrier, row, flush=True)
json.dump(out, open("gpurun_out/multi_first_run/%smeasured_vs_assumed.json" % ("dry/" if dry else ""), "w"), indent=1)
PY

echo "=== 2. byte identity: the two-GPU test, then the whole multi-device and process-per-rank suites"
[ "${OLA_KIT_DRY:-0}" = 1 ] || timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k two_physical 2>&1 | tail -5 | tee $O/02_two_gpus.txt
[ "${OLA_KIT_DRY:-0}" = 1 ] || OLA_FULL_SUITE=1 timeout 2400 python -m pytest tests/test_gpu_multi.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -8 | tee $O/02_suites.txt

echo "=== 3. where the time goes on real ranks against the one-GPU projection (tools/proof_stats.py)"
for g in $GS; do
  DEV=$(seq -s ' ' 0 $((g - 1))); [ "${OLA_KIT_DRY:-0}" = 1 ] && DEV=$(printf '0 %.0s' $(seq 1 $g))
  timeout 900 python tools/proof_stats.py --log-n ${OLA_KIT_LOG_N:-22} --devices $DEV --out $O/03_proof_stats_$g.json 2>&1 | grep -v amdgpu | tail -6
done

[ "${OLA_KIT_DRY:-0}" = 1 ] && { echo "dry run: steps 2 and 4 skipped"; exit 0; }
echo "=== 4. bench.py as the driver launches it (one process per GPU, RCCL under torch.distributed), then single-process under both carriers"
for g in $GS; do
  timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port $((29600 + g)) bench.py --gpus $g --steps 10 --warmup 2 \
      > $O/04_bench_torchrun_$g.json 2> $O/04_bench_torchrun_$g.err; tail -c 600 $O/04_bench_torchrun_$g.json; echo
  for c in peer rccl; do
    OLA_COLLECTIVE=$c timeout 1200 python bench.py --gpus $g --steps 10 --warmup 2 --no-cpu-baseline --no-config4 --no-2p24 > $O/04_bench_single_process_${g}_$c.json 2> $O/04_bench_single_process_${g}_$c.err
    tail -c 400 $O/04_bench_single_process_${g}_$c.json; echo
  done
done
echo "=== done: $O/measured_vs_assumed.json holds the per-exchange latency and link rate next to the assumed ones; if they differ, update"
echo "    EXCHANGE_LATENCY_S / XGMI_LINK_BPS in tools/proof_stats.py and the selection rule of DESIGN.md section 5 (peer is the default until both carriers are timed)."
