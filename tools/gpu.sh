#!/bin/bash
# The GPU-box runs of a round, one named step per argument (gpurun -- 'bash tools/gpu.sh step1 step2 ...'); everything a step
# prints or measures lands under gpurun_out/<step>/.  Replaces the one-off scripts of earlier rounds.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export TMPDIR=/tmp
for step in "$@"; do
  O=gpurun_out/$step; mkdir -p "$O"
  echo "=== $step"
  case $step in
    upload)      # trace upload: the box's H2D ceilings, the tests of the per-column entry point, the paths alternated inside the 2^22-row proof
                 hipcc --offload-arch=gfx950 -O3 -pthread -o /tmp/h2d_rates tools/ubench/h2d_rates.hip 2>/dev/null && timeout 300 /tmp/h2d_rates 4 2>&1 | grep -v amdgpu | tee $O/h2d_rates.txt
                 timeout 900 python -m pytest tests/test_gpu_upload.py tests/test_gpu_host_api.py -x -q 2>&1 | tail -15 | tee $O/pytest.log
                 timeout 900 python tools/bench_upload.py 22 blake3 3 2>&1 | grep -v amdgpu | tee $O/ab_blake3.txt
                 OLA_HASHER=blake3 OLA_TIMING=1 timeout 300 python tools/bench_prove.py 22 2 2> $O/phases_blake3.txt | tail -2 ;;
    sbox_ab)     # Poseidon S-box multiplication: ab_tmp/libola_<v>.so alternated; commitment of 94 x 2^22 (leaf + tree hashing dominate) and the whole Poseidon-config proof
                 cp olavm_amd/lib/libola_gpu.so ab_tmp/libola_cur.so
                 for v in ${AB_VARIANTS:-cur onechain cur onechain}; do cp ab_tmp/libola_$v.so olavm_amd/lib/libola_gpu.so; echo "-- $v"
                   timeout 300 python tools/bench_commit.py 22 94 3 2>/dev/null | tail -2
                   timeout 300 python tools/bench_prove.py 22 3 2>/dev/null | tail -2
                 done 2>&1 | grep -v amdgpu | tee $O/ab.txt; cp ab_tmp/libola_cur.so olavm_amd/lib/libola_gpu.so ;;
    multi)       timeout 900 python -m pytest tests/test_gpu_multi.py -x -q 2>&1 | tail -15 | tee $O/pytest.log ;;
    dist)        timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_multi.py -x -q 2>&1 | tail -15 | tee $O/pytest.log ;;
    start)       # round 6: ola_gpu_init taken apart (OLA_TIMING prints the split), the early hook's tests, the per-pass timing entry point
                 timeout 600 python -m pytest tests/test_gpu_start.py -x -q 2>&1 | tail -8 | tee $O/pytest.log
                 for i in 1 2; do OLA_TIMING=1 timeout 300 python -c "
import time, sys
sys.path.insert(0, '.')
import torch
from olavm_amd import backend as B
t0 = time.perf_counter(); B.load_library(); t1 = time.perf_counter()
be = B.Backend(device=0); t2 = time.perf_counter()
print('dlopen %.1f ms, ola_gpu_init %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
" 2>&1 | grep -v amdgpu; done | tee $O/init_split.txt
                 OLA_TIMING=1 timeout 300 python -c "
import time, sys
sys.path.insert(0, '.')
import torch
from olavm_amd import backend as B
from olavm_amd.air import ola_tables as T
blob = T.ola_stark().blob()
B.load_library(); t0 = time.perf_counter(); B.warmup(0, airset=blob); t1 = time.perf_counter(); ms = B.warmup_wait(); t2 = time.perf_counter()
be = B.Backend(device=0); t3 = time.perf_counter()
print('ola_gpu_warmup call %.2f ms, thread %.1f ms, ola_gpu_init after it %.1f ms' % ((t1 - t0) * 1e3, ms, (t3 - t2) * 1e3))
" 2>&1 | grep -v amdgpu | tee -a $O/init_split.txt ;;
    suite)       timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log ;;
    smoke)       timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -5 | tee $O/smoke.log ;;
    bench)       timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; cp bench_details.json $O/ 2>/dev/null ;;
    bench2)      timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err ;;
    pool)        OLA_TIMING=0 timeout 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_host_api.py -x -q 2>&1 | tail -8 | tee $O/pytest.log
                 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/handover.txt
import sys, time
sys.path.insert(0, ".")
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend
blob = T.ola_stark().blob()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=22, log_n_mem=22)
import os
for share in ("1", "0"):
    os.environ["OLA_POOL_SHARE"] = share
    a = Backend(device=0); b = Backend(device=0, hasher="blake3")
    ts = []
    for be in (a, a, b, b, a, b):
        t0 = time.perf_counter(); be.prove_with_traces(blob, traces, params, compress); ts.append(time.perf_counter() - t0)
    print("OLA_POOL_SHARE=%s  poseidon first %.3f warm %.3f | blake3 first %.3f warm %.3f | back to poseidon %.3f, to blake3 %.3f | pools %.1f + %.1f GB"
          % (share, ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], a.memory_stats()["reserved"] / 1e9, b.memory_stats()["reserved"] / 1e9))
    a.close(); b.close()
PY
                 ;;
    torchrun2)   # the launcher path of `bench.py --gpus N` (what the driver's scaling run uses), dry: two ranks share the one GPU, gloo carries the collectives
                 OLA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
                 tail -c 2500 $O/bench.json; tail -3 $O/bench.err ;;
    stats)       timeout 600 python tools/proof_stats.py --out $O/proof_stats.json 2>&1 | tail -8 ;;
    stats_real)  timeout 600 python tools/proof_stats.py --real --out $O/proof_stats_real.json 2>&1 | tail -8 ;;
    stats_multi) timeout 600 python tools/proof_stats.py --log-n 20 --devices 0 0 0 0 0 0 0 0 --out $O/proof_stats.json 2>&1 | tail -8 ;;
    stats_multi22) timeout 900 python tools/proof_stats.py --log-n 22 --devices 0 0 0 0 0 0 0 0 --out $O/proof_stats.json 2>&1 | tail -8 ;;
    stats_multi23) # dry run towards config 5 on ONE GPU: per-rank pool high-water marks (8 aliased ranks at 2^23 rows do not fit 288 GB: OLA_E_OOM, clean)
                 timeout 900 python tools/proof_stats.py --log-n 22 --hashers blake3 --devices 0 0 0 0 0 0 0 0 --out $O/proof_stats_8x2p22.json 2>&1 | tail -3
                 timeout 900 python tools/proof_stats.py --log-n 23 --hashers blake3 --devices 0 0 0 0 --out $O/proof_stats_4x2p23.json 2>&1 | tail -3 ;;
    sq_passes)   bash tools/pmc_sq.sh sq_passes python tools/pmc_workload_lde.py 2>&1 | tail -12 | tee $O/sq.txt
                 cd /tmp; for ctr in "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
                   (cd $R && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$O/lds -o p -- python tools/pmc_workload_lde.py > $R/$O/lds.log 2>&1); done; cd $R
                 f=$(find $O/lds -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 - "$f" <<'PY' | tee $O/lds.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-52:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:52s} n={cnt[k]:3d} " + " ".join(f"{n[3:]}={c.get(n, 0) / wc:6.3f}" for n in ("SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INST_CYCLES_VMEM")) +
          f" lds_insts={c.get('SQ_INSTS_LDS', 0):.3e} vmem_rd={c.get('SQ_INSTS_VMEM_RD', 0):.3e} vmem_wr={c.get('SQ_INSTS_VMEM_WR', 0):.3e}")
PY
                 ;;
    banks)       hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/ubench/valu_rates.hip 2>/dev/null
                 for w in 8 4 3; do /tmp/valu_rates $w 46; done 2>&1 | grep -v amdgpu | tee $O/banks.txt ;;
    quot_check)  timeout 1200 python -m pytest tests/test_gpu_stark.py tests/test_gpu_blake3.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
                 for h in blake3 poseidon; do AB_HASHER=$h timeout 300 python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend
blob = T.ola_stark().blob()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=22, log_n_mem=22)
be = Backend(device=0, hasher=os.environ["AB_HASHER"])
be.proof_stats(enable=True)
best = None
for _ in range(4):
    be.prove_with_traces(blob, traces, params, compress)
    st, ph = be.proof_stats(), be.phase_stats()
    if best is None or st["wall_ms"] < best[0]:
        best = (st["wall_ms"], ph["quotient"][0], ph["open_eval"][0], ph["lde"][0], ph["leaf_hash"][0])
print(os.environ["AB_HASHER"], "wall %.1f ms, quotient kernels %.2f ms, opening evaluations %.2f ms, LDE %.2f ms, leaves %.2f ms" % best)
PY
                 done 2>&1 | grep -v amdgpu | tee $O/quot.txt ;;
    openings)    # round 6: eval_points_wide_kernel, fold16_kernel, leaf_b3_ext16_kernel against the kernels they replace (environment switches, same box, alternating)
                 for h in blake3 poseidon; do for v in 0 1 0 1; do AB_HASHER=$h OLA_EVAL_WIDE=$v OLA_FOLD16=$v OLA_LEAF_EXT_STAGED=$v timeout 300 python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend
blob = T.ola_stark().blob()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=22, log_n_mem=22)
be = Backend(device=0, hasher=os.environ["AB_HASHER"])
be.proof_stats(enable=True)
best = None
import hashlib
for _ in range(4):
    proof = be.prove_with_traces(blob, traces, params, compress)
    st, ph = be.proof_stats(), be.phase_stats()
    if best is None or st["wall_ms"] < best[0]:
        best = (st["wall_ms"], ph["open_eval"][0], ph["fri_fold"][0], ph["leaf_hash"][0], ph["quotient"][0])
print(os.environ["AB_HASHER"], "new" if os.environ["OLA_EVAL_WIDE"] == "1" else "old", "wall %.1f ms, opening evaluations %.2f ms, folds %.3f ms, leaves %.2f ms, quotient %.2f ms" % best,
      "proof sha256", hashlib.sha256(bytes(proof)).hexdigest()[:16])
PY
                 done; done 2>&1 | grep -v amdgpu | tee $O/ab.txt ;;
    quot_ab)     # round 6: lazy sums of the quotient kernels as 22-bit limb products (Acc3) against the 160-bit sums (ab_tmp/libola_acc160.so, tools/build_variant_gen.sh)
                 timeout 1200 python -m pytest tests/test_gpu_stark.py tests/test_gpu_blake3.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
                 cp olavm_amd/lib/libola_gpu.so ab_tmp/libola_cur.so
                 for v in ${AB_VARIANTS:-acc160 cur acc160 cur}; do cp ab_tmp/libola_$v.so olavm_amd/lib/libola_gpu.so; AB_NAME=$v AB_HASHER=blake3 timeout 300 python - <<'PY'
import os, sys, hashlib
sys.path.insert(0, ".")
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend
blob = T.ola_stark().blob()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=22, log_n_mem=22)
be = Backend(device=0, hasher=os.environ["AB_HASHER"])
be.proof_stats(enable=True)
best = None
for _ in range(4):
    proof = be.prove_with_traces(blob, traces, params, compress)
    st, ph = be.proof_stats(), be.phase_stats()
    if best is None or st["wall_ms"] < best[0]:
        best = (st["wall_ms"], ph["quotient"][0])
print(os.environ["AB_NAME"], "wall %.1f ms, quotient kernels %.2f ms" % best, "proof sha256", hashlib.sha256(bytes(proof)).hexdigest()[:16])
PY
                 done 2>&1 | grep -v amdgpu | tee $O/ab.txt; cp ab_tmp/libola_cur.so olavm_amd/lib/libola_gpu.so ;;
    merkle_ab)   # round 6: eight Blake3 Merkle levels per launch (merkle_levels_b3_kernel) against one launch per level (OLA_MERKLE_FUSED_LEVELS=0)
                 [ "${AB_SKIP_TESTS:-0}" = 1 ] || timeout 1200 python -m pytest tests/test_gpu_blake3.py tests/test_gpu_fullsize.py -x -q -k "blake3 or b3 or commitment" 2>&1 | tail -3 | tee $O/pytest.log
                 for v in ${AB_LEVELS:-0 2 3 4 0 2 3 4}; do AB_HASHER=blake3 OLA_MERKLE_FUSED_LEVELS=$v timeout 300 python - <<'PY'
import os, sys, hashlib
sys.path.insert(0, ".")
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend
blob = T.ola_stark().blob()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=22, log_n_mem=22)
be = Backend(device=0, hasher=os.environ["AB_HASHER"])
be.proof_stats(enable=True)
best = None
for _ in range(4):
    proof = be.prove_with_traces(blob, traces, params, compress)
    st, ph = be.proof_stats(), be.phase_stats()
    if best is None or st["wall_ms"] < best[0]:
        best = (st["wall_ms"], ph["merkle_levels"][0], ph["leaf_hash"][0])
print("levels per launch:", os.environ["OLA_MERKLE_FUSED_LEVELS"], "wall %.1f ms, Merkle levels %.2f ms, leaves %.2f ms" % best, "proof sha256", hashlib.sha256(bytes(proof)).hexdigest()[:16])
PY
                 done 2>&1 | grep -v amdgpu | tee $O/ab.txt ;;
    selftest_fault) # the device self-test must notice a fault injected into the limb-product sums (ab_tmp/libola_fault.so: tools/build_variant.sh fault -DOLA_SELFTEST_FAULT_ACC3)
                 cp olavm_amd/lib/libola_gpu.so ab_tmp/libola_cur.so
                 for v in cur fault; do cp ab_tmp/libola_$v.so olavm_amd/lib/libola_gpu.so; timeout 300 python -c "
import sys; sys.path.insert(0, '.')
from olavm_amd.backend import Backend
be = Backend(device=0); print('$v library: ola_gpu_selftest(2^31) mismatches =', be.selftest(1 << 31)); be.close()" 2>&1 | grep -v amdgpu; done | tee $O/selftest.txt
                 cp ab_tmp/libola_cur.so olavm_amd/lib/libola_gpu.so ;;
    phases)      OLA_TIMING=1 timeout 600 python tools/bench_prove.py 22 2 2> $O/phases.txt | tail -3; OLA_HASHER=blake3 OLA_TIMING=1 timeout 600 python tools/bench_prove.py 22 2 2> $O/phases_blake3.txt | tail -3 ;;
    ntt_group)   # Infinity-Cache blocking of the transforms: working-set target in MB (0 = whole batch per launch)
                 for mb in 0 32 64 96 128 192; do
                   echo "-- OLA_NTT2_GROUP_MB=$mb"
                   OLA_NTT2_GROUP_MB=$mb timeout 300 python bench.py --steps 20 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ntt ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
                   OLA_NTT2_GROUP_MB=$mb timeout 300 python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 3 --out $O/m_$mb.json 2>&1 | grep -E "lde|intt|ntt" | cut -c1-120
                 done 2>&1 | tee $O/sweep.txt ;;
    canon_passes) # the canonical-arithmetic passes (OLA_NTT2_TFORM=0: the A/B control, and the path of transforms beyond 2^28) against the oracle
                 OLA_NTT2_TFORM=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "ntt or lde or NTT or coset or commit" 2>&1 | tail -4 | tee $O/pytest.log ;;
    tform_ab)    # T-form passes against the canonical-arithmetic passes, same box, alternating (OLA_NTT2_TFORM=0/1)
                 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "ntt or lde or NTT or coset" 2>&1 | tail -8 | tee $O/pytest.log
                 for v in 0 1 0 1; do echo "-- OLA_NTT2_TFORM=$v"
                   OLA_NTT2_TFORM=$v timeout 300 python bench.py --steps 30 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ntt ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
                   OLA_NTT2_TFORM=$v timeout 300 python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 5 --out $O/m_$v.json 2>&1 | grep -E '"op"' | cut -c1-110
                 done 2>&1 | tee $O/ab.txt ;;
    tform_prof)  # per-kernel durations of the two pass families (NTT, iNTT and LDE of 94 x 2^22)
                 for v in 0 1; do cd /tmp; OLA_NTT2_TFORM=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$v -o p -- python $R/tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 5 --out $R/$O/m_$v.json > /dev/null 2> $R/$O/err$v.txt; cd $R
                   f=$(find $O/prof$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$v.csv && head -14 $O/kernel_stats_$v.csv | cut -c1-170; done ;;
    ntt_tests)   timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "ntt or lde or NTT or coset" 2>&1 | tail -8 | tee $O/pytest.log ;;
    lib_ab)      # same-box A/B of two builds of the library: ab_tmp/libola_<name>.so, alternating (take ab_tmp/ out of .gpurunignore for the call)
                 cp olavm_amd/lib/libola_gpu.so ab_tmp/libola_cur.so
                 for v in ${AB_VARIANTS:-old new old new}; do cp ab_tmp/libola_$v.so olavm_amd/lib/libola_gpu.so; echo "-- $v"
                   timeout 300 python bench.py --steps 30 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ntt ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
                   timeout 300 python tools/bench_ntt_matrix.py --log-n 22 --cols 94 --reps 5 --out $O/m_$v.json 2>&1 | grep -E '"op"' | cut -c1-110
                 done 2>&1 | tee $O/ab.txt; cp ab_tmp/libola_cur.so olavm_amd/lib/libola_gpu.so ;;
    pmc)         bash tools/pmc_ntt.sh r05 2>&1 | tail -30 ;;
    quot_final)  # after a change to the quotient kernels: the proof tests, the full-size and multi-rank ones, smoke
                 timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_pool.py tests/test_gpu_host_api.py -x -q > $O/pytest_full.log 2>&1; grep -E ' passed| failed| error' $O/pytest_full.log | tee $O/pytest.log
                 ;;
    sq_proof)    # SQ issue / stall counters of every kernel of one 2^22-row proof (Blake3 configuration)
                 OLA_HASHER=blake3 bash tools/pmc_sq.sh sq_proof python tools/bench_prove.py 22 1 2>&1 | tail -10 | tee $O/sq.txt ;;
    pmc_proof)   bash tools/pmc_proof.sh r05 blake3 2>&1 | tail -34; bash tools/pmc_proof.sh r05 poseidon 2>&1 | tail -34 ;;
    cold_preheat) OLA_COLD_PREHEAT=8 OLA_HASHER=blake3 timeout 300 python tools/cold_phases.py 22 2> $O/cold_b3_preheat.txt >/dev/null; grep "\[cold\]" $O/cold_b3_preheat.txt | head -8 ;;
    cold_b3)     # one figure per call: only the first process on a fresh box sees a cold runtime (and, with luck, clean VRAM)
                 OLA_HASHER=blake3 timeout 300 python tools/cold_phases.py 22 2> $O/cold_b3.txt >/dev/null; grep -E "\[cold\]|trace upload:" $O/cold_b3.txt | head -12 ;;
    lean24)      OLA_TIMING=1 timeout 600 python tools/bench_prove.py 24 2 2> $O/phases_2p24_lean.txt | tail -3 ;;
    syncs)       timeout 1200 python -m pytest tests/test_gpu_stark.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_blake3.py tests/test_gpu_host_api.py -x -q 2>&1 | tail -3
                 for i in 1 2; do OLA_HASHER=blake3 timeout 300 python tools/bench_prove.py 22 4 2>/dev/null | tail -2; timeout 300 python tools/bench_prove.py 10 6 2>/dev/null | tail -2; done ;;
    zcols)       timeout 900 python -m pytest tests/test_gpu_stark.py tests/test_gpu_multi.py -x -q 2>&1 | tail -3
                 OLA_HASHER=blake3 OLA_TIMING=1 timeout 300 python tools/bench_prove.py 22 2 2> $O/phases_b3.txt | tail -1; grep -E "permutation Z|prove_with_traces total" $O/phases_b3.txt | tail -14 | head -4 ;;
    quot_ab)     # quotient-kernel variants (ab_tmp/libola_<v>.so): kernel-family block of the 2^22-row Blake3 proof
                 cp olavm_amd/lib/libola_gpu.so ab_tmp/libola_cur.so
                 for v in ${AB_VARIANTS:-seg96 seg48 seg160 seg256 seg96}; do cp ab_tmp/libola_$v.so olavm_amd/lib/libola_gpu.so; echo "-- $v"
                   timeout 300 python - <<'PY'
import sys, json
sys.path.insert(0, ".")
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend
blob = T.ola_stark().blob()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=22, log_n_mem=22)
import os
be = Backend(device=0, hasher=os.environ.get("AB_HASHER", "blake3"))
be.proof_stats(enable=True)
best = None
for _ in range(4):
    be.prove_with_traces(blob, traces, params, compress)
    st, ph = be.proof_stats(), be.phase_stats()
    if best is None or st["wall_ms"] < best[0]:
        best = (st["wall_ms"], ph["quotient"][0], ph["open_eval"][0], ph["merkle_levels"][0], ph["leaf_hash"][0])
print("  wall %.1f ms, quotient kernels %.2f ms, opening evaluations %.2f ms, tree levels %.2f ms, leaves %.2f ms" % best)
PY
                 done 2>&1 | grep -v amdgpu | tee $O/ab.txt; cp ab_tmp/libola_cur.so olavm_amd/lib/libola_gpu.so ;;
    peer_gather) hipcc --offload-arch=gfx950 -O2 -std=c++17 -pthread -o /tmp/gpu_peer_gather_check tests/gpu_peer_gather_check.cpp 2>/dev/null
                 for w in 2 4 8; do timeout 300 /tmp/gpu_peer_gather_check $w; done 2>&1 | grep -v amdgpu | tee $O/peer_gather.txt ;;
    prof_b3)     cd /tmp; OLA_HASHER=blake3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/tools/bench_prove.py 22 5 > $R/$O/run.log 2> $R/$O/err.txt; cd $R
                 f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -16 $O/kernel_stats.csv | cut -c1-150; tail -3 $O/run.log ;;
    matrix)      timeout 1200 python tools/bench_ntt_matrix.py --out $O/ntt_matrix.json 2>&1 | tail -40 ;;
    prof_bench)  cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --steps 20 --warmup 5 > $R/$O/bench.json 2> $R/$O/err.txt; cd $R
                 f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -25 $O/kernel_stats.csv | cut -c1-160 ;;
    *)           echo "unknown step $step" ;;
  esac
done
