#!/usr/bin/env python3
"""What does the first proof of a process still pay after ola_gpu_warmup?  Fresh children, Blake3 configuration, 2^22 rows:
prime = none | commit (one 8 x 2^16 commitment before the proof) | mini (a whole 2^10-row 12-table proof before it) | upload
(one 256 MB host-to-device copy through the pinned ring's path).  Prints first / second proof seconds per variant, alternated."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd import backend as B
blob = T.ola_stark().blob()
B.load_library(); B.warmup(0, airset=blob if os.environ.get('OLA_PRIME') == '1' else None)
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=22, log_n_mem=22)
mini = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=12, log_n_mem=12)
be = B.Backend(device=0, hasher="blake3")
prime = %(prime)r
t0 = time.perf_counter()
if prime in ("commit", "both"):
    v = np.random.default_rng(1).integers(0, 2**62, (8, 1 << 16), dtype=np.uint64)
    be.commit(v).free()
if prime in ("mini", "both"):
    be.prove_with_traces(blob, mini[0], mini[1], mini[2])
if prime == "upload":
    x = torch.empty(32 << 20, dtype=torch.int64).pin_memory(); y = x.cuda(non_blocking=True); torch.cuda.synchronize()
tp = time.perf_counter() - t0
ts = []
for _ in range(3):
    t0 = time.perf_counter(); be.prove_with_traces(blob, traces, params, compress); ts.append(round(time.perf_counter() - t0, 4))
print(json.dumps({"prime": prime, "prime_s": round(tp, 4), "proofs_s": ts, "upload_wait_ms_last": round(be.upload_stats()["waited_ms"], 1)}))
"""
for rep in range(2):
    for prime in ("none", "commit", "mini", "upload"):
        out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "prime": prime}], capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(line[-1] if line else out.stderr[-400:], flush=True)
