#!/usr/bin/env python3
"""Time PolynomialBatch::from_values on the device (iNTT + coset LDE x8 + Poseidon Merkle) for a table shape.
usage: python tools/bench_commit.py [log_n] [ncols] [reps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from olavm_amd.backend import Backend

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 94
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
be = Backend(device=0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.randint(0, 2**63 - 1, (ncols, 1 << log_n), dtype=torch.int64, device="cuda", generator=g)
for i in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = be.commit_dev(vals.data_ptr(), ncols, log_n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b.free()
    if i:
        leaves = 8 << log_n
        perms = leaves * ((ncols + 7) // 8 + 1)
        print(f"commit {ncols} x 2^{log_n}: {dt*1e3:.1f} ms  ({perms/dt/1e9:.3f} G Poseidon perm/s equiv)")
be.close()
