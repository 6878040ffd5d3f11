#!/usr/bin/env python3
"""Where the launch-bound small tables' time goes: the README-shape proof (Fibonacci loop, 2^20-row CPU table, everything else
small) with the library's scope clock on -- per table the device time of its prove_single_table scope and of the scopes inside it,
the sum over the tables below the partition threshold (bench.py's `tables_below_the_partition_threshold`), and the whole proof.
    python tools/small_tables.py [blake3|poseidon] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from olavm_amd.air import fastexec, miniexec, ola_tables as T  # noqa: E402
from olavm_amd.backend import Backend  # noqa: E402

hasher = sys.argv[1] if len(sys.argv) > 1 else "blake3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
blob = T.ola_stark().blob()
traces, params, compress = fastexec.instance(miniexec.fibonacci_loop(47, 3000), range_bits=16, limb_bits=8, max_steps=1 << 21)
heights = [int(t.shape[1]).bit_length() - 1 for t in traces]
be = Backend(device=0, hasher=hasher)
be.prove_with_traces(blob, traces, params, compress)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    be.prove_with_traces(blob, traces, params, compress)
    ts.append(time.perf_counter() - t0)
be.scope_times(enable=True)
be.prove_with_traces(blob, traces, params, compress)
scopes = be.scope_times()
be.scope_times(enable=False)
print(f"{hasher}: heights 2^{heights}; proof {sorted(ts)[len(ts) // 2] * 1e3:.2f} ms (median of {reps}), min {min(ts) * 1e3:.2f}")
small = 0.0
inner = {}
for s in scopes:
    if s["name"].endswith("prove_single_table"):
        tag = "small" if heights[s["table"]] < 12 else "LARGE"
        print(f"  table {s['table']:2d} 2^{heights[s['table']]:<2d} {tag}  {s['ms']:8.3f} ms")
        if heights[s["table"]] < 12:
            small += s["ms"]
    elif s["depth"] == 2 and s["table"] >= 0 and heights[s["table"]] < 12:
        inner[s["name"]] = inner.get(s["name"], 0.0) + s["ms"]
print(f"  tables below the partition threshold: {small:.3f} ms")
for k, v in inner.items():
    print(f"    {k:40s} {v:8.3f} ms")
