"""ola_prove_with_traces on a REAL execution with a 2^20-row CPU table.

The traces come from the executor (native generator include/ola_tracegen.h, or olavm_amd/air/miniexec.py with --python)
running the memory program -- a store loop and a
load / add / store / load loop over `count` cells -- against the full-size fixed tables (range_bits 16, limb_bits 8):
count = 70000 gives 980 k executed CPU rows (2^20), 280 k memory cells, a 2^21-row program table (every fetched instruction and
immediate word) and 280 k range-checked sort values.  The proof is checked with the oracle's verifier; with OLA_TIMING=1 the
library prints its per-phase times (named after the reference's `timed!` scopes) to stderr.

    python tools/bench_prove_real.py [count] [reps] [--json out.json] [--phases] [--oracle] [--python] [--storage-slots N] [--hasher blake3]

--hasher blake3 proves under the reference's Blake3GoldilocksConfig (the configuration of its README numbers): BLAKE3 Merkle
trees and challenger, Poseidon proof of work; the oracle checks under the same configuration.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    count = int(args[0]) if args else 70000
    reps = int(args[1]) if len(args) > 1 else 3
    out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    hasher = sys.argv[sys.argv.index("--hasher") + 1] if "--hasher" in sys.argv else "poseidon"
    from olavm_amd.air import fastexec, miniexec as M, ola_tables as T
    from olavm_amd.backend import Backend
    s = T.ola_stark()
    blob = s.blob()
    t0 = time.time()
    gen = M if "--python" in sys.argv else fastexec          # the native generator reproduces the Python executor word for word
    slots = int(sys.argv[sys.argv.index("--storage-slots") + 1]) if "--storage-slots" in sys.argv else 0
    if slots:       # BASELINE config 4: a Poseidon table of 1026 * slots live rows next to the CPU / memory tables
        prog, kw = M.storage_heavy_program(slots, count), {"prove_program_hash": True}
    else:
        prog, kw = M.memory_program(count), {}
    traces, params, compress = gen.instance(prog, range_bits=16, limb_bits=8, max_steps=1 << 24, **kw)
    gen_s = time.time() - t0
    heights = [int(t.shape[1]).bit_length() - 1 for t in traces]
    print("executed + filled 12 tables in %.1f s; log2 heights %s" % (gen_s, heights), flush=True)
    be = Backend(device=0, hasher=hasher)
    times = []
    for _ in range(reps):
        t0 = time.time()
        proof = be.prove_with_traces(blob, traces, params, compress)
        times.append(time.time() - t0)
        print("prove_with_traces: %.3f s, proof %d bytes" % (times[-1], len(proof)), flush=True)
    if "--phases" in sys.argv:          # one more proof on a context created with OLA_TIMING=1 (phase lines go to stderr)
        os.environ["OLA_TIMING"] = "1"
        be2 = Backend(device=0, hasher=hasher)
        be2.prove_with_traces(blob, traces, params, compress)          # cold context: tables, twiddles, allocations
        print("[ola-timing] ---- warm proof ----", file=sys.stderr, flush=True)
        be2.prove_with_traces(blob, traces, params, compress)
        be2.close()
    from tests import oracle_lib
    o = oracle_lib.load()
    import contextlib
    cfg = o.hasher(hasher)
    with cfg:
        rc, why = o.verify_all_proof(blob, proof, params)
    print("oracle verifier (%s configuration):" % hasher, rc, why, flush=True)
    oracle_s = None
    if "--oracle" in sys.argv:          # the CPU restatement on the same traces (all host cores it uses), byte comparison included
        t0 = time.time()
        with o.hasher(hasher):
            ref = o.prove_with_traces(blob, traces, params, compress)
        oracle_s = time.time() - t0
        print("oracle prove_with_traces (CPU port, %d threads): %.1f s, bytes identical: %s" % (o.lib.oracle_num_threads(), oracle_s, ref == proof), flush=True)
        if ref != proof:
            rc = 1
    if out:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        with open(out, "w") as f:
            json.dump({"workload": ("storage_heavy_program(%d, %d)" % (slots, count) if slots else "memory_program(%d)" % count) + ", ola_stark(range_bits=16, limb_bits=8)", "log2_heights": heights, "hasher": hasher,
                       "trace_generation_s": round(gen_s, 1), "prove_s": [round(t, 4) for t in times], "proof_bytes": len(proof),
                       "oracle_verifier_rc": rc, "oracle_cpu_port_prove_s": None if oracle_s is None else round(oracle_s, 1)}, f, indent=1)
    if rc != 0:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
