"""Times ola_prove_with_traces on the 12-table OlaStark with an empty-program (all padding) execution of a given CPU-table
height; the prover's work does not depend on the cell values.  Usage: python tools/bench_prove.py [log_n_cpu] [reps] [log_n_poseidon]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from olavm_amd.air import ola_tables as T
from olavm_amd.backend import Backend
from olavm_amd.air import tracegen

L = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
LP = int(sys.argv[3]) if len(sys.argv) > 3 else None      # height of the Poseidon table (BASELINE config 4: 22)
s = T.ola_stark()
blob = s.blob()
t0 = time.time()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=L, log_n_mem=L, log_n_poseidon=LP)
print("tracegen %.1fs; heights" % (time.time() - t0), [int(t.shape[1]).bit_length() - 1 for t in traces], flush=True)
HASHER = os.environ.get("OLA_HASHER", "poseidon")      # blake3 = the reference's Blake3GoldilocksConfig
be = Backend(device=0, hasher=HASHER)
for r in range(reps):
    t0 = time.time()
    proof = be.prove_with_traces(blob, traces, params, compress)
    print("prove_with_traces: %.3f s, proof %d bytes" % (time.time() - t0, len(proof)), flush=True)
if os.environ.get("OLA_VERIFY"):
    from tests import oracle_lib
    o = oracle_lib.load()
    with o.hasher(HASHER):
        rc, why = o.verify_all_proof(blob, proof, params)
    print("oracle verifier:", rc, why)
