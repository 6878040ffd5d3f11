"""Randomised end-to-end check: 12-table instances with random table heights; the proof from the specialised kernels must
equal the proof from the interpreter kernel byte for byte and be accepted by the oracle verifier.
usage: python tools/fuzz_prove.py [iterations] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend
from tests import oracle_lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
o = oracle_lib.load()
s = T.ola_stark(range_bits=4, limb_bits=2)
blob = s.blob()
be = Backend(device=0)
for it in range(iters):
    ln, lc, lm, lp = (int(x) for x in rng.integers(3, 14, size=4))
    traces, params, compress = tracegen.empty_program_instance(log_n=ln, log_n_cpu=lc, log_n_mem=lm, log_n_poseidon=lp, live=rng)
    os.environ.pop("OLA_AIR_KERNELS", None)
    fast = be.prove_with_traces(blob, traces, params, compress)
    os.environ["OLA_AIR_KERNELS"] = "interpreter"
    slow = be.prove_with_traces(blob, traces, params, compress)
    os.environ.pop("OLA_AIR_KERNELS", None)
    assert fast == slow, ("kernels disagree", ln, lc, lm, lp)
    rc, why = o.verify_all_proof(blob, fast, params)
    assert rc == 0, (why, ln, lc, lm, lp)
    print("ok", it, "heights", ln, lc, lm, lp, "bytes", len(fast), flush=True)
be.close()
print("prove fuzz passed")
