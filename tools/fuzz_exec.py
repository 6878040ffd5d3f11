"""Randomised end-to-end check on EXECUTED programs: random straight-line programs (tests/test_tracegen_native.random_program)
run through the native trace generator; the GPU proof from the specialised kernels must equal the interpreter kernel's byte
for byte and be accepted by the oracle verifier.   usage: python tools/fuzz_exec.py [iterations] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from olavm_amd.air import fastexec, ola_tables as T
from olavm_amd.backend import Backend
from tests import oracle_lib
from tests.test_tracegen_native import random_program

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
o = oracle_lib.load()
s = T.ola_stark()                 # full-size fixed tables: clock differences of long programs exceed the miniature range table
blob = s.blob()
be = Backend(device=0)
for it in range(iters):
    prog = random_program(rng, length=int(rng.integers(10, 400)))
    traces, params, compress = fastexec.instance(prog, range_bits=16, limb_bits=8)
    os.environ.pop("OLA_AIR_KERNELS", None)
    fast = be.prove_with_traces(blob, traces, params, compress)
    os.environ["OLA_AIR_KERNELS"] = "interpreter"
    slow = be.prove_with_traces(blob, traces, params, compress)
    os.environ.pop("OLA_AIR_KERNELS", None)
    assert fast == slow, ("kernels disagree", it)
    rc, why = o.verify_all_proof(blob, fast, params)
    assert rc == 0, (it, why)
    print("ok", it, "instructions", len(prog.ins), "cpu rows 2^%d" % (int(traces[0].shape[1]).bit_length() - 1), "bytes", len(fast), flush=True)
print("exec fuzz passed")
