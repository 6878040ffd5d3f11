"""Is the first-call cost of ola_prove_with_traces a property of the context (one-time) or of the host buffers (every new
trace)?  Proves the same instance from the original arrays twice, then from fresh copies of them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from olavm_amd.air import fastexec, miniexec as M, ola_tables as T
from olavm_amd.backend import Backend
count = int(sys.argv[1]) if len(sys.argv) > 1 else 290000
blob = T.ola_stark().blob()
traces, params, compress = fastexec.instance(M.memory_program(count), range_bits=16, limb_bits=8, max_steps=1 << 25)
be = Backend(device=0)
def run(tag, tr):
    t0 = time.perf_counter(); n = len(be.prove_with_traces(blob, tr, params, compress)); print("%-28s %.3f s" % (tag, time.perf_counter() - t0), flush=True)
run("first call", traces); run("same arrays again", traces)
fresh = [t.copy() for t in traces]
run("fresh copies of the arrays", fresh); run("those again", fresh)
