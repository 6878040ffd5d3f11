"""Cold start of the prover, phase by phase: a fresh process and context prove the 2^22-row instance twice with OLA_TIMING=1;
the first proof's phase lines against the second's show where the cold seconds go (stderr carries the [ola-timing] lines)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OLA_TIMING"] = "1"
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
from olavm_amd.air import ola_tables as T, tracegen  # noqa: E402
from olavm_amd.backend import Backend  # noqa: E402

blob = T.ola_stark().blob()
# the order of `ola prove` (client/src/main.rs:174-214): context first, then the host produces the traces, then the proof
t0 = time.perf_counter()
be = Backend(device=0, hasher=os.environ.get("OLA_HASHER", "poseidon"))
print("[cold] ola_gpu_init %.3f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
if os.environ.get("OLA_COLD_RESERVE"):
    be.reserve(blob, [log_n, log_n, 18, 1, 16, 10, 10, 10, 10, 10, 10, 10])
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=log_n, log_n_mem=log_n)
if os.environ.get("OLA_COLD_PREHEAT"):
    # is the first proof's excess the device coming out of idle?  keep it busy for a few hundred ms right before proof 0
    t0, bad, calls = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < 0.1 * int(os.environ["OLA_COLD_PREHEAT"]):
        bad += be.selftest(1 << 30)
        calls += 1
    print("[cold] preheat: %d field self-tests, %d mismatches, %.3f s" % (calls, bad, time.perf_counter() - t0), file=sys.stderr, flush=True)
for i in range(3):
    print("[cold] ---- proof %d ----" % i, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    p = be.prove_with_traces(blob, traces, params, compress)
    print("[cold] proof %d: %.3f s, %d bytes" % (i, time.perf_counter() - t0, len(p)), file=sys.stderr, flush=True)

# fresh host arrays, warm context: is the first proof's excess a property of the host pages (pinning / first DMA from them) or of the device side?
import numpy as np  # noqa: E402
fresh = [np.array(t, copy=True) for t in traces]
print("[cold] ---- proof on fresh host arrays (warm context) ----", file=sys.stderr, flush=True)
t0 = time.perf_counter()
p = be.prove_with_traces(blob, fresh, params, compress)
print("[cold] fresh-host-arrays proof: %.3f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
t0 = time.perf_counter()
p = be.prove_with_traces(blob, fresh, params, compress)
print("[cold] same arrays again: %.3f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
