"""Start-up and per-pass timing entry points added in ABI revision 6 (include/ola_gpu.h): ola_gpu_warmup / ola_gpu_warmup_wait --
the reference's early hook, OlaStark::default() -> init_gpu() (circuits/src/stark/ola_stark.rs:47) -- and ola_gpu_ntt_pass_times."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import torch
from olavm_amd import backend as B
from tests import oracle_lib
B.load_library()
from olavm_amd.air import ola_tables as T
blob = T.ola_stark().blob() if %(early)d == 2 else None
t0 = time.perf_counter()
if %(early)d:
    B.warmup(0, airset=blob)
    t_call = time.perf_counter() - t0
    time.sleep(2.0)                      # the host "generates traces"
    t1 = time.perf_counter()
    be = B.Backend(device=0)
    t_init = time.perf_counter() - t1
    warm_ms = B.warmup_wait()
    B.warmup(0)                          # a second call is a no-op
else:
    t_call, warm_ms = 0.0, None
    t1 = time.perf_counter()
    be = B.Backend(device=0)
    t_init = time.perf_counter() - t1
o = oracle_lib.load()
vals = oracle_lib.rand_field(np.random.default_rng(5), (3, 1 << 10))
b = be.commit(vals)
ok = bool(np.array_equal(b.cap(), o.batch(vals).cap()))
print(json.dumps({"call_s": t_call, "init_s": t_init, "warm_ms": warm_ms, "cap_ok": ok}))
"""


def _child(early):
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "early": early}], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_warmup_returns_at_once_and_takes_the_start_up_off_the_first_context():
    early, primed, lazy = _child(1), _child(2), _child(0)
    assert early["cap_ok"] and lazy["cap_ok"] and primed["cap_ok"]  # a context created after a warm-up (or taken over from it) computes the same commitment
    assert early["call_s"] < 0.05 and primed["call_s"] < 0.05, (early, primed)   # ola_gpu_warmup does not block (the AIR set is copied)
    assert early["warm_ms"] > 50 and primed["warm_ms"] > early["warm_ms"]          # the thread did the start-up (and the priming proofs)
    assert early["init_s"] < 0.25 * lazy["init_s"] + 0.05, (early, lazy)   # ... and ola_gpu_init no longer pays for it
    assert primed["init_s"] < 0.25 * lazy["init_s"] + 0.05, (primed, lazy)


def test_a_primed_context_proves_the_same_bytes_under_both_hashers():
    """The context ola_gpu_init takes over from the warm-up ran two throw-away proofs of an all-zero instance with the
    divisibility check off: afterwards it must be indistinguishable -- same AllProof bytes as the oracle prover under the
    hasher it is given, and a trace that violates its constraints is refused again (OLA_E_QUOTIENT_DEGREE)."""
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from olavm_amd import backend as B
from olavm_amd.air import miniexec as M, ola_tables as T
from tests import oracle_lib
o = oracle_lib.load()
blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
B.warmup(0, airset=blob)
ms = B.warmup_wait()
traces, params, compress = M.instance(M.fibonacci(12))
for hasher in ("blake3", "poseidon"):
    be = B.Backend(device=0, hasher=hasher)      # the first one takes the primed context over, the second is created afresh
    got = be.prove_with_traces(blob, traces, params, compress)
    with o.hasher(hasher):
        assert got == o.prove_with_traces(blob, traces, params, compress), hasher
    bad = [np.array(t, copy=True) for t in traces]
    bad[0][3, 1] ^= 1
    try:
        be.prove_with_traces(blob, bad, params, compress)
        raise SystemExit("a broken trace was accepted")
    except B.OlaGpuError as e:
        assert e.code == -4, e
    be.close()
print("ok %%.1f" %% ms)
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


def test_warmup_wait_without_warmup_is_an_error():
    from olavm_amd import backend as B
    # (this process may have warmed up already in another test; only check the contract when it has not)
    L = B.load_library()
    import ctypes as C
    ms = C.c_double()
    rc = L.ola_gpu_warmup_wait(C.byref(ms))
    assert rc in (0, -1)


def test_pass_times_name_the_three_passes_of_a_2p16_and_2p20_transform():
    import torch
    from olavm_amd.backend import Backend, OLA_NTT_EVALUATE
    from tests import oracle_lib
    o = oracle_lib.load()
    be = Backend(device=0)
    for log_n, passes in ((16, 2), (20, 3)):
        vals = oracle_lib.rand_field(np.random.default_rng(log_n), (8, 1 << log_n))
        d = torch.from_numpy(vals.view(np.int64)).cuda()
        out, scratch = torch.empty_like(d), torch.empty_like(d)
        assert be.ntt_pass_times(enable=True) == {}
        for _ in range(3):
            be.ntt_dev(OLA_NTT_EVALUATE, d.data_ptr(), out.data_ptr(), log_n, 8, scratch_ptr=scratch.data_ptr())
        pt = be.ntt_pass_times(enable=False)
        assert len(pt) == passes and all(v["launches"] == 3 and v["total_ms"] > 0 and v["elements"] == 3 * 8 * (1 << log_n) for v in pt.values()), pt
        assert all(k.startswith("ntt2t_pass_kernel<") for k in pt)
        assert sum(k.endswith(",2>") for k in pt) == 1               # one closing pass (load multipliers in registers)
        # the events change nothing: same values as the oracle's evaluate_poly
        got = out.cpu().numpy().view(np.uint64)
        assert all(np.array_equal(got[c], o.evaluate_poly(vals[c])) for c in (0, 7))
        # switched off: nothing is recorded
        be.ntt_dev(OLA_NTT_EVALUATE, d.data_ptr(), out.data_ptr(), log_n, 8, scratch_ptr=scratch.data_ptr())
        assert be.ntt_pass_times() == {}
    be.close()
