"""Randomised cross-check of the device library against the oracle over shapes the fixed test matrix does not list:
random (log_n, ncols) commitments (cap, coefficients, sampled leaves + paths), NTT ops, and sharded commitments.
usage: python tools/fuzz_gpu.py [iterations] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from olavm_amd.backend import Backend, OLA_NTT_EVALUATE, OLA_NTT_INTERPOLATE, OLA_NTT_COSET_LDE
from tests import oracle_lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o = oracle_lib.load()
be = Backend(device=0)
for it in range(iters):
    log_n = int(rng.integers(1, 17))
    ncols = int(rng.integers(1, 12))
    vals = oracle_lib.rand_field(rng, (ncols, 1 << log_n))
    b = be.commit(vals)
    ob = o.batch(vals)
    assert np.array_equal(b.cap(), ob.cap()), ("cap", log_n, ncols)
    assert np.array_equal(b.coeffs(), ob.coeffs()), ("coeffs", log_n, ncols)
    N = 8 << log_n
    leaves = ob.leaves()
    for j in [0, N - 1] + [int(x) for x in rng.integers(0, N, size=3)]:
        row, sib = b.leaf(j)
        assert np.array_equal(row, leaves[j]) and np.array_equal(sib, ob.prove(j)), ("leaf", log_n, ncols, j)
    world = int(rng.choice([2, 4, 8]))
    caps = []
    for r in range(world):
        sh = be.commit_shard(vals, r, world)
        caps.append(sh.cap())
        sh.free()
    assert np.array_equal(np.concatenate(caps), b.cap()), ("shard", log_n, ncols, world)
    b.free()
    # transforms
    ev = be.ntt(OLA_NTT_EVALUATE, vals)
    back = be.ntt(OLA_NTT_INTERPOLATE, ev)
    assert np.array_equal(back, vals), ("roundtrip", log_n, ncols)
    assert np.array_equal(ev[0], o.evaluate_poly(vals[0])), ("evaluate", log_n)
    print("ok", it, "log_n", log_n, "ncols", ncols, "world", world, flush=True)
be.close()
print("fuzz passed")
