"""Randomised cross-check of the LARGE transforms (2^14 rows and more: the T-form pass kernels, olavm_amd/csrc/ntt2t.cuh)
against the oracle, over shapes the fixed test matrix does not list: random sizes 2^14 ... 2^20, 1 ... 21 columns (not multiples
of the kernels' eight columns per workgroup), every operation of the C ABI -- evaluate, interpolate, the x2 / x4 / x8 coset LDE in
natural and (x8) leaf order, coset transforms with random shifts and blow-up 1 -- on canonical, non-canonical (>= p) and sparse
inputs.  A column at a time is compared with the oracle (all words).
usage: python tools/fuzz_ntt_large.py [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from olavm_amd.backend import (Backend, OLA_NTT_EVALUATE, OLA_NTT_INTERPOLATE, OLA_NTT_COSET_LDE, OLA_NTT_COSET_INTERPOLATE,
                               OLA_NTT_COSET_LDE_LEAF_ORDER)
from tests import oracle_lib

P = 0xFFFFFFFF00000001
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
o = oracle_lib.load()
be = Backend(device=0)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


t0 = time.time()
for it in range(iters):
    log_n = int(rng.integers(14, 21))
    ncols = int(rng.integers(1, 22))
    n = 1 << log_n
    kind = it % 3
    vals = oracle_lib.rand_field(rng, (ncols, n))
    if kind == 1:        # any u64, including words >= p
        vals = rng.integers(0, 1 << 64, size=(ncols, n), dtype=np.uint64)
    elif kind == 2:      # sparse: mostly zeros and ones, a few large words
        vals = (rng.integers(0, 50, size=(ncols, n)) == 0).astype(np.uint64) * vals + (rng.integers(0, 7, size=(ncols, n)) == 0).astype(np.uint64)
    canon = np.where(vals >= np.uint64(P), vals - np.uint64(P), vals)
    cols = sorted({0, ncols - 1, int(rng.integers(0, ncols))})
    ev = be.ntt(OLA_NTT_EVALUATE, vals)
    for c in cols:
        assert np.array_equal(ev[c], o.evaluate_poly(canon[c])), ("evaluate", log_n, ncols, c)
    back = be.ntt(OLA_NTT_INTERPOLATE, ev)
    assert np.array_equal(back, canon), ("interpolate(evaluate)", log_n, ncols)
    blow = int(rng.choice([1, 2, 3]))
    shift = 7
    lde = be.ntt(OLA_NTT_COSET_LDE, vals, shift=shift, blowup_log=blow)
    for c in cols[:2]:
        assert np.array_equal(lde[c], o.evaluate_poly_with_offset(canon[c], shift, 1 << blow)), ("coset lde", log_n, ncols, blow, c)
    leaf = be.ntt(OLA_NTT_COSET_LDE_LEAF_ORDER, vals, shift=7, blowup_log=3)
    nat = o.evaluate_poly_with_offset(canon[cols[0]], 7, 8)
    idx = np.array([bitrev(j, log_n + 3) for j in rng.integers(0, 8 * n, size=4096)], dtype=np.int64)
    js = np.array([bitrev(int(i), log_n + 3) for i in idx], dtype=np.int64)
    assert np.array_equal(leaf[cols[0]][js], nat[idx]), ("leaf order", log_n, ncols)
    s = int(rng.integers(2, P, dtype=np.uint64))
    ce = be.ntt(OLA_NTT_COSET_LDE, vals[:1], shift=s, blowup_log=0)
    assert np.array_equal(ce[0], o.evaluate_poly_with_offset(canon[0], s, 1)), ("coset fft, random shift", log_n, s)
    ci = be.ntt(OLA_NTT_COSET_INTERPOLATE, ce, shift=s)
    assert np.array_equal(ci[0], canon[0]), ("coset ifft, random shift", log_n, s)
    print("ok", it, "log_n", log_n, "ncols", ncols, "inputs", ("canonical", "any u64", "sparse")[kind], "blowup", 1 << blow, "%.0fs" % (time.time() - t0), flush=True)
print("fuzz ok:", iters, "iterations")
