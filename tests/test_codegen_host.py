"""The generated constraint-quotient kernels (olavm_amd/air/codegen.py), executed on the HOST.

A generated kernel is straight-line code written in a dozen macros (olavm_amd/csrc/airq.cuh).  Here the same text is compiled
with g++ against host definitions of those macros -- table cells from plain arrays, the LDS cell cache as a local array, the
segment barrier as nothing -- and one point is evaluated on random inputs.  The value must equal an independent evaluation, in
Python integers, of what the AIR data say: the table's constraint program, its permutation checks (permutation.rs:302-360) and
its cross-table-lookup checks (cross_table_lookup.rs:380-421), combined with the descriptor's weights the way the kernel's
epilogue does.  This covers what the generator adds on top of the AIR -- emit numbering, lazy 160-bit accumulation, the cells
it parks in registers / LDS slots and the hand-over of the slots between phases -- without a GPU; the GPU suite then checks the
same kernels inside whole proofs."""
import os
import subprocess

import numpy as np
import pytest

from olavm_amd.air import codegen
from olavm_amd.air import ola_tables as T
from olavm_amd.air.dsl import (KIND_ALL, KIND_FIRST, KIND_LAST, KIND_TRANSITION, OP_ADD, OP_CONST, OP_ISZERO, OP_LOCAL, OP_MUL, OP_NEXT,
                               OP_PARAM, OP_SUB, P)

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "olavm_amd", "csrc")
NCH = 2

HARNESS = r"""
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl.cuh"
using namespace ola;
struct Acc160 { u64 lo, hi; u32 top; };
static inline void acc_mad(Acc160& a, u64 x, u64 w) {
    u64 plo, phi;
    mul_wide(x, w, plo, phi);
    a.lo += plo;
    phi += (a.lo < plo) ? 1ull : 0ull;
    a.hi += phi;
    a.top += (a.hi < phi) ? 1u : 0u;
}
static inline u64 acc_reduce(const Acc160& a) { return gl_sub(gl_reduce128_cc(a.lo, a.hi), (u64)a.top << 32); }
struct QuotParams { const u64 *Tl, *Tn, *Zl, *Zn, *D; u64 lag_first, lag_last, z_last; u64* out; };
#define AIRQ_PROLOGUE(K_) constexpr int AIRQ_K = (K_); const u64* D = P.D; const u64 lag_first = P.lag_first, lag_last = P.lag_last, z_last = P.z_last; \
    (void)lag_first; (void)lag_last; Acc160 accA0 = {0, 0, 0}, accA1 = {0, 0, 0}, accT0 = {0, 0, 0}, accT1 = {0, 0, 0};
#define LC(c) P.Tl[c]
#define NC(c) P.Tn[c]
#define ZL(c) P.Zl[c]
#define ZN(c) P.Zn[c]
#define AIRQ_CACHE_DECL(S_) u64 cache_[S_]; for (int i_ = 0; i_ < (S_); i_++) cache_[i_] = 0xDEADBEEFDEADBEEFull
#define AIRQ_CACHE_PUT(s, v) (cache_[s] = (v))
#define CL(s) cache_[s]
#define AIRQ_SEGMENT_BARRIER
#define AIRQ_SEGMENT_BARRIER_C
// the accumulator interface of airq.cuh, on the u64 forms of the multipliers (the limb forms are the device's: tests/test_gpu_stark.py)
#define AIRQ_LIMBS_AT(o) constexpr int AIRQ_L0 = (o); (void)AIRQ_L0
#define AIRQ_ACC Acc160
#define AIRQ_ACC_INIT(v) {(v), 0, 0}
#define AIRQ_W(s)
#define AIRQ_MAD(a, x, wi, s) acc_mad(a, x, D[wi])
#define AIRQ_ACC_REDUCE(a) acc_reduce(a)
#define AIRQ_REFOLD_ALL
#define AIRQ_REFOLD_TRANS
#define AIRQ_EMIT_ALL(i, s0, s1, v) { const u64 v_ = (v); acc_mad(accA0, v_, D[8 + (i)]); acc_mad(accA1, v_, D[8 + AIRQ_K + (i)]); }
#define AIRQ_EMIT_TRANS(i, s0, s1, v) { const u64 v_ = (v); acc_mad(accT0, v_, D[8 + (i)]); acc_mad(accT1, v_, D[8 + AIRQ_K + (i)]); }
#define AIRQ_EPILOGUE { const u64 zh_inv = D[0]; \
    P.out[0] = gl_mul(gl_add(acc_reduce(accA0), gl_mul(z_last, acc_reduce(accT0))), zh_inv); \
    P.out[1] = gl_mul(gl_add(acc_reduce(accA1), gl_mul(z_last, acc_reduce(accT1))), zh_inv); }
#define AIRQ_THREADS 256
#define __global__
#define __launch_bounds__(...)
%(kernels)s
typedef void (*kern_t)(QuotParams);
static kern_t KERNELS[] = {%(names)s};
// stdin: per case "k ncols nz nd" then the words Tl, Tn, Zl, Zn, D, lag_first, lag_last, z_last; stdout: two words per case
int main() {
    unsigned k, ncols, nz, nd;
    while (scanf("%%u %%u %%u %%u", &k, &ncols, &nz, &nd) == 4) {
        std::vector<u64> w(2 * (size_t)ncols + 2 * (size_t)nz + nd + 3);
        for (u64& x : w) if (scanf("%%llu", &x) != 1) return 2;
        u64 out[2] = {0, 0};
        const u64* p = w.data();
        QuotParams q;
        q.Tl = p; q.Tn = p + ncols; q.Zl = p + 2 * ncols; q.Zn = q.Zl + nz; q.D = q.Zn + nz;
        q.lag_first = q.D[nd]; q.lag_last = q.D[nd + 1]; q.z_last = q.D[nd + 2]; q.out = out;
        KERNELS[k](q);
        printf("%%llu %%llu\n", out[0], out[1]);
    }
    return 0;
}
"""


def col_value(col, row):
    return (sum(int(row[c]) * int(f) for c, f in col.terms) + int(col.constant)) % P


def reference_point(airset, t, Tl, Tn, Zl, Zn, D, lag_first, lag_last, z_last):
    """What the kernel of table t must return for one point, from the AIR data alone (Python integers mod p)."""
    tab = airset.tables[t]
    jobs = airset.ctl_jobs(t, NCH)
    nperm = tab.num_permutation_batches(NCH)
    bs = tab.quotient_degree_factor
    K = codegen.num_emits(airset, t, NCH)
    d_params = 8 + 2 * K
    d_perm = d_params + tab.n_params
    d_ctl = d_perm + 2 * nperm * bs
    val = {}
    emits = []                                            # (kind, value) in emit order
    for it in tab.schedule():
        if it[0] == "emit":
            emits.append((it[1], val[it[2]]))
            continue
        j = it[1]
        op, a, b = tab.nodes[j]
        if op == OP_LOCAL:
            v = int(Tl[a])
        elif op == OP_NEXT:
            v = int(Tn[a])
        elif op == OP_CONST:
            v = int(a) % P
        elif op == OP_PARAM:
            v = int(D[d_params + a])
        elif op == OP_ADD:
            v = (val[a] + val[b]) % P
        elif op == OP_SUB:
            v = (val[a] - val[b]) % P
        elif op == OP_MUL:
            v = val[a] * val[b] % P
        elif op == OP_ISZERO:
            v = 1 if val[a] == 0 else 0
        else:
            raise ValueError(op)
        val[j] = v
    # permutation checks: Z(1) = 1; Z(gx) * prod(rhs) = Z(x) * prod(lhs) per batch of `bs` (pair, challenge) instances
    for b in range(nperm):
        emits.append((KIND_FIRST, (int(Zl[b]) - 1) % P))
    total = len(tab.permutation_pairs) * NCH
    inst = 0
    for b in range(nperm):
        pl = pr = 1
        for i in range(bs):
            if inst >= total:
                break
            pair = tab.permutation_pairs[inst // NCH]
            beta, gamma = int(D[d_perm + 2 * (b * bs + i)]), int(D[d_perm + 2 * (b * bs + i) + 1])
            lhs = sum(int(Tl[lc]) * pow(beta, k, P) for k, (lc, _) in enumerate(pair)) % P
            rhs = sum(int(Tl[rc]) * pow(beta, k, P) for k, (_, rc) in enumerate(pair)) % P
            pl = pl * (lhs + gamma) % P
            pr = pr * (rhs + gamma) % P
            inst += 1
        emits.append((KIND_ALL, (int(Zn[b]) * pr - int(Zl[b]) * pl) % P))
    # cross-table lookups: Z(1) = select(f, combo)(1); Z(gx) = Z(x) * select(f, combo)(gx)
    off = d_ctl
    for i, twc in enumerate(jobs):
        gamma = int(D[off])
        weights = [1] + [int(D[off + 1 + k]) for k in range(1, len(twc.columns))]      # beta^0: the kernel does not read that slot
        off += 1 + len(twc.columns)
        sel = []
        for row in (Tl, Tn):
            combo = (sum(col_value(c, row) * w for c, w in zip(twc.columns, weights)) + gamma) % P
            if twc.filter_column is not None:
                f = col_value(twc.filter_column, row)
                combo = (f * combo + 1 - f) % P
            sel.append(combo)
        zl, zn = int(Zl[nperm + i]), int(Zn[nperm + i])
        emits.append((KIND_FIRST, (zl - sel[0]) % P))
        emits.append((KIND_TRANSITION, (zn - zl * sel[1]) % P))
    assert len(emits) == K
    out = []
    for c in range(NCH):
        a = tr = 0
        for i, (kind, v) in enumerate(emits):
            w = int(D[8 + c * K + i])
            if kind == KIND_TRANSITION:
                tr += v * w
            elif kind == KIND_FIRST:
                a += v * lag_first % P * w
            elif kind == KIND_LAST:
                a += v * lag_last % P * w
            else:
                assert kind == KIND_ALL
                a += v * w
        out.append((a + z_last * (tr % P)) % P * int(D[0]) % P)
    return out


@pytest.mark.parametrize("variant", ["full", "miniature"])
def test_generated_kernels_on_the_host_equal_the_air_data(tmp_path, variant):
    airset = T.ola_stark() if variant == "full" else T.ola_stark(range_bits=4, limb_bits=2)
    nt = len(airset.tables)
    bodies, names, used_cache = [], [], 0
    for t in range(nt):
        name = "airq_host_%d" % t
        src, K = codegen.table_kernel(airset, t, name, NCH)
        assert K == codegen.num_emits(airset, t, NCH)
        used_cache += "AIRQ_CACHE_PUT" in src
        bodies.append(src)
        names.append(name)
    assert used_cache >= 2, "no table parks cells in LDS any more: the cache path is not being tested"
    cpp = tmp_path / "airq_host.cpp"
    cpp.write_text(HARNESS % {"kernels": "\n".join(bodies), "names": ", ".join(names)})
    exe = str(tmp_path / "airq_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wno-unknown-pragmas", "-I" + CSRC, str(cpp), "-o", exe])
    rng = np.random.default_rng(20240928)
    cases, lines = [], []
    for t in range(nt):
        tab = airset.tables[t]
        nz = tab.num_permutation_batches(NCH) + len(airset.ctl_jobs(t, NCH))
        K = codegen.num_emits(airset, t, NCH)
        nd = 8 + 2 * K + tab.n_params + 2 * tab.num_permutation_batches(NCH) * tab.quotient_degree_factor + \
            sum(1 + len(j.columns) for j in airset.ctl_jobs(t, NCH))
        for rep in range(3):
            w = rng.integers(0, P, size=2 * tab.ncols + 2 * nz + nd + 3, dtype=np.uint64)
            if rep == 2:                                   # small values: is_zero sees zeros, filters see 0 / 1
                w[:2 * tab.ncols] = rng.integers(0, 2, size=2 * tab.ncols, dtype=np.uint64)
            cases.append((t, tab.ncols, nz, nd, w))
            lines.append("%d %d %d %d\n%s" % (t, tab.ncols, nz, nd, " ".join(str(int(x)) for x in w)))
    r = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = [tuple(int(x) for x in ln.split()) for ln in r.stdout.strip().split("\n")]
    assert len(got) == len(cases)
    for (t, ncols, nz, nd, w), g in zip(cases, got):
        Tl, Tn = w[:ncols], w[ncols:2 * ncols]
        Zl, Zn = w[2 * ncols:2 * ncols + nz], w[2 * ncols + nz:2 * ncols + 2 * nz]
        D = w[2 * ncols + 2 * nz:2 * ncols + 2 * nz + nd]
        lag_first, lag_last, z_last = (int(x) for x in w[2 * ncols + 2 * nz + nd:])
        want = reference_point(airset, t, Tl, Tn, Zl, Zn, D, lag_first, lag_last, z_last)
        assert list(g) == want, "table %d (%s)" % (t, airset.tables[t].name)


def test_slot_plan_check_rejects_a_print_that_segments_differently():
    """the LDS slot plan comes from a counting print; the final print re-records the cells of every segment and
    check_plan compares (and replays the slots).  Here: the real plan of the CPU table passes, doctored ones do not."""
    segs = [{("L", 1), ("L", 2)}, {("L", 1), ("N", 3)}, {("L", 2), ("L", 1)}]
    hits, puts, _ = codegen.plan_lds(segs, 2)
    plan = (hits, puts, segs)
    codegen.check_plan(plan, [set(s) for s in segs])
    with pytest.raises(AssertionError):                 # one segment more
        codegen.check_plan(plan, [set(s) for s in segs] + [set()])
    with pytest.raises(AssertionError):                 # same cut, another cell
        codegen.check_plan(plan, [set(segs[0]), {("L", 1), ("N", 4)}, set(segs[2])])
    bad_hits = [dict(h) for h in hits]
    for i, h in enumerate(bad_hits):                    # a cell served from a slot nobody parked it in
        if h:
            c = next(iter(h))
            h[c] = 1 - h[c]
            break
    with pytest.raises(AssertionError):
        codegen.check_plan((bad_hits, puts, segs), [set(s) for s in segs])
