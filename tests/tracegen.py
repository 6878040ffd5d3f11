"""Test-side generator of VALID traces for the tables implemented so far (SURVEY f-1), restating the reference's
generation/padding rules:
  circuits/src/generation/builtin.rs:208-247   generate_cmp_trace   (pad rows: op0 = gte = abs_diff = abs_diff_inv = 1)
  circuits/src/generation/builtin.rs:249-316   generate_rc_trace    (fixed u16 table column, permuted lookup columns)
  circuits/src/stark/lookup.rs:68-132          permuted_cols        (Halo2-style permuted input / table columns)
Test infrastructure only."""
import numpy as np

from tests.oracle_lib import P
from olavm_amd.air import ola_tables as T


def next_pow2(n):
    return 2 if n < 2 else 1 << (n - 1).bit_length()


def permuted_cols(inputs, table):
    n = len(inputs)
    si, st = sorted(int(x) % P for x in inputs), sorted(int(x) % P for x in table)
    unused_inds, unused_vals, permuted_table = [], [], [0] * n
    i = j = 0
    while j < n and i < n:
        a, b = si[i], st[j]
        if a > b:
            unused_vals.append(b)
            j += 1
        elif a < b:
            if unused_vals:
                permuted_table[i] = unused_vals.pop()
            else:
                unused_inds.append(i)
            i += 1
        else:
            permuted_table[i] = b
            i += 1
            j += 1
    unused_vals += st[j:]
    unused_inds += list(range(i, n))
    assert len(unused_inds) == len(unused_vals)
    for ind, val in zip(unused_inds, unused_vals):
        permuted_table[ind] = val
    return si, permuted_table


def cmp_rows(rng, count, max_val=1 << 32):
    """(op0, op1) pairs -> CmpRow fields (core/src/trace/trace.rs CmpRow)."""
    rows = []
    for _ in range(count):
        a, b = int(rng.integers(0, max_val)), int(rng.integers(0, max_val))
        if rng.integers(0, 8) == 0:
            b = a
        gte = 1 if a >= b else 0
        d = abs(a - b)
        rows.append((a, b, gte, d, pow(d, P - 2, P) if d else 0, 1))
    return rows


def generate_cmp_trace(rows):
    n = next_pow2(len(rows))
    t = np.zeros((T.COL_NUM_CMP, n), dtype=np.uint64)
    for i, r in enumerate(rows):
        for c in range(6):
            t[c, i] = r[c]
    for i in range(len(rows), n):
        t[T.COL_CMP_OP0, i] = t[T.COL_CMP_GTE, i] = t[T.COL_CMP_ABS_DIFF, i] = t[T.COL_CMP_ABS_DIFF_INV, i] = 1
    return t


def generate_rc_trace(vals_with_filters, range_bits=16):
    """vals_with_filters: list of (val, cpu_f, mem_sort_f, mem_region_f, cmp_f).  range_bits = 16 in the reference
    (RANGE_CHECK_U16_SIZE); smaller values give a structurally identical miniature table for CPU-sized tests."""
    size = 1 << range_bits
    n = next_pow2(max(len(vals_with_filters), size))
    t = np.zeros((T.COL_NUM_RC, n), dtype=np.uint64)
    for i, (v, f0, f1, f2, f3) in enumerate(vals_with_filters):
        assert v < size * size
        t[T.RC_CPU_FILTER, i], t[T.RC_MEMORY_SORT_FILTER, i], t[T.RC_MEMORY_REGION_FILTER, i], t[T.RC_CMP_FILTER, i] = f0, f1, f2, f3
        t[T.RC_VAL, i], t[T.RC_LIMB_LO, i], t[T.RC_LIMB_HI, i] = v, v % size, v // size
    fix = list(range(size)) + [size - 1] * (n - size)
    t[T.RC_FIX_RANGE_CHECK_U16] = fix
    t[T.RC_LIMB_LO_PERMUTED], t[T.RC_FIX_RANGE_CHECK_U16_PERMUTED_LO] = permuted_cols(t[T.RC_LIMB_LO], fix)
    t[T.RC_LIMB_HI_PERMUTED], t[T.RC_FIX_RANGE_CHECK_U16_PERMUTED_HI] = permuted_cols(t[T.RC_LIMB_HI], fix)
    return t


def cmp_rangecheck_instance(rng, n_cmp, range_bits=16):
    """Two-table instance (Cmp, RangeCheck) consistent with the cmp<->rangecheck cross-table lookup
    (stark/ola_stark.rs:282-296): every real cmp row's abs_diff appears as a rangecheck VAL with CMP_FILTER = 1."""
    rows = cmp_rows(rng, n_cmp, max_val=1 << (2 * range_bits))
    cmp_t = generate_cmp_trace(rows)
    rc_t = generate_rc_trace([(r[3], 0, 0, 0, 1) for r in rows], range_bits)
    return cmp_t, rc_t
