"""Generator of VALID traces without the reference's (Rust) executor (SURVEY f-1), restating its generation / padding
rules -- used by the tests and, for the end-to-end timing, by bench.py:
  circuits/src/generation/builtin.rs:208-247   generate_cmp_trace   (pad rows: op0 = gte = abs_diff = abs_diff_inv = 1)
  circuits/src/generation/builtin.rs:249-316   generate_rc_trace    (fixed u16 table column, permuted lookup columns)
  circuits/src/stark/lookup.rs:68-132          permuted_cols        (Halo2-style permuted input / table columns)
The per-table padding generators and `empty_program_instance` (all 12 tables, optionally with live range-check / bitwise /
Poseidon rows) are further down."""
import numpy as np

from . import ola_tables as T
from .dsl import P


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


# ------------------------------------------------------------------------------------------------ "empty program" instance
# Padding rows of every table as the reference's generators emit them for an execution without rows
# (generation/{cpu,memory,builtin,poseidon,poseidon_chunk,storage,tape,sccall,prog}.rs), so that all 19 cross-table
# lookups are trivially consistent (every filter is 0).  Two places deviate from the generators because their empty-input
# output does not satisfy the AIR (memory_stark.rs:265-270 are un-gated `constraint`s, so the wrap-around from the last
# prophet row to row 0 needs row 0 outside the prophet region): memory row 0 is a stack-region row with its S_PROPHET
# selector set.
def cpu_padding_trace(n):
    t = np.zeros((T.NUM_CPU_COLS, n), dtype=np.uint64)
    t[T.COL_INST] = 1048576
    t[T.COL_OPCODE] = T.op_mask("END")
    t[T.COL_S_END] = 1
    t[T.COL_IS_ENTRY_SC] = 1
    t[T.COL_IS_NEXT_LINE_DIFF_INST] = 1
    t[T.COL_IS_PADDING] = 1
    return t


def memory_padding_trace(n, reference_quirks=False):
    """Memory table of an execution without memory cells.  reference_quirks: the rows generation/memory.rs:95-153 writes -- EVERY row a
    prophet-region row starting at address p - (2^32 - 1), row 0 without its S_PROPHET selector and address step (the table then breaks
    memory_stark.rs:265-270 on the wrap-around from the last row to row 0; for comparing whole pipelines with a build of the reference,
    integration/pin/).  Default: one stack-region row first, the reference's prophet rows one place later."""
    t = np.zeros((T.NUM_MEM_COLS, n), dtype=np.uint64)
    span = 2**32 - 1
    addr = (0 - span) % P
    if reference_quirks:
        i = np.arange(n, dtype=np.uint64)
        a = (np.uint64(addr) + i) % np.uint64(P)
        neg = (np.uint64(P) - a) % np.uint64(P)
        t[T.COL_MEM_ADDR] = a
        t[T.COL_MEM_IS_WRITE] = 1
        t[T.COL_MEM_DIFF_ADDR_COND] = neg
        t[T.COL_MEM_RC_VALUE] = neg
        t[T.COL_MEM_REGION_PROPHET] = 1
        t[T.COL_MEM_S_PROPHET, 1:] = 1
        t[T.COL_MEM_DIFF_ADDR, 1:] = 1
        t[T.COL_MEM_DIFF_ADDR_INV, 1:] = 1
        return t
    t[T.COL_MEM_S_PROPHET] = 1
    t[T.COL_MEM_IS_WRITE] = 1                        # row 0: stack region, address 0, written once
    i = np.arange(1, n, dtype=np.uint64)
    a = (np.uint64(addr) + (i - np.uint64(1))) % np.uint64(P)        # addr + i - 1 < 2^64 for every n used here
    t[T.COL_MEM_ADDR, 1:] = a
    t[T.COL_MEM_DIFF_ADDR, 1:] = 1
    t[T.COL_MEM_DIFF_ADDR_INV, 1:] = 1
    if n > 1:
        t[T.COL_MEM_DIFF_ADDR, 1] = addr
        t[T.COL_MEM_DIFF_ADDR_INV, 1] = pow(addr, P - 2, P)
    neg = (np.uint64(P) - a) % np.uint64(P)
    t[T.COL_MEM_DIFF_ADDR_COND, 1:] = neg
    t[T.COL_MEM_RC_VALUE, 1:] = neg
    t[T.COL_MEM_REGION_PROPHET, 1:] = 1
    return t


def bitwise_trace(beta, limb_bits=8, ops=(), looked_by_cpu=False, transcript=None, reference_quirks=False):
    """generation/builtin.rs:35-205 with `limb_bits`-wide limbs: the fixed AND/OR/XOR table, and one row per operation in
    `ops` = [(name, op0, op1)] (operands of 4 limbs) with its limbs, compressed limbs and the permuted lookup columns.
    The rows carry FILTER = 0 (nothing in the CPU table looks them up, but every bitwise constraint and in-table lookup
    is live on them) unless looked_by_cpu, in which case they are the looked-up side of the CPU's AND / OR / XOR rows.
    `transcript` (a factory of Challenger objects): derive the compress challenge from the twelve limb columns as
    generation/builtin.rs:120-131 does and return (trace, beta) instead of using `beta`.
    reference_quirks: the fourth limb of op0 / op1 / res is NOT stored -- generation/builtin.rs:66,71,76 write it to `OP*_LIMBS.end`, the
    exclusive end of the column range, where the next group's write replaces it, so the reference's limb-3 columns are zero whatever the
    operand (docs/EXPERIMENTS.md "a defect of the reference's bitwise trace generator"); everything derived from the limb columns follows."""
    size = 1 << limb_bits
    per = size * size
    n = next_pow2(max(size, 3 * per, len(ops)))
    t = np.zeros((T.COL_NUM_BITWISE, n), dtype=np.uint64)
    index = 0
    for op0 in range(size):
        t[T.BW_FIX_RANGE_CHECK_U8, op0] = op0
        for op1 in range(size):
            for k, (res, tag) in enumerate(((op0 & op1, T.op_mask("AND")), (op0 | op1, T.op_mask("OR")), (op0 ^ op1, T.op_mask("XOR")))):
                r = k * per + index
                t[T.BW_FIX_BITWSIE_OP0, r], t[T.BW_FIX_BITWSIE_OP1, r], t[T.BW_FIX_BITWSIE_RES, r], t[T.BW_FIX_TAG, r] = op0, op1, res, tag
            index += 1
    fn = {"AND": lambda x, y: x & y, "OR": lambda x, y: x | y, "XOR": lambda x, y: x ^ y}
    limbs = lambda v: [(v >> (limb_bits * i)) & (size - 1) for i in range(4)]
    for r, (name, x, y) in enumerate(ops):
        assert x < size ** 4 and y < size ** 4
        z = fn[name](x, y)
        t[T.BW_TAG, r], t[T.BW_OP0, r], t[T.BW_OP1, r], t[T.BW_RES, r] = T.op_mask(name), x, y, z
        t[T.BW_FILTER, r] = int(looked_by_cpu)
        for i, (lx, ly, lz) in enumerate(zip(limbs(x), limbs(y), limbs(z))):
            if reference_quirks and i == 3:
                continue
            t[T.BW_OP0_LIMBS.start + i, r], t[T.BW_OP1_LIMBS.start + i, r], t[T.BW_RES_LIMBS.start + i, r] = lx, ly, lz
    if transcript is not None:
        ch = transcript()
        for cols in (T.BW_OP0_LIMBS, T.BW_OP1_LIMBS, T.BW_RES_LIMBS):
            for i in range(4):
                ch.observe(t[cols.start + i])
        beta = ch.get()
    b = int(beta) % P
    compress = lambda tag, x, y, z: (tag + x * b + y * b * b + z * b * b * b) % P
    fix = [compress(int(t[T.BW_FIX_TAG, i]), int(t[T.BW_FIX_BITWSIE_OP0, i]), int(t[T.BW_FIX_BITWSIE_OP1, i]), int(t[T.BW_FIX_BITWSIE_RES, i]))
           for i in range(n)]
    t[T.BW_FIX_COMPRESS] = fix
    for r in range(len(ops)):
        for i in range(4):
            t[T.BW_COMPRESS_LIMBS.start + i, r] = compress(int(t[T.BW_TAG, r]), int(t[T.BW_OP0_LIMBS.start + i, r]), int(t[T.BW_OP1_LIMBS.start + i, r]),
                                                           int(t[T.BW_RES_LIMBS.start + i, r]))
    rc8 = [int(x) for x in t[T.BW_FIX_RANGE_CHECK_U8]]
    for i in range(4):
        for src, limbs_perm, off in ((T.BW_OP0_LIMBS, T.BW_OP0_LIMBS_PERMUTED, 0), (T.BW_OP1_LIMBS, T.BW_OP1_LIMBS_PERMUTED, 4),
                                     (T.BW_RES_LIMBS, T.BW_RES_LIMBS_PERMUTED, 8)):
            pi, pt = permuted_cols([int(v) for v in t[src.start + i]], rc8)
            t[limbs_perm.start + i], t[T.BW_FIX_RANGE_CHECK_U8_PERMUTED.start + off + i] = pi, pt
        pi, pt = permuted_cols([int(v) for v in t[T.BW_COMPRESS_LIMBS.start + i]], fix)
        t[T.BW_COMPRESS_PERMUTED.start + i], t[T.BW_FIX_COMPRESS_PERMUTED.start + i] = pi, pt
    return (t, b) if transcript is not None else t


def bitwise_padding_trace(beta, limb_bits=8):
    return bitwise_trace(beta, limb_bits)


def poseidon_padding_trace(n, live_rows=0):
    """ZERO-hash padding rows (generation/poseidon.rs); the first `live_rows` rows are the reference's golden row of
    hashing [1000, 1001, ...] instead -- a full non-trivial permutation, with its looked-up filters at 0."""
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    g = json.load(open(os.path.join(root, "tests", "golden", "poseidon_air_rows.json")))["rows"]
    t = np.tile(np.array(g["ZERO"], dtype=np.uint64)[:, None], (1, n))
    for r in range(min(live_rows, n)):
        t[:, r] = np.array(g["1000"], dtype=np.uint64)
    return t


def flag_padding_trace(ncols, n, flag_col):
    t = np.zeros((ncols, n), dtype=np.uint64)
    t[flag_col] = 1
    return t


def tape_padding_trace(n):
    t = np.zeros((T.NUM_COL_TAPE, n), dtype=np.uint64)
    t[T.COL_TAPE_OPCODE] = T.op_mask("TLOAD")
    return t


def program_padding_trace(n):
    return np.zeros((T.NUM_PROG_COLS, n), dtype=np.uint64)   # all-zero rows: compress = 0, lookups trivially hold


def empty_program_instance(log_n=3, range_bits=4, limb_bits=2, bitwise_beta=12345, program_beta=67890, log_n_cpu=None, log_n_mem=None,
                           live=None, log_n_poseidon=None):
    """12 traces in `enum Table` order + per-table params + compress_challenges, for ola_stark(range_bits, limb_bits).
    log_n_cpu / log_n_mem override the height of the two tables that dominate a real execution."""
    n = 1 << log_n
    # `live` (a numpy Generator): on top of the padding rows, the tables that can carry real rows without a CPU row
    # looking at them do so -- range-checked values, bitwise operations and full Poseidon permutations.  (Cmp rows cannot:
    # their filter column also drives the lookup from the CPU table, cmp_stark.rs:95-97.)
    cmp_rows_, rc_rows, bw_ops, pos_live = [], [], (), 0
    if live is not None:
        rc_rows = [(int(live.integers(0, 1 << (2 * range_bits))), 0, 0, 0, 0) for _ in range(6)]
        top = 1 << (4 * limb_bits)
        bw_ops = [(name, int(live.integers(0, top)), int(live.integers(0, top))) for name in ("AND", "OR", "XOR", "XOR", "AND")]
        pos_live = max(1, n // 2)
    traces = [
        cpu_padding_trace(1 << (log_n_cpu or log_n)), memory_padding_trace(1 << (log_n_mem or log_n)),
        bitwise_trace(bitwise_beta, limb_bits, bw_ops),
        generate_cmp_trace(cmp_rows_), generate_rc_trace(rc_rows, range_bits),
        poseidon_padding_trace(1 << log_n_poseidon, 1 << (log_n_poseidon - 1)) if log_n_poseidon else poseidon_padding_trace(n, pos_live),
        flag_padding_trace(T.NUM_POSEIDON_CHUNK_COLS, n, T.COL_POSEIDON_CHUNK_IS_PADDING_LINE),
        flag_padding_trace(T.NUM_COL_ST, n, T.COL_ST_IS_PADDING),
        tape_padding_trace(n),
        flag_padding_trace(T.NUM_COL_SCCALL, n, T.COL_SCCALL_IS_PADDING),
        program_padding_trace(n),
        flag_padding_trace(T.NUM_PROG_CHUNK_COLS, n, T.COL_PROG_CHUNK_IS_PADDING_LINE),
    ]
    params = [bitwise_beta, program_beta]            # bitwise (table 2) and program (table 10) take one parameter each
    compress = [0, 0, bitwise_beta, 0, 0, 0, 0, 0, 0, 0, program_beta, 0]
    # Padding rows are mostly zero columns, and numpy hands those out as never-written (calloc'd) memory: virtual pages without a
    # physical page behind them.  An executor writes every cell of its traces; to stand in for one, write every word here, so
    # that the prover's first upload of these tables does not pay a million host page faults that no real trace would cause
    # (measured: the first proof of a process 0.46 s with untouched padding, 0.32 s with this, 0.29 s warm -- DESIGN.md, cold start).
    for t in traces:
        np.bitwise_or(t, np.uint64(0), out=t)       # in place, stays uint64
    return traces, params, compress
