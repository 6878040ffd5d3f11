"""OlaVM AIR tables, transcribed constraint by constraint from the reference (paths relative to
/root/reference/circuits/src).  Table ids follow `enum Table` (stark/ola_stark.rs:104-120)."""
from .dsl import AirTable, Col, CrossTableLookup, TableWithColumns

CPU, MEMORY, BITWISE, CMP, RANGECHECK, POSEIDON, POSEIDON_CHUNK, STORAGE_ACCESS, TAPE, SCCALL, PROGRAM, PROG_CHUNK = range(12)


# ------------------------------------------------------------------------------------------------ Cmp
# builtins/cmp/columns.rs:19-25
COL_CMP_OP0, COL_CMP_OP1, COL_CMP_GTE, COL_CMP_ABS_DIFF, COL_CMP_ABS_DIFF_INV, COL_CMP_FILTER_LOOKING_RC = range(6)
COL_NUM_CMP = 6


def cmp_table():
    """builtins/cmp/cmp_stark.rs:19-79 (constraint_degree 3, no permutation pairs)."""
    t = AirTable("cmp", COL_NUM_CMP, 3)
    op0, op1, gte = t.local(COL_CMP_OP0), t.local(COL_CMP_OP1), t.local(COL_CMP_GTE)
    abs_diff, abs_diff_inv = t.local(COL_CMP_ABS_DIFF), t.local(COL_CMP_ABS_DIFF_INV)
    one = t.const(1)
    t.constraint(gte * (one - gte))                             # :37 gte must be binary
    t.constraint(gte * (op0 - op1 - abs_diff))                  # :39
    t.constraint((one - gte) * (op1 - op0 - abs_diff))          # :40
    t.constraint((one - gte) * (one - abs_diff * abs_diff_inv))  # :42
    return t


def cmp_ctl_data_with_rangecheck(): return Col.singles([COL_CMP_ABS_DIFF])            # cmp_stark.rs:82-84
def cmp_ctl_filter_with_rangecheck(): return Col.single(COL_CMP_FILTER_LOOKING_RC)   # :86-88
def cmp_ctl_data_with_cpu(): return Col.singles([COL_CMP_OP0, COL_CMP_OP1, COL_CMP_GTE])  # :91-93
def cmp_ctl_filter_with_cpu(): return Col.single(COL_CMP_FILTER_LOOKING_RC)          # :95-97


# ------------------------------------------------------------------------------------------------ RangeCheck
# builtins/rangecheck/columns.rs:28-43
(RC_CPU_FILTER, RC_MEMORY_SORT_FILTER, RC_MEMORY_REGION_FILTER, RC_CMP_FILTER, RC_VAL, RC_LIMB_LO, RC_LIMB_HI,
 RC_LIMB_LO_PERMUTED, RC_LIMB_HI_PERMUTED, RC_FIX_RANGE_CHECK_U16, RC_FIX_RANGE_CHECK_U16_PERMUTED_LO,
 RC_FIX_RANGE_CHECK_U16_PERMUTED_HI) = range(12)
COL_NUM_RC = 12


def rangecheck_table(range_bits=16):
    """builtins/rangecheck/rangecheck_stark.rs:26-107 (degree 3, 4 permutation pairs).  `range_bits` is 16 in the
    reference (BASE = 1 << 16, :22-24); a smaller value gives the miniature table used by CPU-sized tests."""
    t = AirTable("rangecheck", COL_NUM_RC, 3)
    val, limb_lo, limb_hi = t.local(RC_VAL), t.local(RC_LIMB_LO), t.local(RC_LIMB_HI)
    base = t.const(1 << range_bits)
    t.constraint(val - (limb_lo + limb_hi * base))                                  # :44-48
    t.eval_lookups(RC_LIMB_LO_PERMUTED, RC_FIX_RANGE_CHECK_U16_PERMUTED_LO)          # :50-55
    t.eval_lookups(RC_LIMB_HI_PERMUTED, RC_FIX_RANGE_CHECK_U16_PERMUTED_HI)          # :56-61
    t.permutation_pair([(RC_LIMB_LO, RC_LIMB_LO_PERMUTED)])                          # :99-106
    t.permutation_pair([(RC_LIMB_HI, RC_LIMB_HI_PERMUTED)])
    t.permutation_pair([(RC_FIX_RANGE_CHECK_U16, RC_FIX_RANGE_CHECK_U16_PERMUTED_LO)])
    t.permutation_pair([(RC_FIX_RANGE_CHECK_U16, RC_FIX_RANGE_CHECK_U16_PERMUTED_HI)])
    return t


def rc_ctl_data_with_cmp(): return Col.singles([RC_VAL])
def rc_ctl_filter_with_cmp(): return Col.single(RC_CMP_FILTER)


def ctl_cmp_rangecheck(cmp_idx=CMP, rc_idx=RANGECHECK):
    """stark/ola_stark.rs:282-296: looking = RangeCheck (VAL where CMP_FILTER), looked = Cmp (abs_diff)."""
    return CrossTableLookup(
        [TableWithColumns(rc_idx, rc_ctl_data_with_cmp(), rc_ctl_filter_with_cmp())],
        TableWithColumns(cmp_idx, cmp_ctl_data_with_rangecheck(), cmp_ctl_filter_with_rangecheck()))


# ------------------------------------------------------------------------------------------------ opcodes
# core/src/vm/opcodes.rs:81-114 binary_bit_shift / binary_bit_mask
OPCODE_SHIFT = dict(ADD=31, MUL=30, EQ=29, ASSERT=28, MOV=27, JMP=26, CJMP=25, CALL=24, RET=23, MLOAD=22, MSTORE=21, END=20,
                    RC=19, AND=18, OR=17, XOR=16, NOT=15, NEQ=14, GTE=13, POSEIDON=12, SLOAD=11, SSTORE=10, TLOAD=9,
                    TSTORE=8, SCCALL=7, SIGCHECK=6)


def op_mask(name):
    return 1 << OPCODE_SHIFT[name]


REGISTER_NUM = 10      # core/src/program/mod.rs:13
CTX_REGISTER_NUM = 4   # core/src/program/mod.rs:15
NEG_ONE = -1


# ------------------------------------------------------------------------------------------------ SCCall
# builtins/sccall/columns.rs:4-20
COL_SCCALL_TX_IDX = 0
COL_SCCALL_CALLER_ENV_IDX = 1
COL_SCCALL_CALLER_EXE_CTX_RANGE = range(2, 2 + CTX_REGISTER_NUM)
COL_SCCALL_CALLER_CODE_CTX_RANGE = range(6, 6 + CTX_REGISTER_NUM)
COL_SCCALL_CALLER_OP1_IMM = 10
COL_SCCALL_CLK_CALLER_CALL = 11
COL_SCCALL_CLK_CALLER_RET = 12
COL_SCCALL_CALLER_REG_RANGE = range(13, 13 + REGISTER_NUM)
COL_SCCALL_CALLEE_ENV_IDX = 23
COL_SCCALL_CLK_CALLEE_END = 24
COL_SCCALL_IS_PADDING = 25
NUM_COL_SCCALL = 26


def sccall_table():
    """builtins/sccall/sccall_stark.rs:67-94 (constraint_degree() returns 1 -- SURVEY F10 -- reproduced as is)."""
    t = AirTable("sccall", NUM_COL_SCCALL, 1)
    t.constraint(t.local(COL_SCCALL_CLK_CALLER_RET) - t.local(COL_SCCALL_CLK_CALLER_CALL) - t.local(COL_SCCALL_CALLER_OP1_IMM))  # :78-82
    return t


def sccall_ctl_data():            # sccall_stark.rs:23-38
    res = [COL_SCCALL_TX_IDX, COL_SCCALL_CALLER_ENV_IDX] + list(COL_SCCALL_CALLER_EXE_CTX_RANGE) + list(COL_SCCALL_CALLER_CODE_CTX_RANGE)
    res += [COL_SCCALL_CLK_CALLER_CALL, COL_SCCALL_CALLER_OP1_IMM] + list(COL_SCCALL_CALLER_REG_RANGE) + [COL_SCCALL_CALLEE_ENV_IDX]
    return Col.singles(res)


def sccall_ctl_filter():          # :40-42
    return Col.linear_combination([(COL_SCCALL_IS_PADDING, NEG_ONE)], 1)


def sccall_ctl_data_end():        # :44-58
    res = [COL_SCCALL_TX_IDX, COL_SCCALL_CALLER_ENV_IDX] + list(COL_SCCALL_CALLER_EXE_CTX_RANGE) + list(COL_SCCALL_CALLER_CODE_CTX_RANGE)
    res += [COL_SCCALL_CLK_CALLER_CALL] + list(COL_SCCALL_CALLER_REG_RANGE) + [COL_SCCALL_CALLEE_ENV_IDX, COL_SCCALL_CLK_CALLEE_END]
    return Col.singles(res)


def sccall_ctl_filter_end():      # :60-62
    return Col.linear_combination([(COL_SCCALL_IS_PADDING, NEG_ONE)], 1)


# ------------------------------------------------------------------------------------------------ Tape
# builtins/tape/columns.rs:3-9
COL_TAPE_TX_IDX, COL_TAPE_IS_INIT_SEG, COL_TAPE_OPCODE, COL_TAPE_ADDR, COL_TAPE_VALUE, COL_TAPE_FILTER_LOOKED = range(6)
NUM_COL_TAPE = 6


def tape_table():
    """builtins/tape/tape_stark.rs:44-143 (degree 5)."""
    t = AirTable("tape", NUM_COL_TAPE, 5)
    lv, nv = t.local, t.next
    one = t.const(1)
    op_tload, op_tstore, op_sccall = t.const(op_mask("TLOAD")), t.const(op_mask("TSTORE")), t.const(op_mask("SCCALL"))
    opc = lv(COL_TAPE_OPCODE)
    t.constraint(opc * (opc - op_tstore) * (opc - op_tload) * (opc - op_sccall))                        # :62-67
    t.constraint_first_row(lv(COL_TAPE_TX_IDX))                                                          # :70
    d_tx = nv(COL_TAPE_TX_IDX) - lv(COL_TAPE_TX_IDX)
    t.constraint_transition(d_tx * (d_tx - one))                                                         # :71-74
    is_in_same_tx = one - d_tx                                                                           # :75
    t.constraint(lv(COL_TAPE_IS_INIT_SEG) * (one - lv(COL_TAPE_IS_INIT_SEG)))                            # :77
    t.constraint_transition((one - is_in_same_tx) * (one - nv(COL_TAPE_IS_INIT_SEG)))                    # :79-81
    t.constraint_transition(is_in_same_tx * (nv(COL_TAPE_IS_INIT_SEG) - lv(COL_TAPE_IS_INIT_SEG))
                            * (lv(COL_TAPE_IS_INIT_SEG) - nv(COL_TAPE_IS_INIT_SEG) - one))               # :82-86
    t.constraint(lv(COL_TAPE_IS_INIT_SEG) * opc * (opc - op_tload))                                      # :88-90
    t.constraint((one - lv(COL_TAPE_IS_INIT_SEG)) * (opc - op_tload) * (opc - op_tstore) * (opc - op_sccall))  # :92-97
    t.constraint_first_row(lv(COL_TAPE_ADDR))                                                            # :99
    t.constraint_transition((one - is_in_same_tx) * nv(COL_TAPE_ADDR))                                   # :100
    d_addr = nv(COL_TAPE_ADDR) - lv(COL_TAPE_ADDR)
    t.constraint_transition(is_in_same_tx * d_addr * (d_addr - one))                                     # :101-105
    t.constraint_transition(is_in_same_tx * (one - d_addr) * (nv(COL_TAPE_VALUE) - lv(COL_TAPE_VALUE)))  # :107-111
    t.constraint_transition(is_in_same_tx * (one - d_addr) * (nv(COL_TAPE_OPCODE) - op_tload))           # :112-116
    t.constraint(is_in_same_tx * d_addr * nv(COL_TAPE_OPCODE) * (nv(COL_TAPE_OPCODE) - op_tstore)
                 * (nv(COL_TAPE_OPCODE) - op_sccall))                                                    # :119-125
    t.constraint(opc * (opc - op_tload) * (one - lv(COL_TAPE_FILTER_LOOKED)))                            # :127-131
    return t


def tape_ctl_data(): return Col.singles([COL_TAPE_TX_IDX, COL_TAPE_OPCODE, COL_TAPE_ADDR, COL_TAPE_VALUE])  # tape_stark.rs:26-34
def tape_ctl_filter(): return Col.single(COL_TAPE_FILTER_LOOKED)                                            # :36-38


# ------------------------------------------------------------------------------------------------ Program
# program/columns.rs:3-17
COL_PROG_CODE_ADDR_RANGE = range(0, 4)
COL_PROG_PC, COL_PROG_INST, COL_PROG_COMP_PROG, COL_PROG_COMP_PROG_PERM = 4, 5, 6, 7
COL_PROG_EXEC_CODE_ADDR_RANGE = range(8, 12)
COL_PROG_EXEC_PC, COL_PROG_EXEC_INST, COL_PROG_EXEC_COMP_PROG, COL_PROG_EXEC_COMP_PROG_PERM = 12, 13, 14, 15
COL_PROG_FILTER_EXEC, COL_PROG_FILTER_PROG_CHUNK = 16, 17
NUM_PROG_COLS = 18


def program_table():
    """program/program_stark.rs:60-115 (degree 3; parameter 0 = the compress challenge beta, :70)."""
    t = AirTable("program", NUM_PROG_COLS, 3, n_params=1)
    lv = t.local
    beta = t.param(0)
    b2 = beta * beta          # beta.square()
    b3 = b2 * beta            # beta.cube()

    def compress(addr0, pc, inst, comp):
        return (lv(addr0) + lv(addr0 + 1) * beta + lv(addr0 + 2) * b2 + lv(addr0 + 3) * b3
                + lv(pc) * b2 * b2 + lv(inst) * b2 * b3 - lv(comp))
    t.constraint(compress(COL_PROG_CODE_ADDR_RANGE.start, COL_PROG_PC, COL_PROG_INST, COL_PROG_COMP_PROG))                     # :71-79
    t.constraint(compress(COL_PROG_EXEC_CODE_ADDR_RANGE.start, COL_PROG_EXEC_PC, COL_PROG_EXEC_INST, COL_PROG_EXEC_COMP_PROG))  # :80-88
    t.eval_lookups(COL_PROG_EXEC_COMP_PROG_PERM, COL_PROG_COMP_PROG_PERM)                                                      # :89-94
    t.permutation_pair([(COL_PROG_COMP_PROG, COL_PROG_COMP_PROG_PERM)])                                                        # :109-114
    t.permutation_pair([(COL_PROG_EXEC_COMP_PROG, COL_PROG_EXEC_COMP_PROG_PERM)])
    return t


def prog_ctl_data_by_cpu(): return Col.singles(list(COL_PROG_EXEC_CODE_ADDR_RANGE) + [COL_PROG_EXEC_PC, COL_PROG_EXEC_INST])   # :25-28
def prog_ctl_filter_by_cpu(): return Col.single(COL_PROG_FILTER_EXEC)                                                          # :30-32
def prog_ctl_data_by_program_chunk(): return Col.singles(list(COL_PROG_CODE_ADDR_RANGE) + [COL_PROG_PC, COL_PROG_INST])        # :34-36
def prog_ctl_filter_by_program_chunk(): return Col.single(COL_PROG_FILTER_PROG_CHUNK)                                          # :38-40


# ------------------------------------------------------------------------------------------------ ProgChunk
# program/columns.rs (second half)
COL_PROG_CHUNK_CODE_ADDR_RANGE = range(0, 4)
COL_PROG_CHUNK_START_PC = 4
COL_PROG_CHUNK_INST_RANGE = range(5, 13)
COL_PROG_CHUNK_CAP_RANGE = range(13, 17)
COL_PROG_CHUNK_HASH_RANGE = range(17, 29)
COL_PROG_CHUNK_IS_FIRST_LINE, COL_PROG_CHUNK_IS_RESULT_LINE = 29, 30
COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE = range(31, 39)
COL_PROG_CHUNK_IS_PADDING_LINE = 39
NUM_PROG_CHUNK_COLS = 40


def prog_chunk_table():
    """program/prog_chunk_stark.rs:67-175 (degree 4)."""
    t = AirTable("prog_chunk", NUM_PROG_CHUNK_COLS, 4)
    lv, nv = t.local, t.next
    one = t.const(1)
    lv_is_padding, nv_is_padding = lv(COL_PROG_CHUNK_IS_PADDING_LINE), nv(COL_PROG_CHUNK_IS_PADDING_LINE)
    lv_is_first_line, nv_is_first_line = lv(COL_PROG_CHUNK_IS_FIRST_LINE), nv(COL_PROG_CHUNK_IS_FIRST_LINE)
    lv_is_result_line = lv(COL_PROG_CHUNK_IS_RESULT_LINE)
    t.constraint(lv_is_padding * (one - lv_is_padding))                                                   # :89
    t.constraint_transition((nv_is_padding - lv_is_padding) * (nv_is_padding - lv_is_padding - one))      # :90-92
    t.constraint_first_row((one - lv_is_padding) * (one - lv_is_first_line))                              # :100
    t.constraint_transition((one - nv_is_padding) * (one - lv_is_result_line) * nv_is_first_line)         # :102-104
    t.constraint_transition((one - nv_is_padding) * lv_is_result_line * (one - nv_is_first_line))         # :106-108
    for c in COL_PROG_CHUNK_CODE_ADDR_RANGE:                                                              # :110-119
        t.constraint_transition((one - nv_is_padding) * (one - lv_is_result_line) * (nv(c) - lv(c)))
    t.constraint(lv_is_first_line * lv(COL_PROG_CHUNK_START_PC))                                          # :122
    t.constraint_transition((one - nv_is_padding) * (one - lv_is_result_line)
                            * (nv(COL_PROG_CHUNK_START_PC) - lv(COL_PROG_CHUNK_START_PC) - t.const(8)))   # :123-129
    for c in COL_PROG_CHUNK_CAP_RANGE:                                                                    # :132-134
        t.constraint(lv_is_first_line * lv(c))
    for cap_c, hash_c in zip(COL_PROG_CHUNK_CAP_RANGE, list(COL_PROG_CHUNK_HASH_RANGE)[8:]):              # :135-144
        t.constraint((one - nv_is_padding) * (one - nv_is_first_line) * (nv(cap_c) - lv(hash_c)))
    for c in COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE:                                                    # :147-154
        f = lv(c)
        t.constraint(f * (one - f))
        t.constraint((one - lv_is_padding) * (one - lv_is_result_line) * (one - f))
    t.constraint(lv_is_result_line * (one - lv(COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE.start)))          # :155-157
    fl = list(COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE)
    for a_c, p_c in zip(fl[:7], fl[1:]):                                                                  # :158-164
        after, pre = lv(a_c), lv(p_c)
        t.constraint(lv_is_result_line * (after - pre) * (one - (after - pre)))
    return t


def prog_chunk_ctl_data_to_program(i):                                                                    # :23-31
    res = Col.singles(COL_PROG_CHUNK_CODE_ADDR_RANGE)
    res.append(Col.linear_combination([(COL_PROG_CHUNK_START_PC, 1)], i))
    res.append(Col.single(COL_PROG_CHUNK_INST_RANGE.start + i))
    return res


def prog_chunk_ctl_filter_to_program(i): return Col.single(COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE.start + i)  # :33-35
def prog_chunk_ctl_data_to_poseidon():                                                                    # :37-45
    return Col.singles(list(COL_PROG_CHUNK_INST_RANGE) + list(COL_PROG_CHUNK_CAP_RANGE) + list(COL_PROG_CHUNK_HASH_RANGE))
def prog_chunk_ctl_filter_to_poseidon(): return Col.linear_combination([(COL_PROG_CHUNK_IS_PADDING_LINE, NEG_ONE)], 1)  # :47-49
def prog_chunk_ctl_data_to_storage_access():                                                              # :51-58
    return [Col.zero()] + Col.singles(list(COL_PROG_CHUNK_CODE_ADDR_RANGE) + list(COL_PROG_CHUNK_HASH_RANGE)[:4])
def prog_chunk_ctl_filter_to_storage_access(): return Col.single(COL_PROG_CHUNK_IS_RESULT_LINE)           # :59-61


# ------------------------------------------------------------------------------------------------ PoseidonChunk
# builtins/poseidon/columns.rs (second half)
(COL_POSEIDON_CHUNK_TX_IDX, COL_POSEIDON_CHUNK_ENV_IDX, COL_POSEIDON_CHUNK_CLK, COL_POSEIDON_CHUNK_OPCODE, COL_POSEIDON_CHUNK_OP0,
 COL_POSEIDON_CHUNK_OP1, COL_POSEIDON_CHUNK_DST, COL_POSEIDON_CHUNK_ACC_CNT) = range(8)
COL_POSEIDON_CHUNK_VALUE_RANGE = range(8, 16)
COL_POSEIDON_CHUNK_CAP_RANGE = range(16, 20)
COL_POSEIDON_CHUNK_HASH_RANGE = range(20, 32)
COL_POSEIDON_CHUNK_IS_EXT_LINE, COL_POSEIDON_CHUNK_IS_RESULT_LINE = 32, 33
COL_POSEIDON_CHUNK_IS_FIRST_PADDING_RANGE = range(34, 42)
COL_POSEIDON_CHUNK_FILTER_LOOKED_CPU = 42
COL_POSEIDON_CHUNK_FILTER_LOOKING_MEM_RANGE = range(43, 51)
COL_POSEIDON_CHUNK_FILTER_LOOKING_POSEIDON = 51
COL_POSEIDON_CHUNK_IS_PADDING_LINE = 52
NUM_POSEIDON_CHUNK_COLS = 53


def poseidon_chunk_table():
    """builtins/poseidon/poseidon_chunk_stark.rs:98-284 (degree 3)."""
    t = AirTable("poseidon_chunk", NUM_POSEIDON_CHUNK_COLS, 3)
    lv, nv = t.local, t.next
    one = t.const(1)
    PAD, EXT = COL_POSEIDON_CHUNK_IS_PADDING_LINE, COL_POSEIDON_CHUNK_IS_EXT_LINE
    t.constraint(lv(PAD) * (one - lv(PAD)))                                                    # :111-114
    t.constraint_transition((nv(PAD) - lv(PAD)) * (nv(PAD) - lv(PAD) - one))                   # :115-120
    t.constraint(lv(EXT) * (one - lv(EXT)))                                                    # :122-124
    for c in (COL_POSEIDON_CHUNK_TX_IDX, COL_POSEIDON_CHUNK_ENV_IDX, COL_POSEIDON_CHUNK_CLK, COL_POSEIDON_CHUNK_OPCODE,
              COL_POSEIDON_CHUNK_OP1, COL_POSEIDON_CHUNK_DST):                                 # :126-149
        t.constraint(nv(EXT) * (nv(c) - lv(c)))
    t.constraint_first_row((one - lv(PAD)) * lv(EXT))                                          # :151-153
    for c in COL_POSEIDON_CHUNK_IS_FIRST_PADDING_RANGE:                                        # :156-158
        t.constraint(lv(c) * (one - lv(c)))
    sum_is_first_padding = t.const(0)
    for c in COL_POSEIDON_CHUNK_IS_FIRST_PADDING_RANGE:                                        # :159-161 fold(P::ZEROS, sum + v)
        sum_is_first_padding = sum_is_first_padding + lv(c)
    t.constraint(sum_is_first_padding * (one - sum_is_first_padding))                          # :162

    def acc_addends(row):                                                                      # :165-186 scan + (1 - v)
        s, out = t.const(0), []
        for c in COL_POSEIDON_CHUNK_IS_FIRST_PADDING_RANGE:
            s = s + row(c)
            out.append(one - s)
        return out
    v_line_acc_addends = acc_addends(lv)
    n_v_line_acc_addends = acc_addends(nv)
    n_v_line_acc_total_addend = t.const(0)
    for v in n_v_line_acc_addends:                                                             # :187-189
        n_v_line_acc_total_addend = n_v_line_acc_total_addend + v
    t.constraint(nv(EXT) * (nv(COL_POSEIDON_CHUNK_ACC_CNT) - lv(COL_POSEIDON_CHUNK_ACC_CNT) - n_v_line_acc_total_addend))  # :191-196
    t.constraint(sum_is_first_padding * nv(EXT))                                               # :201
    t.constraint(sum_is_first_padding * (one - lv(COL_POSEIDON_CHUNK_IS_RESULT_LINE)))         # :202-203
    t.constraint(sum_is_first_padding * (lv(COL_POSEIDON_CHUNK_ACC_CNT) - lv(COL_POSEIDON_CHUNK_OP1)))  # :204-206
    t.constraint((lv(COL_POSEIDON_CHUNK_ACC_CNT) - lv(COL_POSEIDON_CHUNK_OP1)) * (one - nv(EXT)))       # :208-211
    for c in COL_POSEIDON_CHUNK_HASH_RANGE:                                                    # :213-217
        t.constraint((one - lv(EXT)) * lv(c))
    for col_hash, col_cap in zip(list(COL_POSEIDON_CHUNK_HASH_RANGE)[8:], COL_POSEIDON_CHUNK_CAP_RANGE):  # :218-224
        t.constraint(nv(EXT) * (nv(col_cap) - lv(col_hash)))
    t.constraint((one - lv(EXT)) * nv(EXT) * (nv(COL_POSEIDON_CHUNK_OP0) - lv(COL_POSEIDON_CHUNK_OP0)))  # :226-230
    t.constraint(lv(EXT) * nv(EXT) * (nv(COL_POSEIDON_CHUNK_OP0) - lv(COL_POSEIDON_CHUNK_OP0) - t.const(8)))  # :231-237
    FC = COL_POSEIDON_CHUNK_FILTER_LOOKED_CPU
    t.constraint((one - lv(PAD)) * (one - lv(EXT)) * (one - lv(FC)))                           # :239-243
    t.constraint((one - lv(PAD)) * lv(EXT) * lv(FC))                                           # :244-248
    t.constraint(lv(PAD) * lv(FC))                                                             # :249-251
    for c, addend in zip(COL_POSEIDON_CHUNK_FILTER_LOOKING_MEM_RANGE, v_line_acc_addends):     # :255-262
        t.constraint((one - lv(EXT)) * lv(c))
        t.constraint(lv(EXT) * (lv(c) - addend))
    FP = COL_POSEIDON_CHUNK_FILTER_LOOKING_POSEIDON
    t.constraint((one - lv(PAD)) * lv(EXT) * (one - lv(FP)))                                   # :264-268
    t.constraint((one - lv(PAD)) * (one - lv(EXT)) * lv(FP))                                   # :269-273
    return t


def pc_ctl_data_with_cpu():                                                                    # :23-35
    return Col.singles([COL_POSEIDON_CHUNK_TX_IDX, COL_POSEIDON_CHUNK_ENV_IDX, COL_POSEIDON_CHUNK_CLK, COL_POSEIDON_CHUNK_OPCODE,
                        COL_POSEIDON_CHUNK_OP0, COL_POSEIDON_CHUNK_OP1, COL_POSEIDON_CHUNK_DST])
def pc_ctl_filter_with_cpu(): return Col.single(COL_POSEIDON_CHUNK_FILTER_LOOKED_CPU)          # :37-39
def pc_ctl_data_with_mem_src(i):                                                               # :41-55
    return Col.singles([COL_POSEIDON_CHUNK_TX_IDX, COL_POSEIDON_CHUNK_ENV_IDX, COL_POSEIDON_CHUNK_CLK, COL_POSEIDON_CHUNK_OPCODE]) + [
        Col.linear_combination([(COL_POSEIDON_CHUNK_OP0, 1)], i), Col.single(COL_POSEIDON_CHUNK_VALUE_RANGE.start + i), Col.zero()]
def pc_ctl_filter_with_mem_src(i): return Col.single(COL_POSEIDON_CHUNK_FILTER_LOOKING_MEM_RANGE.start + i)  # :57-59
def pc_ctl_data_with_mem_dst(i):                                                               # :61-75
    return Col.singles([COL_POSEIDON_CHUNK_TX_IDX, COL_POSEIDON_CHUNK_ENV_IDX, COL_POSEIDON_CHUNK_CLK, COL_POSEIDON_CHUNK_OPCODE]) + [
        Col.linear_combination([(COL_POSEIDON_CHUNK_DST, 1)], i), Col.single(COL_POSEIDON_CHUNK_HASH_RANGE.start + i), Col.one()]
def pc_ctl_filter_with_mem_dst(): return Col.single(COL_POSEIDON_CHUNK_IS_RESULT_LINE)         # :77-79
def pc_ctl_data_with_poseidon():                                                               # :81-88
    return Col.singles(list(COL_POSEIDON_CHUNK_VALUE_RANGE) + list(COL_POSEIDON_CHUNK_CAP_RANGE) + list(COL_POSEIDON_CHUNK_HASH_RANGE))
def pc_ctl_filter_with_poseidon(): return Col.single(COL_POSEIDON_CHUNK_FILTER_LOOKING_POSEIDON)  # :90-92


# ------------------------------------------------------------------------------------------------ Poseidon
# builtins/poseidon/columns.rs (first half); POSEIDON_* sizes core/src/util/poseidon_utils.rs:6-9
FILTER_LOOKED_NORMAL, FILTER_LOOKED_TREEKEY, FILTER_LOOKED_STORAGE_LEAF, FILTER_LOOKED_STORAGE_BRANCH = range(4)
COL_POSEIDON_INPUT_RANGE = range(4, 16)
COL_POSEIDON_OUTPUT_RANGE = range(16, 28)
COL_POSEIDON_FULL_ROUND_0_1_STATE_RANGE = range(28, 40)
COL_POSEIDON_FULL_ROUND_0_2_STATE_RANGE = range(40, 52)
COL_POSEIDON_FULL_ROUND_0_3_STATE_RANGE = range(52, 64)
COL_POSEIDON_PARTIAL_ROUND_ELEMENT_RANGE = range(64, 86)
COL_POSEIDON_FULL_ROUND_1_0_STATE_RANGE = range(86, 98)
COL_POSEIDON_FULL_ROUND_1_1_STATE_RANGE = range(98, 110)
COL_POSEIDON_FULL_ROUND_1_2_STATE_RANGE = range(110, 122)
COL_POSEIDON_FULL_ROUND_1_3_STATE_RANGE = range(122, 134)
NUM_POSEIDON_COLS = 134


def poseidon_table():
    """builtins/poseidon/poseidon_stark.rs:58-150 (degree 7).  The permutation inside the constraints follows
    core/src/util/poseidon_utils.rs:289-376 (constant_layer_field, sbox_monomial, mds_layer_field,
    partial_first_constant_layer, mds_partial_layer_init, mds_partial_layer_fast_field) with our factorisation."""
    from . import poseidon_params as PP
    t = AirTable("poseidon", NUM_POSEIDON_COLS, 7)
    lv = t.local
    one = t.const(1)
    inp = list(COL_POSEIDON_INPUT_RANGE)
    for c in inp[9:12]:                                                                        # :69-77
        cap = lv(c)
        t.constraint(lv(FILTER_LOOKED_TREEKEY) * cap)
        t.constraint(lv(FILTER_LOOKED_STORAGE_LEAF) * cap)
        t.constraint(lv(FILTER_LOOKED_STORAGE_BRANCH) * cap)
    t.constraint(lv(FILTER_LOOKED_STORAGE_LEAF) * (one - lv(inp[8])))                          # :78-81

    def sbox(x):                                                                               # poseidon_utils.rs:295-300
        x2 = x * x
        x4 = x2 * x2
        x3 = x * x2
        return x3 * x4

    def constant_layer(state, round_ctr):                                                      # :289-293
        return [s + t.const(PP.RC[i + 12 * round_ctr]) for i, s in enumerate(state)]

    def mds_layer(state):                                                                      # :308-326
        out = []
        for r in range(12):
            res = t.const(0)
            for i in range(12):
                res = res + state[(i + r) % 12] * t.const(PP.MDS_CIRC[i])
            res = res + state[r] * t.const(PP.MDS_DIAG[r])
            out.append(res)
        return out

    full0 = {1: COL_POSEIDON_FULL_ROUND_0_1_STATE_RANGE, 2: COL_POSEIDON_FULL_ROUND_0_2_STATE_RANGE, 3: COL_POSEIDON_FULL_ROUND_0_3_STATE_RANGE}
    full1 = {0: COL_POSEIDON_FULL_ROUND_1_0_STATE_RANGE, 1: COL_POSEIDON_FULL_ROUND_1_1_STATE_RANGE,
             2: COL_POSEIDON_FULL_ROUND_1_2_STATE_RANGE, 3: COL_POSEIDON_FULL_ROUND_1_3_STATE_RANGE}
    state = [lv(c) for c in inp]                                                               # :83-85
    round_ctr = 0
    for r in range(4):                                                                         # :89-101
        state = constant_layer(state, round_ctr)
        if r != 0:
            for i in range(12):
                sbox_in = lv(full0[r].start + i)
                t.constraint(state[i] - sbox_in)
                state[i] = sbox_in
        state = mds_layer([sbox(s) for s in state])
        round_ctr += 1
    # partial rounds (:104-118)
    state = [s + t.const(PP.FAST_FIRST_C[i]) for i, s in enumerate(state)]
    init = []
    for r in range(11):
        acc = t.const(0)
        for c in range(11):
            acc = acc + state[c + 1] * t.const(PP.FAST_INIT[r * 11 + c])
        init.append(acc)
    state = [state[0]] + init
    for r in range(22):
        sbox_in = lv(COL_POSEIDON_PARTIAL_ROUND_ELEMENT_RANGE.start + r)
        t.constraint(state[0] - sbox_in)
        x0 = sbox(sbox_in)
        if r < 21:
            x0 = x0 + t.const(PP.FAST_POST_C[r])
        d = x0 * t.const(PP.MDS_CIRC[0] + PP.MDS_DIAG[0])
        for j in range(11):
            d = d + state[j + 1] * t.const(PP.FAST_VHAT[r * 11 + j])
        state = [d] + [x0 * t.const(PP.FAST_W[r * 11 + j]) + state[j + 1] for j in range(11)]
    round_ctr += 22
    for r in range(4):                                                                         # :121-131
        state = constant_layer(state, round_ctr)
        for i in range(12):
            sbox_in = lv(full1[r].start + i)
            t.constraint(state[i] - sbox_in)
            state[i] = sbox_in
        state = mds_layer([sbox(s) for s in state])
        round_ctr += 1
    for i in range(12):                                                                        # :133-136
        t.constraint(state[i] - lv(COL_POSEIDON_OUTPUT_RANGE.start + i))
    return t


def poseidon_ctl_data_cpu_tree_key(): return Col.singles(list(COL_POSEIDON_INPUT_RANGE) + list(COL_POSEIDON_OUTPUT_RANGE)[:4])  # :153-155
def poseidon_ctl_filter_cpu_tree_key(): return Col.single(FILTER_LOOKED_TREEKEY)                                               # :157-159
def poseidon_ctl_data_with_poseidon_chunk(): return Col.singles(list(COL_POSEIDON_INPUT_RANGE) + list(COL_POSEIDON_OUTPUT_RANGE))  # :161-163
def poseidon_ctl_filter_with_poseidon_chunk(): return Col.single(FILTER_LOOKED_NORMAL)                                         # :165-167
def poseidon_ctl_data_with_storage():                                                                                          # :169-177
    return Col.singles(list(COL_POSEIDON_INPUT_RANGE) + list(COL_POSEIDON_OUTPUT_RANGE)[:4] + [FILTER_LOOKED_STORAGE_LEAF, FILTER_LOOKED_STORAGE_BRANCH])
def poseidon_ctl_filter_with_storage(): return Col.sum([FILTER_LOOKED_STORAGE_LEAF, FILTER_LOOKED_STORAGE_BRANCH])             # :179-181


# ------------------------------------------------------------------------------------------------ StorageAccess
# builtins/storage/columns.rs
COL_ST_ACCESS_IDX = 0
COL_ST_PRE_ROOT_RANGE = range(1, 5)
COL_ST_ROOT_RANGE = range(5, 9)
COL_ST_IS_WRITE, COL_ST_LAYER, COL_ST_LAYER_BIT, COL_ST_ADDR_ACC = 9, 10, 11, 12
COL_ST_ADDR_RANGE = range(13, 17)
COL_ST_PRE_PATH_RANGE = range(17, 21)
COL_ST_PATH_RANGE = range(21, 25)
COL_ST_SIB_RANGE = range(25, 29)
COL_ST_HASH_TYPE = 29
COL_ST_PRE_HASH_RANGE = range(30, 34)
COL_ST_HASH_RANGE = range(34, 38)
(COL_ST_IS_LAYER_1, COL_ST_IS_LAYER_64, COL_ST_IS_LAYER_128, COL_ST_IS_LAYER_192, COL_ST_IS_LAYER_256, COL_ST_ACC_LAYER_MARKER,
 COL_ST_FILTER_IS_HASH_BIT_0, COL_ST_FILTER_IS_HASH_BIT_1, COL_ST_FILTER_IS_FOR_PROG, COL_ST_IS_PADDING) = range(38, 48)
NUM_COL_ST = 48


def storage_access_table():
    """builtins/storage/storage_access_stark.rs:110-334 (degree 4)."""
    t = AirTable("storage_access", NUM_COL_ST, 4)
    lv, nv = t.local, t.next
    one = t.const(1)
    lv_is_padding, nv_is_padding = lv(COL_ST_IS_PADDING), nv(COL_ST_IS_PADDING)
    lv_idx, nv_idx = lv(COL_ST_ACCESS_IDX), nv(COL_ST_ACCESS_IDX)
    lv_layer, nv_layer = lv(COL_ST_LAYER), nv(COL_ST_LAYER)
    d_idx = nv_idx - lv_idx
    c256 = t.const(256)
    t.constraint((one - lv_is_padding) * lv_is_padding)                                        # :131
    t.constraint_transition((nv_is_padding - lv_is_padding) * (nv_is_padding - lv_is_padding - one))  # :132-134
    t.constraint_first_row((one - lv_is_padding) * (lv_idx - one))                             # :136
    t.constraint_transition((one - nv_is_padding) * d_idx * (d_idx - one))                     # :137-141
    t.constraint_first_row((one - lv_is_padding) * (one - lv_layer))                           # :145
    t.constraint_transition((one - nv_is_padding) * (one - d_idx) * (nv_layer - lv_layer - one))   # :147-151
    t.constraint_transition((one - nv_is_padding) * d_idx * (lv_layer - c256))                 # :153-157
    t.constraint_transition((one - nv_is_padding) * d_idx * (nv_layer - one))                  # :158-162
    t.constraint((one - nv_is_padding) * (lv_layer - c256) * (nv_layer - lv_layer - one))      # :164-168
    for c in (COL_ST_IS_LAYER_1, COL_ST_IS_LAYER_64, COL_ST_IS_LAYER_128, COL_ST_IS_LAYER_192, COL_ST_IS_LAYER_256):  # :172-176
        t.constraint(lv(c) * (one - lv(c)))
    t.constraint_first_row((one - lv_is_padding) * (one - lv(COL_ST_IS_LAYER_1)))              # :178-179
    t.constraint_transition((one - nv_is_padding) * d_idx * (one - nv(COL_ST_IS_LAYER_1)))     # :180-184
    t.constraint((lv(COL_ST_LAYER) - one) * lv(COL_ST_IS_LAYER_1))                             # :186
    for n, c in ((64, COL_ST_IS_LAYER_64), (128, COL_ST_IS_LAYER_128), (192, COL_ST_IS_LAYER_192), (256, COL_ST_IS_LAYER_256)):  # :187-198
        t.constraint((lv(COL_ST_LAYER) - t.const(n)) * lv(c))
    t.constraint_transition((one - nv_is_padding) * (one - d_idx)
                            * (nv(COL_ST_ACC_LAYER_MARKER) - lv(COL_ST_ACC_LAYER_MARKER)
                               - (nv(COL_ST_IS_LAYER_1) + nv(COL_ST_IS_LAYER_64) + nv(COL_ST_IS_LAYER_128) + nv(COL_ST_IS_LAYER_192)
                                  + nv(COL_ST_IS_LAYER_256))))                                  # :201-212
    t.constraint_transition((one - nv_is_padding) * d_idx * (lv(COL_ST_ACC_LAYER_MARKER) - t.const(5)))  # :214-218
    t.constraint_transition((one - nv_is_padding) * d_idx * (lv(COL_ST_HASH_TYPE) - one))      # :222-226
    t.constraint_transition((one - nv_is_padding) * (one - d_idx) * lv(COL_ST_HASH_TYPE))      # :228-232
    for c in COL_ST_ROOT_RANGE:                                                                # :236-238
        t.constraint(nv_is_padding * (nv(c) - lv(c)))
    for pr, rt, ph, hs in zip(COL_ST_PRE_ROOT_RANGE, COL_ST_ROOT_RANGE, COL_ST_PRE_HASH_RANGE, COL_ST_HASH_RANGE):  # :240-270
        t.constraint_transition((one - nv_is_padding) * d_idx * (nv(pr) - lv(rt)))
        t.constraint_transition((one - nv_is_padding) * (one - d_idx) * (nv(pr) - lv(pr)))
        t.constraint_transition((one - nv_is_padding) * (one - d_idx) * (nv(rt) - lv(rt)))
        t.constraint(lv(COL_ST_IS_LAYER_1) * (lv(pr) - lv(ph)))
        t.constraint(lv(COL_ST_IS_LAYER_1) * (lv(rt) - lv(hs)))
    t.constraint(lv(COL_ST_LAYER_BIT) * (one - lv(COL_ST_LAYER_BIT)))                          # :274
    t.constraint_transition((one - lv(COL_ST_IS_LAYER_64) - lv(COL_ST_IS_LAYER_128) - lv(COL_ST_IS_LAYER_192) - lv(COL_ST_IS_LAYER_256))
                            * (nv(COL_ST_ADDR_ACC) - lv(COL_ST_ADDR_ACC) * t.const(2) - nv(COL_ST_LAYER_BIT)))  # :277-286
    for k, c in enumerate((COL_ST_IS_LAYER_64, COL_ST_IS_LAYER_128, COL_ST_IS_LAYER_192, COL_ST_IS_LAYER_256)):  # :288-299
        t.constraint(lv(c) * (lv(COL_ST_ADDR_ACC) - lv(COL_ST_ADDR_RANGE.start + k)))
    for col_hash, col_path in zip(COL_ST_HASH_RANGE, COL_ST_PATH_RANGE):                       # :302-310
        t.constraint_transition((one - nv_is_padding) * (one - d_idx) * (lv(col_path) - nv(col_hash)))
    t.constraint((one - lv_is_padding) * (lv(COL_ST_FILTER_IS_HASH_BIT_0) + lv(COL_ST_LAYER_BIT) - one))  # :313-316
    t.constraint((one - lv_is_padding) * (lv(COL_ST_FILTER_IS_HASH_BIT_1) - lv(COL_ST_LAYER_BIT)))       # :317-319
    t.constraint(lv_is_padding * lv(COL_ST_FILTER_IS_HASH_BIT_0))                              # :320
    t.constraint(lv_is_padding * lv(COL_ST_FILTER_IS_HASH_BIT_1))                              # :321
    t.constraint(lv(COL_ST_FILTER_IS_FOR_PROG) * lv(COL_ST_IS_WRITE))                          # :322
    t.constraint(lv(COL_ST_FILTER_IS_FOR_PROG) * (one - lv(COL_ST_IS_LAYER_256)))              # :323-324
    return t


def st_ctl_data_for_prog_chunk(): return Col.singles([COL_ST_IS_WRITE] + list(COL_ST_ADDR_RANGE) + list(COL_ST_PATH_RANGE))   # :22-26
def st_ctl_filter_for_prog_chunk(): return Col.single(COL_ST_FILTER_IS_FOR_PROG)                                             # :28-30
def st_ctl_data_with_cpu(): return Col.singles([COL_ST_ACCESS_IDX, COL_ST_IS_WRITE] + list(COL_ST_ADDR_RANGE) + list(COL_ST_PATH_RANGE))  # :32-36
def st_ctl_filter_with_cpu_sstore(): return Col.linear_combination([(COL_ST_IS_LAYER_256, 1), (COL_ST_FILTER_IS_FOR_PROG, NEG_ONE)], 0)  # :38-46


def _st_poseidon(first, second, hash_range):
    res = Col.singles(list(first) + list(second))
    res.append(Col.single(COL_ST_HASH_TYPE))
    res += [Col.zero(), Col.zero(), Col.zero()]
    res += Col.singles(hash_range)
    res.append(Col.single(COL_ST_IS_LAYER_256))
    res.append(Col.linear_combination([(COL_ST_IS_LAYER_256, NEG_ONE)], 1))
    return res


def st_ctl_data_with_poseidon_bit0(): return _st_poseidon(COL_ST_PATH_RANGE, COL_ST_SIB_RANGE, COL_ST_HASH_RANGE)            # :48-59
def st_ctl_data_with_poseidon_bit0_pre(): return _st_poseidon(COL_ST_PRE_PATH_RANGE, COL_ST_SIB_RANGE, COL_ST_PRE_HASH_RANGE)  # :60-71
def st_ctl_filter_with_poseidon_bit0(): return Col.single(COL_ST_FILTER_IS_HASH_BIT_0)                                        # :72-74
def st_ctl_data_with_poseidon_bit1(): return _st_poseidon(COL_ST_SIB_RANGE, COL_ST_PATH_RANGE, COL_ST_HASH_RANGE)            # :76-87
def st_ctl_data_with_poseidon_bit1_pre(): return _st_poseidon(COL_ST_SIB_RANGE, COL_ST_PRE_PATH_RANGE, COL_ST_PRE_HASH_RANGE)  # :88-99
def st_ctl_filter_with_poseidon_bit1(): return Col.single(COL_ST_FILTER_IS_HASH_BIT_1)                                        # :101-103


# ------------------------------------------------------------------------------------------------ Bitwise
# builtins/bitwise/columns.rs (the "//NN" comments there are off by one; COL_NUM_BITWISE = 59)
BW_FILTER, BW_TAG, BW_OP0, BW_OP1, BW_RES = range(5)
BW_OP0_LIMBS = range(5, 9)
BW_OP1_LIMBS = range(9, 13)
BW_RES_LIMBS = range(13, 17)
BW_OP0_LIMBS_PERMUTED = range(17, 21)
BW_OP1_LIMBS_PERMUTED = range(21, 25)
BW_RES_LIMBS_PERMUTED = range(25, 29)
BW_COMPRESS_LIMBS = range(29, 33)
BW_COMPRESS_PERMUTED = range(33, 37)
BW_FIX_RANGE_CHECK_U8 = 37
BW_FIX_RANGE_CHECK_U8_PERMUTED = range(38, 50)
BW_FIX_TAG, BW_FIX_BITWSIE_OP0, BW_FIX_BITWSIE_OP1, BW_FIX_BITWSIE_RES, BW_FIX_COMPRESS = 50, 51, 52, 53, 54
BW_FIX_COMPRESS_PERMUTED = range(55, 59)
COL_NUM_BITWISE = 59


def bitwise_table(limb_bits=8):
    """builtins/bitwise/bitwise_stark.rs:40-362 (degree 3; parameter 0 = compress challenge beta, :77).  `limb_bits` is 8
    in the reference (BASE = 1 << 8, :27); smaller values give the miniature table used by CPU-sized tests."""
    t = AirTable("bitwise", COL_NUM_BITWISE, 3, n_params=1)
    lv = t.local
    base = t.const(1 << limb_bits)

    def reduce_with_powers(cols, alpha):   # plonk_common.rs:116-128: Horner from the last term
        s = t.const(0)
        for c in reversed(list(cols)):
            s = s * alpha + lv(c)
        return s
    t.constraint(reduce_with_powers(BW_OP0_LIMBS, base) - lv(BW_OP0))                          # :59-63
    t.constraint(reduce_with_powers(BW_OP1_LIMBS, base) - lv(BW_OP1))                          # :65-68
    t.constraint(reduce_with_powers(BW_RES_LIMBS, base) - lv(BW_RES))                          # :70-73
    beta = t.param(0)
    for i in range(4):                                                                         # :78-86
        t.constraint(lv(BW_TAG) + lv(BW_OP0_LIMBS.start + i) * beta + lv(BW_OP1_LIMBS.start + i) * beta * beta
                     + lv(BW_RES_LIMBS.start + i) * beta * beta * beta - lv(BW_COMPRESS_LIMBS.start + i))
    for i in range(4):                                                                         # :88-111
        t.eval_lookups(BW_OP0_LIMBS_PERMUTED.start + i, BW_FIX_RANGE_CHECK_U8_PERMUTED.start + i)
    for i in range(4):                                                                         # :112-135
        t.eval_lookups(BW_OP1_LIMBS_PERMUTED.start + i, BW_FIX_RANGE_CHECK_U8_PERMUTED.start + 4 + i)
    for i in range(4):                                                                         # :136-159
        t.eval_lookups(BW_RES_LIMBS_PERMUTED.start + i, BW_FIX_RANGE_CHECK_U8_PERMUTED.start + 8 + i)
    for i in range(4):                                                                         # :161-184
        t.eval_lookups(BW_COMPRESS_PERMUTED.start + i, BW_FIX_COMPRESS_PERMUTED.start + i)
    for i in range(4):                                                                         # :351-361
        t.permutation_pair([(BW_COMPRESS_LIMBS.start + i, BW_COMPRESS_PERMUTED.start + i)])
    for i in range(4):
        t.permutation_pair([(BW_FIX_COMPRESS, BW_FIX_COMPRESS_PERMUTED.start + i)])
    return t


def bitwise_ctl_data_with_cpu(): return Col.singles([BW_TAG, BW_OP0, BW_OP1, BW_RES])          # :365-367
def bitwise_ctl_filter_with_cpu(): return Col.single(BW_FILTER)                                # :369-371


# ------------------------------------------------------------------------------------------------ Memory
# memory/columns.rs:2-31
(COL_MEM_TX_IDX, COL_MEM_ENV_IDX, COL_MEM_IS_RW, COL_MEM_ADDR, COL_MEM_CLK, COL_MEM_OP, COL_MEM_S_MLOAD, COL_MEM_S_MSTORE,
 COL_MEM_S_CALL, COL_MEM_S_RET, COL_MEM_S_TLOAD, COL_MEM_S_TSTORE, COL_MEM_S_SCCALL, COL_MEM_S_POSEIDON, COL_MEM_S_SSTORE,
 COL_MEM_S_SLOAD, COL_MEM_S_PROPHET, COL_MEM_IS_WRITE, COL_MEM_VALUE, COL_MEM_DIFF_ADDR, COL_MEM_DIFF_ADDR_INV, COL_MEM_DIFF_CLK,
 COL_MEM_DIFF_ADDR_COND, COL_MEM_RW_ADDR_UNCHANGED, COL_MEM_REGION_PROPHET, COL_MEM_REGION_HEAP, COL_MEM_RC_VALUE,
 COL_MEM_FILTER_LOOKING_RC, COL_MEM_FILTER_LOOKING_RC_COND) = range(29)
NUM_MEM_COLS = 29
ADDR_HEAP_PTR = 18446744060824649731           # memory_stark.rs:80
INIT_VALUE_HEAP_PTR = ADDR_HEAP_PTR + 1        # :81


def memory_table():
    """memory/memory_stark.rs:89-344 (degree 8), including the data-dependent `is_next_addr_heap_ptr` indicator
    (:290-307): with scalar packing it is 1 exactly when nv_addr == ADDR_HEAP_PTR at the evaluation point."""
    t = AirTable("memory", NUM_MEM_COLS, 8)
    lv, nv = t.local, t.next
    one = t.const(1)
    same_tx = one - nv(COL_MEM_TX_IDX) + lv(COL_MEM_TX_IDX)
    same_env = one - nv(COL_MEM_ENV_IDX) + lv(COL_MEM_ENV_IDX)
    t.constraint_transition((nv(COL_MEM_TX_IDX) - lv(COL_MEM_TX_IDX)) * same_tx)                 # :104-107
    t.constraint_transition(same_tx * (nv(COL_MEM_ENV_IDX) - lv(COL_MEM_ENV_IDX)) * same_env)     # :109-113
    p = t.const(0)                                                                               # :115
    span = t.const(2**32 - 1)                                                                    # :116
    addr_heap_ptr = t.const(ADDR_HEAP_PTR)
    is_rw = lv(COL_MEM_IS_RW)
    region_prophet, nv_region_prophet = lv(COL_MEM_REGION_PROPHET), nv(COL_MEM_REGION_PROPHET)
    region_heap, nv_region_heap = lv(COL_MEM_REGION_HEAP), nv(COL_MEM_REGION_HEAP)
    region_stack = one - lv(COL_MEM_REGION_HEAP) - lv(COL_MEM_REGION_PROPHET)
    nv_region_stack = one - nv(COL_MEM_REGION_HEAP) - nv(COL_MEM_REGION_PROPHET)
    is_write, nv_is_write = lv(COL_MEM_IS_WRITE), nv(COL_MEM_IS_WRITE)
    addr, nv_addr = lv(COL_MEM_ADDR), nv(COL_MEM_ADDR)
    nv_diff_addr_inv = nv(COL_MEM_DIFF_ADDR_INV)
    diff_addr, nv_diff_addr = lv(COL_MEM_DIFF_ADDR), nv(COL_MEM_DIFF_ADDR)
    rw_addr_unchanged, nv_rw_addr_unchanged = lv(COL_MEM_RW_ADDR_UNCHANGED), nv(COL_MEM_RW_ADDR_UNCHANGED)
    diff_addr_cond = lv(COL_MEM_DIFF_ADDR_COND)
    value, nv_value = lv(COL_MEM_VALUE), nv(COL_MEM_VALUE)
    diff_clk, rc_value = lv(COL_MEM_DIFF_CLK), lv(COL_MEM_RC_VALUE)
    filter_looking_rc, lv_filter_looking_rc_cond = lv(COL_MEM_FILTER_LOOKING_RC), lv(COL_MEM_FILTER_LOOKING_RC_COND)
    sel = [("MLOAD", COL_MEM_S_MLOAD), ("MSTORE", COL_MEM_S_MSTORE), ("CALL", COL_MEM_S_CALL), ("RET", COL_MEM_S_RET),
           ("TLOAD", COL_MEM_S_TLOAD), ("TSTORE", COL_MEM_S_TSTORE), ("SCCALL", COL_MEM_S_SCCALL), ("POSEIDON", COL_MEM_S_POSEIDON),
           ("SSTORE", COL_MEM_S_SSTORE), ("SLOAD", COL_MEM_S_SLOAD)]
    for name, c in sel:                                                                          # :158-167
        t.constraint((lv(COL_MEM_OP) - t.const(op_mask(name))) * lv(c))
    t.constraint((lv(COL_MEM_OP) - t.const(0)) * lv(COL_MEM_S_PROPHET))                          # :168 (op_prophet = 0)
    for _, c in sel + [("PROPHET", COL_MEM_S_PROPHET)]:                                          # :169-179
        t.constraint((one - lv(c)) * lv(c))
    acc = one                                                                                    # :180-193
    for _, c in sel + [("PROPHET", COL_MEM_S_PROPHET)]:
        acc = acc - lv(c)
    t.constraint(acc)
    t.constraint(is_rw * (one - is_rw))                                                          # :196
    t.constraint(lv(COL_MEM_IS_RW) * lv(COL_MEM_S_PROPHET))                                      # :197
    t.constraint((one - lv(COL_MEM_IS_RW)) * (one - lv(COL_MEM_S_PROPHET) - lv(COL_MEM_S_MLOAD)))  # :198-200
    t.constraint(lv(COL_MEM_IS_WRITE) * (one - lv(COL_MEM_S_MSTORE) - lv(COL_MEM_S_CALL) - lv(COL_MEM_S_TLOAD)
                                         - lv(COL_MEM_S_POSEIDON) - lv(COL_MEM_S_SLOAD) - lv(COL_MEM_S_PROPHET)))  # :202-211
    t.constraint((one - lv(COL_MEM_IS_WRITE)) * (one - lv(COL_MEM_S_MLOAD) - lv(COL_MEM_S_CALL) - lv(COL_MEM_S_RET) - lv(COL_MEM_S_TSTORE)
                                                 - lv(COL_MEM_S_SCCALL) - lv(COL_MEM_S_POSEIDON) - lv(COL_MEM_S_SSTORE) - lv(COL_MEM_S_SLOAD)))  # :212-223
    t.constraint(one - region_stack - region_heap - region_prophet)                              # :226
    t.constraint(region_stack * (one - region_stack))                                            # :227
    t.constraint(region_heap * (one - region_heap))                                              # :228
    t.constraint(region_prophet * (one - region_prophet))                                        # :229
    t.constraint(region_prophet * (p - addr - diff_addr_cond))                                   # :231
    t.constraint(region_heap * (p - span - addr - diff_addr_cond))                               # :232
    t.constraint_transition(same_tx * same_env * (nv_region_heap - region_heap - one) * (nv_addr - addr - nv_diff_addr))  # :240-245
    t.constraint_transition(same_tx * same_env * region_stack * nv_region_stack
                            * (one - nv_rw_addr_unchanged - nv_diff_addr * nv_diff_addr_inv))    # :247-253
    t.constraint_transition(same_tx * same_env * region_heap * nv_region_heap
                            * (one - nv_rw_addr_unchanged - nv_diff_addr * nv_diff_addr_inv))    # :254-260
    t.constraint(region_prophet * nv_region_prophet * (nv_addr - addr) * (nv_addr - addr - one))  # :265-267
    t.constraint(region_prophet * nv_region_prophet * (nv_addr - addr - one) * nv_is_write)      # :268-270
    t.constraint_first_row(is_rw * (one - is_write) * (addr - addr_heap_ptr))                    # :279
    t.constraint((nv(COL_MEM_TX_IDX) - lv(COL_MEM_TX_IDX)) * (nv(COL_MEM_ENV_IDX) - lv(COL_MEM_ENV_IDX)) * nv(COL_MEM_IS_RW)
                 * (one - nv_is_write) * (nv_addr - addr_heap_ptr))                              # :280-286
    t.constraint((nv_addr - addr) * (one - nv_is_write) * (nv_addr - addr_heap_ptr))             # :287-288
    t.constraint((one - nv_is_write) * (nv_value - value) * (nv_addr - addr_heap_ptr))           # :289-290
    is_next_addr_heap_ptr = t.is_zero(nv_addr - t.const(ADDR_HEAP_PTR))                          # :292-300
    t.constraint(is_next_addr_heap_ptr * (nv_addr - t.const(ADDR_HEAP_PTR)))                     # :301-303
    t.constraint((addr - t.const(ADDR_HEAP_PTR)) * is_next_addr_heap_ptr * (one - nv_is_write)
                 * (nv_value - t.const(INIT_VALUE_HEAP_PTR)))                                    # :304-309
    t.constraint_transition(same_tx * same_env * is_rw * (nv_region_heap - region_heap - one)
                            * (rc_value - rw_addr_unchanged * diff_clk) * (rc_value - (one - rw_addr_unchanged) * diff_addr))  # :312-319
    t.constraint_transition(same_tx * same_env * is_rw * rc_value * (nv_region_heap - region_heap - one) * (one - filter_looking_rc))  # :320-327
    t.constraint((one - lv_filter_looking_rc_cond) * region_heap)                                # :330
    t.constraint((one - lv_filter_looking_rc_cond) * region_prophet * (one - is_write))          # :331-333
    return t


def mem_ctl_data_mem_rc_diff_cond(): return [Col.single(COL_MEM_DIFF_ADDR_COND)]               # :19-21
def mem_ctl_filter_mem_rc_diff_cond(): return Col.single(COL_MEM_FILTER_LOOKING_RC_COND)       # :23-25
def mem_ctl_data_mem_sort_rc(): return [Col.single(COL_MEM_RC_VALUE)]                          # :27-29
def mem_ctl_filter_mem_sort_rc(): return Col.single(COL_MEM_FILTER_LOOKING_RC)                 # :31-33
def mem_ctl_data(): return Col.singles([COL_MEM_TX_IDX, COL_MEM_ENV_IDX, COL_MEM_CLK, COL_MEM_OP, COL_MEM_ADDR, COL_MEM_VALUE])  # :35-46
def mem_ctl_filter():                                                                          # :48-61
    return Col.sum([COL_MEM_S_MLOAD, COL_MEM_S_MSTORE, COL_MEM_S_CALL, COL_MEM_S_RET, COL_MEM_S_TLOAD, COL_MEM_S_TSTORE,
                    COL_MEM_S_SCCALL, COL_MEM_S_SSTORE, COL_MEM_S_SLOAD])
def mem_ctl_data_with_poseidon_chunk():                                                        # :63-75
    return Col.singles([COL_MEM_TX_IDX, COL_MEM_ENV_IDX, COL_MEM_CLK, COL_MEM_OP, COL_MEM_ADDR, COL_MEM_VALUE, COL_MEM_IS_WRITE])
def mem_ctl_filter_with_poseidon_chunk(): return Col.single(COL_MEM_S_POSEIDON)                # :77-79


# ------------------------------------------------------------------------------------------------ Cpu
# cpu/columns.rs:4-66
COL_TX_IDX, COL_ENV_IDX, COL_CALL_SC_CNT = 0, 1, 2
COL_ADDR_STORAGE_RANGE = range(3, 7)
COL_ADDR_CODE_RANGE = range(7, 11)
COL_TP, COL_CLK, COL_PC, COL_IS_EXT_LINE, COL_EXT_CNT = 11, 12, 13, 14, 15
COL_REGS = range(16, 26)
COL_INST, COL_OP1_IMM, COL_OPCODE, COL_IMM_VAL, COL_OP0, COL_OP1, COL_DST, COL_AUX0, COL_AUX1, COL_IDX_STORAGE = range(26, 36)
COL_S_OP0 = range(36, 46)
COL_S_OP1 = range(46, 56)
COL_S_DST = range(56, 66)
(COL_S_SIMPLE_ARITHMATIC_OP, COL_S_MOV, COL_S_JMP, COL_S_CJMP, COL_S_CALL, COL_S_RET, COL_S_MLOAD, COL_S_MSTORE, COL_S_END,
 COL_S_RC, COL_S_BITWISE, COL_S_NOT, COL_S_GTE, COL_S_PSDN, COL_S_SLOAD, COL_S_SSTORE, COL_S_TLOAD, COL_S_TSTORE,
 COL_S_CALL_SC) = range(66, 85)
NUM_OP_SELECTOR = 19
(COL_IS_ENTRY_SC, COL_IS_NEXT_LINE_DIFF_INST, COL_IS_NEXT_LINE_SAME_TX, COL_FILTER_TAPE_LOOKING, IS_SCCALL_EXT_LINE,
 COL_IS_STORAGE_EXT_LINE, COL_FILTER_SCCALL_END, COL_FILTER_LOOKING_PROG_IMM, COL_IS_PADDING) = range(85, 94)
NUM_CPU_COLS = 94


def cpu_table():
    """cpu/cpu_stark.rs:868-958 (degree 7) with its helper blocks (:329-790) and the opcode files
    cpu/{simple_arithmatic_op,mov,call,ret,mload,mstore,storage,tape,call_sc}.rs, in the reference's emission order."""
    t = AirTable("cpu", NUM_CPU_COLS, 7)
    lv, nv = t.local, t.next
    one = t.const(1)
    # ---- CpuAdjacentRowWrapper::from_vars (:822-865)
    regs = [lv(c) for c in COL_REGS]
    n_regs = [nv(c) for c in COL_REGS]
    lv_is_padding, nv_is_padding = lv(COL_IS_PADDING), nv(COL_IS_PADDING)
    lv_is_ext_inst = lv(COL_S_SLOAD) + lv(COL_S_SSTORE) + lv(COL_S_TLOAD) + lv(COL_S_TSTORE) + lv(COL_S_CALL_SC) + lv(COL_S_END)
    nv_is_ext_inst = nv(COL_S_SLOAD) + nv(COL_S_SSTORE) + nv(COL_S_TLOAD) + nv(COL_S_TSTORE) + nv(COL_S_CALL_SC) + nv(COL_S_END)
    lv_is_entry_sc = lv(COL_IS_ENTRY_SC)
    lv_ext_length = (lv(COL_S_SLOAD) + lv(COL_S_SSTORE) + lv(COL_S_TLOAD) * (lv(COL_OP0) * lv(COL_OP1) + (one - lv(COL_OP0)))
                     + lv(COL_S_TSTORE) * lv(COL_OP1) + lv(COL_S_CALL_SC) + lv(COL_S_END) * (one - lv_is_entry_sc))
    is_crossing_inst = lv(COL_IS_NEXT_LINE_DIFF_INST)
    is_in_same_tx = lv(COL_IS_NEXT_LINE_SAME_TX)

    # ---- constraint_wrapper_cols (:338-369)
    t.constraint(lv_is_padding * (lv_is_padding - one))
    t.constraint_transition((nv_is_padding - lv_is_padding) * (nv_is_padding - lv_is_padding - one))
    t.constraint(lv_is_padding * (lv(COL_S_END) - one))
    t.constraint(lv_is_entry_sc * nv(COL_ENV_IDX))
    t.constraint((one - nv_is_padding) * is_in_same_tx * (nv(COL_TX_IDX) - lv(COL_TX_IDX)))
    t.constraint_transition((one - nv_is_padding) * (one - is_in_same_tx) * (nv(COL_TX_IDX) - lv(COL_TX_IDX) - one))
    t.constraint(is_crossing_inst * (lv_ext_length - lv(COL_EXT_CNT)))
    # ---- constraint_tx_init (:371-404)
    t.constraint_first_row(lv(COL_TX_IDX))
    t.constraint_first_row(lv(COL_ENV_IDX))
    t.constraint_first_row(lv(COL_CALL_SC_CNT))
    t.constraint_first_row(lv(COL_CLK))
    t.constraint_first_row(lv(COL_PC))
    for c in COL_REGS:
        t.constraint_first_row(lv(c))
    t.constraint_transition(is_in_same_tx * (nv(COL_TX_IDX) - lv(COL_TX_IDX)))
    t.constraint_transition((one - is_in_same_tx) * nv(COL_ENV_IDX))
    t.constraint_transition((one - is_in_same_tx) * nv(COL_CALL_SC_CNT))
    t.constraint_transition((one - is_in_same_tx) * nv(COL_TP))
    t.constraint_transition((one - is_in_same_tx) * nv(COL_CLK))
    t.constraint_transition((one - is_in_same_tx) * nv(COL_PC))
    for c in COL_REGS:
        t.constraint_transition((one - is_in_same_tx) * nv(c))
    # ---- eval_packed_generic body (:886-918)
    t.constraint_transition((one - nv_is_padding) * (one - lv(COL_S_END)) * (nv(COL_TX_IDX) - lv(COL_TX_IDX)))
    t.constraint_transition((one - nv_is_padding) * lv_is_entry_sc * lv(COL_S_END) * (nv(COL_TX_IDX) - lv(COL_TX_IDX) - one))
    for i in range(CTX_REGISTER_NUM):
        t.constraint_transition((one - nv_is_padding) * (one - lv(COL_S_END)) * (one - lv(COL_S_CALL_SC))
                                * (nv(COL_ADDR_STORAGE_RANGE.start + i) - lv(COL_ADDR_STORAGE_RANGE.start + i)))
        t.constraint_transition((one - nv_is_padding) * (one - lv(COL_S_END)) * (one - lv(COL_S_CALL_SC))
                                * (nv(COL_ADDR_CODE_RANGE.start + i) - lv(COL_ADDR_CODE_RANGE.start + i)))
    t.constraint((one - lv(COL_IS_PADDING) - lv(COL_IS_EXT_LINE)) * lv(COL_OP1_IMM) * (one - lv(COL_FILTER_LOOKING_PROG_IMM)))
    t.constraint((one - lv(COL_IS_PADDING) - lv(COL_IS_EXT_LINE)) * (lv(COL_S_MLOAD) + lv(COL_S_MSTORE))
                 * (one - lv(COL_FILTER_LOOKING_PROG_IMM)))
    # ---- constraint_ext_lines (:645-688)
    t.constraint((one - lv_is_ext_inst) * lv(COL_IS_EXT_LINE))
    t.constraint(lv_is_ext_inst * (lv_ext_length - lv(COL_EXT_CNT)) * (one - nv(COL_IS_EXT_LINE)))
    t.constraint(lv_is_ext_inst * (one - lv(COL_IS_EXT_LINE)) * lv(COL_EXT_CNT))
    t.constraint(nv_is_ext_inst * nv(COL_IS_EXT_LINE) * (nv(COL_EXT_CNT) - lv(COL_EXT_CNT) - one))
    t.constraint(nv(COL_IS_EXT_LINE) * (nv(COL_OPCODE) - lv(COL_OPCODE)))
    for c in range(COL_S_SIMPLE_ARITHMATIC_OP, COL_S_SIMPLE_ARITHMATIC_OP + NUM_OP_SELECTOR):
        t.constraint(nv(COL_IS_EXT_LINE) * (nv(c) - lv(c)))
    t.constraint(nv(COL_IS_EXT_LINE) * (nv(COL_OP1_IMM) - lv(COL_OP1_IMM)))
    # ---- constraint_env_idx (:406-454)
    t.constraint_transition(lv(COL_S_CALL_SC) * is_crossing_inst * (nv(COL_CALL_SC_CNT) - lv(COL_CALL_SC_CNT) - one))
    t.constraint_transition(is_in_same_tx * (one - lv(COL_S_CALL_SC)) * (nv(COL_CALL_SC_CNT) - lv(COL_CALL_SC_CNT)))
    t.constraint(lv(COL_S_CALL_SC) * (one - is_crossing_inst) * (nv(COL_CALL_SC_CNT) - lv(COL_CALL_SC_CNT)))
    t.constraint(lv(COL_S_CALL_SC) * is_crossing_inst * (nv(COL_ENV_IDX) - lv(COL_CALL_SC_CNT)))
    t.constraint((one - lv(COL_S_CALL_SC) - lv(COL_S_END)) * (nv(COL_ENV_IDX) - lv(COL_ENV_IDX)))
    t.constraint(lv(COL_S_CALL_SC) * (one - is_crossing_inst) * (nv(COL_ENV_IDX) - lv(COL_ENV_IDX)))
    t.constraint(lv(COL_S_END) * lv(COL_IS_EXT_LINE) * (nv(COL_ENV_IDX) - lv(COL_ENV_IDX)))
    # ---- constraint_opcode_selector (:456-527)
    ops_to_op = [(COL_S_SIMPLE_ARITHMATIC_OP, 0), (COL_S_MOV, op_mask("MOV")), (COL_S_JMP, op_mask("JMP")), (COL_S_CJMP, op_mask("CJMP")),
                 (COL_S_CALL, op_mask("CALL")), (COL_S_RET, op_mask("RET")), (COL_S_MLOAD, op_mask("MLOAD")), (COL_S_MSTORE, op_mask("MSTORE")),
                 (COL_S_END, op_mask("END")), (COL_S_RC, op_mask("RC")), (COL_S_BITWISE, 0), (COL_S_NOT, op_mask("NOT")),
                 (COL_S_GTE, op_mask("GTE")), (COL_S_PSDN, op_mask("POSEIDON")), (COL_S_SLOAD, op_mask("SLOAD")),
                 (COL_S_SSTORE, op_mask("SSTORE")), (COL_S_TLOAD, op_mask("TLOAD")), (COL_S_TSTORE, op_mask("TSTORE")),
                 (COL_S_CALL_SC, op_mask("SCCALL"))]
    opc = lv(COL_OPCODE)
    t.constraint(lv(COL_S_SIMPLE_ARITHMATIC_OP) * (opc - t.const(op_mask("ADD"))) * (opc - t.const(op_mask("MUL")))
                 * (opc - t.const(op_mask("EQ"))) * (opc - t.const(op_mask("NEQ"))) * (opc - t.const(op_mask("ASSERT"))))
    t.constraint(lv(COL_S_BITWISE) * (opc - t.const(op_mask("AND"))) * (opc - t.const(op_mask("OR"))) * (opc - t.const(op_mask("XOR"))))
    for c, _ in ops_to_op:
        t.constraint(lv(c) * (one - lv(c)))
    sum_s_op = t.const(0)
    for c, _ in ops_to_op:
        sum_s_op = sum_s_op + lv(c)
    t.constraint(one - sum_s_op)
    cal_opcode = t.const(0)
    for c, m in ops_to_op:
        cal_opcode = cal_opcode + lv(c) * t.const(m)
    t.constraint((opc - cal_opcode) * (one - lv(COL_S_BITWISE) - lv(COL_S_SIMPLE_ARITHMATIC_OP)))
    # ---- constraint_instruction_encode (:529-581)
    s_op0s, s_op1s, s_dsts = [lv(c) for c in COL_S_OP0], [lv(c) for c in COL_S_OP1], [lv(c) for c in COL_S_DST]
    t.constraint(lv(COL_OP1_IMM) * (one - lv(COL_OP1_IMM)))
    instruction = lv(COL_OP1_IMM) * t.const(2 ** 62)
    for start_shift, sel in ((61, s_op0s), (51, s_op1s), (41, s_dsts)):
        for index, s in enumerate(reversed(sel)):
            instruction = instruction + s * t.const((2 ** start_shift) // (2 ** index))
    instruction = instruction + lv(COL_OPCODE)
    t.constraint((one - lv(COL_IS_EXT_LINE)) * (lv(COL_INST) - instruction))
    t.constraint((one - lv(COL_IS_EXT_LINE)) * (lv(COL_OP1_IMM) * (lv(COL_OP1) - lv(COL_IMM_VAL))))
    # ---- constraint_operands_mathches_registers (:583-643)
    not_ext = one - lv(COL_IS_EXT_LINE)
    for sel in (s_op0s, s_op1s, s_dsts):
        for s in sel:
            t.constraint(not_ext * s * (one - s))

    def ssum(xs):                     # Iterator::sum starts from P::ZEROS
        acc = t.const(0)
        for x in xs:
            acc = acc + x
        return acc
    sum_s_op0, sum_s_op1, sum_s_dst = ssum(s_op0s), ssum(s_op1s), ssum(s_dsts)
    t.constraint(not_ext * sum_s_op0 * (one - sum_s_op0))
    t.constraint(not_ext * sum_s_op1 * (one - sum_s_op1))
    t.constraint(not_ext * sum_s_dst * (one - sum_s_dst))
    op0_sum = ssum([s * r for s, r in zip(s_op0s, regs)])
    t.constraint(not_ext * sum_s_op0 * (lv(COL_OP0) - op0_sum))
    op1_sum = ssum([s * r for s, r in zip(s_op1s, regs)])
    t.constraint(not_ext * sum_s_op1 * (lv(COL_OP1) - op1_sum))
    dst_sum = ssum([s * r for s, r in zip(s_dsts, n_regs)])
    t.constraint(not_ext * sum_s_dst * (lv(COL_DST) - dst_sum))
    # ---- constraint_env_unchanged_clk (:690-711)
    t.constraint(nv(COL_IS_EXT_LINE) * (one - nv(COL_S_END)) * (nv(COL_CLK) - lv(COL_CLK)))
    t.constraint(is_in_same_tx * (one - lv(COL_S_CALL_SC) - lv(COL_S_END)) * (one - nv(COL_IS_EXT_LINE)) * (nv(COL_CLK) - lv(COL_CLK) - one))
    # ---- constraint_env_unchanged_pc (:713-756)
    t.constraint(nv(COL_IS_EXT_LINE) * (one - nv(COL_S_END)) * (nv(COL_CLK) - lv(COL_CLK)))
    instruction_size = ((one - lv(COL_S_MLOAD) - lv(COL_S_MSTORE)) * (one + lv(COL_OP1_IMM))
                        + (lv(COL_S_MLOAD) + lv(COL_S_MSTORE)) * t.const(2))
    pc_incr = (one - (lv(COL_S_JMP) + lv(COL_S_CJMP) + lv(COL_S_CALL) + lv(COL_S_RET))) * (lv(COL_PC) + instruction_size)
    pc_jmp = lv(COL_S_JMP) * lv(COL_OP1)
    pc_cjmp = lv(COL_S_CJMP) * ((one - lv(COL_OP0)) * (lv(COL_PC) + instruction_size) + lv(COL_OP0) * lv(COL_OP1))
    pc_call = lv(COL_S_CALL) * lv(COL_OP1)
    pc_ret = lv(COL_S_RET) * lv(COL_DST)
    t.constraint((one - nv(COL_IS_EXT_LINE)) * (one - lv(COL_S_END) - lv(COL_S_CALL_SC))
                 * (nv(COL_PC) - (pc_incr + pc_jmp + pc_cjmp + pc_call + pc_ret)))
    t.constraint((one - nv(COL_IS_EXT_LINE)) * lv(COL_S_CJMP) * lv(COL_OP0) * (one - lv(COL_OP0)))
    # ---- constraint_reg_consistency (:758-790)
    multi_reg_change = (lv(COL_S_SLOAD) + lv(COL_S_PSDN) + lv(COL_S_CALL_SC) * is_crossing_inst
                        + lv(COL_S_END) * (one - lv(COL_IS_EXT_LINE)))
    for dst, l_r, n_r in zip(s_dsts[:REGISTER_NUM - 1], regs[:REGISTER_NUM - 1], n_regs[:REGISTER_NUM - 1]):
        t.constraint_transition((one - multi_reg_change) * (one - dst) * (n_r - l_r))
    t.constraint_transition((one - lv(COL_S_RET) - lv(COL_S_CALL_SC) * is_crossing_inst - lv(COL_S_END))
                            * (one - s_dsts[REGISTER_NUM - 1]) * (n_regs[REGISTER_NUM - 1] - regs[REGISTER_NUM - 1]))

    # ---- cpu/simple_arithmatic_op.rs:8-53
    sa = lv(COL_S_SIMPLE_ARITHMATIC_OP)
    m = {k: t.const(op_mask(k)) for k in ("ADD", "MUL", "EQ", "NEQ", "ASSERT")}
    is_add = sa * (opc - m["MUL"]) * (opc - m["EQ"]) * (opc - m["NEQ"]) * (opc - m["ASSERT"])
    is_mul = sa * (opc - m["ADD"]) * (opc - m["EQ"]) * (opc - m["NEQ"]) * (opc - m["ASSERT"])
    is_eq = sa * (opc - m["ADD"]) * (opc - m["MUL"]) * (opc - m["NEQ"]) * (opc - m["ASSERT"])
    is_neq = sa * (opc - m["ADD"]) * (opc - m["MUL"]) * (opc - m["EQ"]) * (opc - m["ASSERT"])
    is_assert = sa * (opc - m["ADD"]) * (opc - m["MUL"]) * (opc - m["EQ"]) * (opc - m["NEQ"])
    t.constraint(is_add * (lv(COL_DST) - (lv(COL_OP0) + lv(COL_OP1))))
    t.constraint(is_mul * (lv(COL_DST) - lv(COL_OP0) * lv(COL_OP1)))
    op_diff = lv(COL_OP0) - lv(COL_OP1)
    diff_aux = op_diff * lv(COL_AUX0)
    res = lv(COL_DST)
    eq_cs = is_eq * (res * op_diff + (one - res) * (one - diff_aux))
    neq_cs = is_neq * ((one - res) * op_diff + res * (one - diff_aux))
    t.constraint(eq_cs + neq_cs)
    t.constraint(is_assert * (one - lv(COL_OP1)))
    # ---- cpu/mov.rs
    t.constraint(lv(COL_S_MOV) * (lv(COL_DST) - lv(COL_OP1)))
    # ---- cpu/call.rs
    two = one + one
    fp = lv(COL_REGS.stop - 1)
    op0_cs = lv(COL_OP0) + one - fp
    op1_cs = lv(COL_OP1_IMM) * (lv(COL_DST) - lv(COL_PC) - two) + (one - lv(COL_OP1_IMM)) * (lv(COL_DST) - lv(COL_PC) - one)
    aux0_cs = lv(COL_AUX0) - fp + two
    t.constraint(lv(COL_S_CALL) * (op0_cs + op1_cs + aux0_cs))
    # ---- cpu/ret.rs
    op0_cs = lv(COL_OP0) + one - fp
    dst_cs = lv(COL_DST) - nv(COL_PC)
    aux0_cs = lv(COL_AUX0) + one + one - fp
    t.constraint(lv(COL_S_RET) * (op0_cs + dst_cs + aux0_cs))
    t.constraint_transition(lv(COL_S_RET) * (nv(COL_REGS.stop - 1) - lv(COL_AUX1)))
    # ---- cpu/mload.rs, cpu/mstore.rs
    for s_col in (COL_S_MLOAD, COL_S_MSTORE):
        t.constraint(lv(s_col) * (one - lv(COL_OP1_IMM)) * (lv(COL_AUX0) - lv(COL_IMM_VAL)))
        t.constraint(lv(s_col) * lv(COL_OP1_IMM) * (lv(COL_AUX1) - lv(COL_OP0) - lv(COL_OP1)))
        t.constraint(lv(s_col) * (one - lv(COL_OP1_IMM)) * (lv(COL_AUX1) - lv(COL_OP0) - lv(COL_AUX0) * lv(COL_OP1)))
    # ---- cpu/storage.rs
    st_op = lv(COL_S_SSTORE) + lv(COL_S_SLOAD)
    ext = lv(COL_IS_EXT_LINE)
    t.constraint_first_row(lv(COL_IDX_STORAGE) - st_op)
    t.constraint_transition(nv(COL_IDX_STORAGE) - lv(COL_IDX_STORAGE) - nv(COL_IS_STORAGE_EXT_LINE))
    t.constraint(st_op * (one - ext) * (nv(COL_OP0) - lv(COL_OP0)))
    t.constraint(st_op * (one - ext) * (nv(COL_OP1) - lv(COL_OP1)))
    for base_col, op_col in ((COL_S_OP0.start, COL_OP0), (COL_S_OP1.start, COL_OP1)):
        t.constraint(st_op * ext * (lv(base_col) - lv(op_col)))
        for k in range(1, 4):
            t.constraint(st_op * ext * (lv(base_col + k) - lv(base_col + k - 1) - one))
    t.constraint(st_op * ext * (one - lv(COL_IS_STORAGE_EXT_LINE)))
    t.constraint((one - st_op) * lv(COL_IS_STORAGE_EXT_LINE))
    t.constraint(st_op * (one - ext) * lv(COL_IS_STORAGE_EXT_LINE))
    # ---- cpu/tape.rs
    n_ext = nv(COL_IS_EXT_LINE)
    S0 = COL_S_OP0.start
    t.constraint((nv(COL_S_TSTORE) + nv(COL_S_TLOAD)) * n_ext * (nv(COL_OP0) - lv(COL_OP0)))
    t.constraint((nv(COL_S_TSTORE) + nv(COL_S_TLOAD)) * n_ext * (nv(COL_OP1) - lv(COL_OP1)))
    t.constraint((lv(COL_S_TSTORE) + lv(COL_S_TLOAD)) * ext * n_ext * (nv(COL_AUX0) - lv(COL_AUX0) - one))
    t.constraint(lv(COL_S_TSTORE) * (one - ext) * (lv(COL_TP) - nv(S0)))
    t.constraint(lv(COL_S_TSTORE) * ext * n_ext * (nv(S0) - lv(S0) - one))
    t.constraint(lv(COL_S_TSTORE) * (one - n_ext) * (nv(COL_TP) - lv(S0) - one))
    t.constraint(lv(COL_S_TLOAD) * lv(COL_OP0) * (one - ext) * (nv(S0) + lv(COL_OP1) - lv(COL_TP)))
    t.constraint(lv(COL_S_TLOAD) * (one - lv(COL_OP0)) * (one - ext) * (nv(S0) - lv(COL_OP1)))
    t.constraint((lv(COL_S_TSTORE) + lv(COL_S_TLOAD)) * ext * n_ext * (nv(S0) - lv(S0) - one))
    t.constraint(lv(COL_S_TSTORE) * (one - ext) * (lv(COL_OP0) - nv(COL_AUX0)))
    t.constraint(lv(COL_S_TLOAD) * (one - ext) * (lv(COL_DST) - nv(COL_AUX0)))
    t.constraint(is_in_same_tx * (one - lv(COL_S_TSTORE) - nv(COL_S_CALL_SC)) * (nv(COL_TP) - lv(COL_TP)))
    t.constraint(lv(COL_S_TSTORE) * n_ext * (nv(COL_TP) - lv(COL_TP)))
    t.constraint(lv(COL_S_TSTORE) * (one - n_ext) * (nv(COL_TP) - lv(S0) - one))
    t.constraint((one - lv(COL_S_CALL_SC)) * nv(COL_S_CALL_SC) * (nv(COL_TP) - lv(COL_TP)))
    t.constraint(lv(COL_S_CALL_SC) * (one - ext) * (nv(COL_TP) - lv(COL_TP)))
    t.constraint(lv(COL_S_CALL_SC) * ext * (nv(COL_TP) - lv(COL_TP) - t.const(12)))
    ftl = lv(COL_FILTER_TAPE_LOOKING)
    t.constraint(ftl * (one - ftl))
    t.constraint(ftl * (one - lv(COL_S_TLOAD) - lv(COL_S_TSTORE)))
    t.constraint(ftl * (one - ext))
    t.constraint((lv(COL_S_TLOAD) + lv(COL_S_TSTORE)) * ext * (one - ftl))
    # ---- cpu/call_sc.rs
    for i in range(4):
        t.constraint(lv(COL_S_CALL_SC) * (one - ext) * (nv(S0 + i) - lv(COL_ADDR_STORAGE_RANGE.start + i)))
    for i in range(4):
        t.constraint(lv(COL_S_CALL_SC) * (one - ext) * (nv(S0 + 4 + i) - lv(COL_ADDR_CODE_RANGE.start + i)))
    t.constraint(lv(COL_S_CALL_SC) * (one - ext) * (nv(COL_OP0) - lv(COL_OP0)))
    t.constraint(lv(COL_S_CALL_SC) * (one - ext) * (nv(COL_OP1) - lv(COL_OP1)))
    t.constraint_transition(lv(COL_S_END) * (one - is_crossing_inst) * (lv(COL_ENV_IDX) - nv(COL_AUX0)))
    t.constraint_transition(lv(COL_S_END) * (one - is_crossing_inst) * (lv(COL_CLK) - nv(COL_AUX1)))
    t.constraint(lv(COL_S_CALL_SC) * is_crossing_inst * nv(COL_CLK))
    t.constraint(lv(COL_S_CALL_SC) * is_crossing_inst * nv(COL_PC))
    for i in range(REGISTER_NUM):
        t.constraint(lv(COL_S_CALL_SC) * is_crossing_inst * nv(COL_REGS.start + i))
    for i in range(CTX_REGISTER_NUM):
        t.constraint(lv(COL_S_CALL_SC) * is_crossing_inst * (nv(COL_ADDR_STORAGE_RANGE.start + i) - lv(COL_ADDR_STORAGE_RANGE.start + i)))
        t.constraint(lv(COL_S_CALL_SC) * is_crossing_inst * (nv(COL_ADDR_CODE_RANGE.start + i) - lv(COL_ADDR_CODE_RANGE.start + i)))
    t.constraint(lv(COL_S_END) * ext * (one - is_crossing_inst) * (nv(COL_PC) - lv(COL_PC)))
    t.constraint(lv(COL_S_END) * ext * (one - is_crossing_inst) * (nv(COL_CLK) - lv(COL_CLK)))
    sce = lv(IS_SCCALL_EXT_LINE)
    t.constraint(sce * (one - sce))
    t.constraint((one - lv(COL_S_CALL_SC)) * sce)
    t.constraint(lv(COL_S_CALL_SC) * ext * (one - sce))
    t.constraint(lv(COL_S_CALL_SC) * (one - ext) * sce)
    fse = lv(COL_FILTER_SCCALL_END)
    t.constraint(fse * (one - fse))
    t.constraint((one - lv(COL_S_END)) * fse)
    t.constraint(lv(COL_S_END) * (one - ext) * fse)
    t.constraint(lv(COL_S_END) * ext * (one - fse))
    return t


# ---- cpu/cpu_stark.rs:20-326 CTL columns / filters ----
def cpu_ctl_data_cpu_mem_store_load(): return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, COL_AUX1, COL_DST])
def cpu_ctl_filter_cpu_mem_store_load(): return Col.sum([COL_S_MSTORE, COL_S_MLOAD])
def cpu_ctl_data_cpu_mem_call_ret_pc(): return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, COL_OP0, COL_DST])
def cpu_ctl_data_cpu_mem_call_ret_fp(): return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, COL_AUX0, COL_AUX1])
def cpu_ctl_filter_cpu_mem_call_ret(): return Col.sum([COL_S_CALL, COL_S_RET])
def cpu_ctl_data_cpu_mem_tload_tstore(): return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, COL_AUX0, COL_AUX1])
def cpu_ctl_filter_cpu_mem_tload_tstore(): return Col.single(COL_FILTER_TAPE_LOOKING)


def cpu_ctl_data_cpu_mem_sccall(i):
    col_addr = [COL_OP0, COL_DST, COL_AUX0, COL_AUX1][i]
    col_value = COL_ADDR_CODE_RANGE.start + i
    return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, col_addr, col_value])


def cpu_ctl_filter_cpu_mem_sccall(): return Col.single(IS_SCCALL_EXT_LINE)
def cpu_ctl_data_with_bitwise(): return Col.singles([COL_OPCODE, COL_OP0, COL_OP1, COL_DST])
def cpu_ctl_filter_with_bitwise(): return Col.single(COL_S_BITWISE)
def cpu_ctl_data_with_cmp(): return Col.singles([COL_OP0, COL_OP1, COL_DST])
def cpu_ctl_filter_with_cmp(): return Col.single(COL_S_GTE)
def cpu_ctl_data_with_rangecheck(): return Col.singles([COL_OP1])
def cpu_ctl_filter_with_rangecheck(): return Col.single(COL_S_RC)
def cpu_ctl_data_with_poseidon_chunk(): return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, COL_OP0, COL_OP1, COL_DST])
def cpu_ctl_filter_with_poseidon_chunk(): return Col.single(COL_S_PSDN)
def cpu_ctl_data_cpu_tape_load_store(): return Col.singles([COL_TX_IDX, COL_OPCODE, COL_S_OP0.start, COL_AUX1])
def cpu_ctl_filter_cpu_tape_load_store(): return Col.single(COL_FILTER_TAPE_LOOKING)


def cpu_ctl_data_poseidon_treekey():
    res = Col.singles(list(COL_ADDR_STORAGE_RANGE) + list(COL_S_OP0)[4:8])
    res += [Col.zero(), Col.zero(), Col.zero(), Col.zero()]
    res += Col.singles(list(COL_S_DST)[:4])
    return res


def cpu_ctl_filter_poseidon_treekey(): return Col.single(COL_IS_STORAGE_EXT_LINE)


def cpu_ctl_data_cpu_storage_access():
    return Col.singles([COL_IDX_STORAGE, COL_S_SSTORE, COL_S_DST.start, COL_S_DST.start + 1, COL_S_DST.start + 2, COL_S_DST.start + 3,
                        COL_S_OP1.start + 4, COL_S_OP1.start + 5, COL_S_OP1.start + 6, COL_S_OP1.start + 7])


def cpu_ctl_filter_cpu_storage_access(): return Col.single(COL_IS_STORAGE_EXT_LINE)
def cpu_ctl_data_cpu_mem_for_storage_addr(i): return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, COL_S_OP0.start + i, COL_S_OP0.start + 4 + i])
def cpu_ctl_data_cpu_mem_for_storage_value(i): return Col.singles([COL_TX_IDX, COL_ENV_IDX, COL_CLK, COL_OPCODE, COL_S_OP1.start + i, COL_S_OP1.start + 4 + i])


def cpu_ctl_data_cpu_sccall():
    res = [Col.single(COL_TX_IDX), Col.single(COL_ENV_IDX)]
    res += Col.singles(range(COL_S_OP0.start, COL_S_OP0.start + 4))
    res += Col.singles(range(COL_S_OP0.start + 4, COL_S_OP0.start + 8))
    res += Col.singles([COL_CLK, COL_OP1_IMM])
    res += Col.singles(COL_REGS)
    res.append(Col.linear_combination([(COL_ENV_IDX, 1)], 1))
    return res


def cpu_ctl_filter_cpu_sccall(): return Col.single(IS_SCCALL_EXT_LINE)


def cpu_ctl_data_cpu_sccall_end():
    return Col.singles([COL_TX_IDX, COL_ENV_IDX] + list(COL_ADDR_STORAGE_RANGE) + list(COL_ADDR_CODE_RANGE) + [COL_CLK] + list(COL_REGS)
                       + [COL_AUX0, COL_AUX1])


def cpu_ctl_filter_cpu_sccall_end(): return Col.single(COL_FILTER_SCCALL_END)
def cpu_ctl_data_inst_to_program(): return Col.singles(list(COL_ADDR_CODE_RANGE) + [COL_PC, COL_INST])
def cpu_ctl_data_imm_to_program(): return Col.singles(COL_ADDR_CODE_RANGE) + [Col.linear_combination([(COL_PC, 1)], 1), Col.single(COL_IMM_VAL)]
def cpu_ctl_filter_with_program_inst(): return Col.linear_combination([(COL_IS_EXT_LINE, NEG_ONE), (COL_IS_PADDING, NEG_ONE)], 1)
def cpu_ctl_filter_with_program_imm(): return Col.single(COL_FILTER_LOOKING_PROG_IMM)
def cpu_ctl_data_cpu_tape_sccall_caller(i): return [Col.single(COL_TX_IDX), Col.single(COL_OPCODE), Col.linear_combination([(COL_TP, 1)], i), Col.single(COL_S_OP0.start + i)]
def cpu_ctl_data_cpu_tape_sccall_callee_code(i): return [Col.single(COL_TX_IDX), Col.single(COL_OPCODE), Col.linear_combination([(COL_TP, 1)], 4 + i), Col.single(COL_ADDR_CODE_RANGE.start + i)]
def cpu_ctl_data_cpu_tape_sccall_callee_storage(i): return [Col.single(COL_TX_IDX), Col.single(COL_OPCODE), Col.linear_combination([(COL_TP, 1)], 8 + i), Col.single(COL_ADDR_STORAGE_RANGE.start + i)]
def cpu_ctl_filter_cpu_is_sccall_ext(): return Col.single(IS_SCCALL_EXT_LINE)


# ---- builtins/rangecheck/rangecheck_stark.rs:110-139 ----
def rc_ctl_data_memory(): return Col.singles([RC_VAL])
def rc_ctl_filter_memory_sort(): return Col.single(RC_MEMORY_SORT_FILTER)
def rc_ctl_filter_memory_region(): return Col.single(RC_MEMORY_REGION_FILTER)
def rc_ctl_data_with_cpu(): return Col.singles([RC_VAL])
def rc_ctl_filter_with_cpu(): return Col.single(RC_CPU_FILTER)


# ------------------------------------------------------------------------------------------------ OlaStark
def all_cross_table_lookups():
    """stark/ola_stark.rs:122-560, same order as all_cross_table_lookups() (:122-144)."""
    TW = TableWithColumns
    ctl_cpu_memory = CrossTableLookup(
        [TW(CPU, cpu_ctl_data_cpu_mem_store_load(), cpu_ctl_filter_cpu_mem_store_load()),
         TW(CPU, cpu_ctl_data_cpu_mem_call_ret_pc(), cpu_ctl_filter_cpu_mem_call_ret()),
         TW(CPU, cpu_ctl_data_cpu_mem_call_ret_fp(), cpu_ctl_filter_cpu_mem_call_ret()),
         TW(CPU, cpu_ctl_data_cpu_mem_tload_tstore(), cpu_ctl_filter_cpu_mem_tload_tstore())]
        + [TW(CPU, cpu_ctl_data_cpu_mem_sccall(i), cpu_ctl_filter_cpu_mem_sccall()) for i in range(4)]
        + [TW(CPU, cpu_ctl_data_cpu_mem_for_storage_addr(i), cpu_ctl_filter_cpu_storage_access()) for i in range(4)]
        + [TW(CPU, cpu_ctl_data_cpu_mem_for_storage_value(i), cpu_ctl_filter_cpu_storage_access()) for i in range(4)],
        TW(MEMORY, mem_ctl_data(), mem_ctl_filter()))
    ctl_memory_rc_sort = CrossTableLookup([TW(MEMORY, mem_ctl_data_mem_sort_rc(), mem_ctl_filter_mem_sort_rc())],
                                          TW(RANGECHECK, rc_ctl_data_memory(), rc_ctl_filter_memory_sort()))
    ctl_memory_rc_region = CrossTableLookup([TW(MEMORY, mem_ctl_data_mem_rc_diff_cond(), mem_ctl_filter_mem_rc_diff_cond())],
                                            TW(RANGECHECK, rc_ctl_data_memory(), rc_ctl_filter_memory_region()))
    ctl_bitwise_cpu = CrossTableLookup([TW(CPU, cpu_ctl_data_with_bitwise(), cpu_ctl_filter_with_bitwise())],
                                       TW(BITWISE, bitwise_ctl_data_with_cpu(), bitwise_ctl_filter_with_cpu()))
    ctl_cmp_cpu = CrossTableLookup([TW(CPU, cpu_ctl_data_with_cmp(), cpu_ctl_filter_with_cmp())],
                                   TW(CMP, cmp_ctl_data_with_cpu(), cmp_ctl_filter_with_cpu()))
    ctl_rangecheck_cpu = CrossTableLookup([TW(CPU, cpu_ctl_data_with_rangecheck(), cpu_ctl_filter_with_rangecheck())],
                                          TW(RANGECHECK, rc_ctl_data_with_cpu(), rc_ctl_filter_with_cpu()))
    ctl_cpu_poseidon_chunk = CrossTableLookup([TW(CPU, cpu_ctl_data_with_poseidon_chunk(), cpu_ctl_filter_with_poseidon_chunk())],
                                              TW(POSEIDON_CHUNK, pc_ctl_data_with_cpu(), pc_ctl_filter_with_cpu()))
    ctl_poseidon_chunk_mem = CrossTableLookup(
        [TW(POSEIDON_CHUNK, pc_ctl_data_with_mem_src(i), pc_ctl_filter_with_mem_src(i)) for i in range(8)]
        + [TW(POSEIDON_CHUNK, pc_ctl_data_with_mem_dst(i), pc_ctl_filter_with_mem_dst()) for i in range(4)],
        TW(MEMORY, mem_ctl_data_with_poseidon_chunk(), mem_ctl_filter_with_poseidon_chunk()))
    ctl_chunk_poseidon = CrossTableLookup(
        [TW(POSEIDON_CHUNK, pc_ctl_data_with_poseidon(), pc_ctl_filter_with_poseidon()),
         TW(PROG_CHUNK, prog_chunk_ctl_data_to_poseidon(), prog_chunk_ctl_filter_to_poseidon())],
        TW(POSEIDON, poseidon_ctl_data_with_poseidon_chunk(), poseidon_ctl_filter_with_poseidon_chunk()))
    ctl_cpu_poseidon_tree_key = CrossTableLookup([TW(CPU, cpu_ctl_data_poseidon_treekey(), cpu_ctl_filter_poseidon_treekey())],
                                                 TW(POSEIDON, poseidon_ctl_data_cpu_tree_key(), poseidon_ctl_filter_cpu_tree_key()))
    ctl_cpu_storage_access = CrossTableLookup([TW(CPU, cpu_ctl_data_cpu_storage_access(), cpu_ctl_filter_cpu_storage_access())],
                                              TW(STORAGE_ACCESS, st_ctl_data_with_cpu(), st_ctl_filter_with_cpu_sstore()))
    ctl_storage_access_poseidon = CrossTableLookup(
        [TW(STORAGE_ACCESS, st_ctl_data_with_poseidon_bit0(), st_ctl_filter_with_poseidon_bit0()),
         TW(STORAGE_ACCESS, st_ctl_data_with_poseidon_bit0_pre(), st_ctl_filter_with_poseidon_bit0()),
         TW(STORAGE_ACCESS, st_ctl_data_with_poseidon_bit1(), st_ctl_filter_with_poseidon_bit1()),
         TW(STORAGE_ACCESS, st_ctl_data_with_poseidon_bit1_pre(), st_ctl_filter_with_poseidon_bit1())],
        TW(POSEIDON, poseidon_ctl_data_with_storage(), poseidon_ctl_filter_with_storage()))
    ctl_cpu_tape = CrossTableLookup(
        [TW(CPU, cpu_ctl_data_cpu_tape_load_store(), cpu_ctl_filter_cpu_tape_load_store())]
        + [TW(CPU, cpu_ctl_data_cpu_tape_sccall_caller(i), cpu_ctl_filter_cpu_is_sccall_ext()) for i in range(4)]
        + [TW(CPU, cpu_ctl_data_cpu_tape_sccall_callee_code(i), cpu_ctl_filter_cpu_is_sccall_ext()) for i in range(4)]
        + [TW(CPU, cpu_ctl_data_cpu_tape_sccall_callee_storage(i), cpu_ctl_filter_cpu_is_sccall_ext()) for i in range(4)],
        TW(TAPE, tape_ctl_data(), tape_ctl_filter()))
    ctl_cpu_sccall = CrossTableLookup([TW(CPU, cpu_ctl_data_cpu_sccall(), cpu_ctl_filter_cpu_sccall())],
                                      TW(SCCALL, sccall_ctl_data(), sccall_ctl_filter()))
    ctl_cpu_sccall_end = CrossTableLookup([TW(CPU, cpu_ctl_data_cpu_sccall_end(), cpu_ctl_filter_cpu_sccall_end())],
                                          TW(SCCALL, sccall_ctl_data_end(), sccall_ctl_filter_end()))
    ctl_cpu_program = CrossTableLookup(
        [TW(CPU, cpu_ctl_data_inst_to_program(), cpu_ctl_filter_with_program_inst()),
         TW(CPU, cpu_ctl_data_imm_to_program(), cpu_ctl_filter_with_program_imm())],
        TW(PROGRAM, prog_ctl_data_by_cpu(), prog_ctl_filter_by_cpu()))
    ctl_prog_chunk_prog = CrossTableLookup(
        [TW(PROG_CHUNK, prog_chunk_ctl_data_to_program(i), prog_chunk_ctl_filter_to_program(i)) for i in range(8)],
        TW(PROGRAM, prog_ctl_data_by_program_chunk(), prog_ctl_filter_by_program_chunk()))
    ctl_prog_chunk_storage = CrossTableLookup(
        [TW(PROG_CHUNK, prog_chunk_ctl_data_to_storage_access(), prog_chunk_ctl_filter_to_storage_access())],
        TW(STORAGE_ACCESS, st_ctl_data_for_prog_chunk(), st_ctl_filter_for_prog_chunk()))
    return [ctl_cpu_memory, ctl_memory_rc_sort, ctl_memory_rc_region, ctl_bitwise_cpu, ctl_cmp_cpu, ctl_cmp_rangecheck(),
            ctl_rangecheck_cpu, ctl_cpu_poseidon_chunk, ctl_poseidon_chunk_mem, ctl_chunk_poseidon, ctl_cpu_poseidon_tree_key,
            ctl_cpu_storage_access, ctl_storage_access_poseidon, ctl_cpu_tape, ctl_cpu_sccall, ctl_cpu_sccall_end, ctl_cpu_program,
            ctl_prog_chunk_prog, ctl_prog_chunk_storage]


def ola_stark(range_bits=16, limb_bits=8):
    """The 12-table OlaStark (stark/ola_stark.rs:29-64), tables in `enum Table` order.  range_bits / limb_bits: see
    rangecheck_table / bitwise_table (16 / 8 in the reference)."""
    from .dsl import AirSet
    tables = [cpu_table(), memory_table(), bitwise_table(limb_bits), cmp_table(), rangecheck_table(range_bits), poseidon_table(), poseidon_chunk_table(),
              storage_access_table(), tape_table(), sccall_table(), program_table(), prog_chunk_table()]
    return AirSet(tables, all_cross_table_lookups())
