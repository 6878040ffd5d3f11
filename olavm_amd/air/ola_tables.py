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
