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
