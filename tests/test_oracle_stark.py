"""CPU tests of the oracle's multi-table STARK prover/verifier and of the AIR descriptions (constraint programs vanish on
valid traces -- the reference's own AIR test recipe, circuits/src/test_utils.rs:152-195)."""
import numpy as np
import pytest

from olavm_amd.air import AirSet, ola_tables as T
from tests import tracegen


def mini_set(range_bits):
    return AirSet([T.cmp_table(), T.rangecheck_table(range_bits)], [T.ctl_cmp_rangecheck(0, 1)])


def test_dsl_blob_roundtrip_shapes():
    s = mini_set(16)
    b = s.blob()
    assert b[0] == 0x4F4C41414952 and b[2] == 2 and b[3] == 1
    assert T.rangecheck_table().num_permutation_batches() == 4     # perm Z = 4
    assert T.cmp_table().num_permutation_batches() == 0
    assert s.num_ctl_zs(0) == 2 and s.num_ctl_zs(1) == 2


@pytest.mark.parametrize("range_bits,n_cmp", [(4, 5), (5, 16)])
def test_constraints_vanish_on_valid_traces(oracle, range_bits, n_cmp):
    rng = np.random.default_rng(range_bits)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, n_cmp, range_bits)
    blob = mini_set(range_bits).blob()
    assert oracle.check_constraints(blob, 0, cmp_t) == -1
    assert oracle.check_constraints(blob, 1, rc_t) == -1
    bad = cmp_t.copy()
    bad[T.COL_CMP_GTE, 1] ^= 1
    assert oracle.check_constraints(blob, 0, bad) == 1
    bad = rc_t.copy()
    bad[T.RC_LIMB_LO, 2] += 1
    assert oracle.check_constraints(blob, 1, bad) >= 0


def test_two_table_proof_verifies_and_tampering_is_caught(oracle):
    rng = np.random.default_rng(9)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, 6, range_bits=4)
    blob = mini_set(4).blob()
    proof = oracle.prove_with_traces(blob, [cmp_t, rc_t])
    rc, why = oracle.verify_all_proof(blob, proof)
    assert rc == 0, why
    # deterministic
    assert proof == oracle.prove_with_traces(blob, [cmp_t, rc_t])
    # an opened value is changed -> rejected
    bad = bytearray(proof)
    bad[4 + 3 * (4 + 16 * 32) + 4 + 5] ^= 1
    rc, why = oracle.verify_all_proof(blob, bytes(bad))
    assert rc != 0
    # a CTL-inconsistent instance (one looked-up value missing) cannot be proven into an accepted proof
    rc_bad = rc_t.copy()
    rc_bad[T.RC_CMP_FILTER, 0] = 0
    p2 = oracle.prove_with_traces(blob, [cmp_t, rc_bad])
    rc, why = oracle.verify_all_proof(blob, p2)
    assert rc != 0 and "Cross-table" in why


def test_poseidon_air_vanishes_on_reference_golden_rows(oracle):
    """The reference ships the intermediate round states of two Poseidon permutations as constants
    (core/src/util/poseidon_utils.rs:11-287; the ZERO row is the Poseidon table's padding row).  Our transcription of
    PoseidonStark (with our own partial-round factorisation) must vanish on them."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_air_rows.json")))["rows"]
    blob = AirSet([T.poseidon_table()], []).blob()
    trace = np.array([g["ZERO"], g["1000"]], dtype=np.uint64).T.copy()     # 134 columns x 2 rows
    assert oracle.check_constraints(blob, 0, trace) == -1
    # outputs really are the permutation of the inputs
    for tag in ("ZERO", "1000"):
        row = g[tag]
        assert [int(x) for x in oracle.poseidon(np.array(row[4:16], dtype=np.uint64))] == row[16:28]
    bad = trace.copy()
    bad[70, 1] ^= 1                                                         # one partial-round s-box input
    assert oracle.check_constraints(blob, 0, bad) == 1


def _per_table_params(airset, params):
    out, k = [], 0
    for t in airset.tables:
        out.append(params[k:k + t.n_params] if t.n_params else None)
        k += t.n_params
    return out


def test_all_twelve_tables_vanish_on_the_empty_program_instance(oracle):
    """The reference's AIR test recipe (test_utils.rs:152-195) for every table, on padding rows as its trace generators
    emit them (olavm_amd/air/tracegen.py); miniature fixed tables (range_bits 4, limb_bits 2) keep it CPU-sized."""
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    traces, params, _ = tracegen.empty_program_instance()
    assert [t.shape[0] for t in traces] == [94, 29, 59, 6, 12, 134, 53, 48, 6, 26, 18, 40]     # widths
    for i, (tr, pr) in enumerate(zip(traces, _per_table_params(s, params))):
        assert oracle.check_constraints(blob, i, tr, pr) == -1, s.tables[i].name
    # and the programs do bite: one flipped cell per table is caught
    flips = {0: (T.COL_S_END, 1), 1: (T.COL_MEM_IS_WRITE, 2), 2: (T.BW_OP0_LIMBS.start, 1), 5: (T.COL_POSEIDON_INPUT_RANGE.start, 1),
             6: (T.COL_POSEIDON_CHUNK_IS_PADDING_LINE, 0), 8: (T.COL_TAPE_OPCODE, 1), 9: (T.COL_SCCALL_CLK_CALLER_RET, 0)}
    for i, (c, r) in flips.items():
        bad = traces[i].copy()
        bad[c, r] = (int(bad[c, r]) + 1) % tracegen.P
        assert oracle.check_constraints(blob, i, bad, _per_table_params(s, params)[i]) >= 0, s.tables[i].name


def test_twelve_table_all_proof_verifies(oracle):
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    # padding rows everywhere plus live range-check, bitwise and Poseidon rows (tracegen.empty_program_instance)
    traces, params, compress = tracegen.empty_program_instance(live=np.random.default_rng(5))
    for i, (tr, pr) in enumerate(zip(traces, _per_table_params(s, params))):
        assert oracle.check_constraints(blob, i, tr, pr) == -1, s.tables[i].name
    proof = oracle.prove_with_traces(blob, traces, params, compress)
    rc, why = oracle.verify_all_proof(blob, proof, params)
    assert rc == 0, why
    # the compress challenges travel in AllProof (prover.rs:307-320)
    assert proof != oracle.prove_with_traces(blob, traces, params, [0] * 12)


# ------------------------------------------------------------------------------------------------ real executions (f-1)
def _check_all_tables(oracle, s, traces, params):
    blob = s.blob()
    for i, (tr, pr) in enumerate(zip(traces, _per_table_params(s, params))):
        assert oracle.check_constraints(blob, i, tr, pr) == -1, s.tables[i].name


@pytest.mark.parametrize("program", ["fibonacci", "mixed", "memory", "hash", "call", "tape", "storage", "heap"])
def test_mini_executor_traces_are_valid_and_provable(oracle, program):
    """A real execution (olavm_amd/air/miniexec.py): CPU rows with live opcodes, the program table they are fetched
    from, the Poseidon-hashed program chunks, and -- for the mixed program -- bitwise, comparison and range-check rows
    behind the CPU's lookups.  All 251 CPU constraints and the other 11 AIRs vanish, and the proof passes the verifier's
    cross-table product check with 3 (fibonacci) / 7 (mixed) / 5 (memory: cpu<->memory and the memory table's range-checked
    sort columns) / 8 (hash: the Poseidon builtin reading and writing memory) / 9 (storage: state-tree proofs, tree keys,
    the program-hash leaf) of the 19 lookups carrying rows; over all programs 17 of the 19 do."""
    from olavm_amd.air import miniexec as M
    s = T.ola_stark(range_bits=4, limb_bits=2)
    factory, kwargs = M.EXAMPLES[program]
    prog = factory()
    rows, side, _ = M.execute(prog)
    if program == "fibonacci":
        assert (rows[-1][T.COL_REGS.start + 1], rows[-1][T.COL_REGS.start + 2]) == (5, 8) and len(rows) == 34
    elif program == "mixed":
        assert len(side["bitwise"]) == 3 and len(side["cmp"]) == 3 and side["rc"] == [200 & 77, 255]
    elif program == "memory":
        assert len(side["mem"]) == 24 and rows[-1][T.COL_REGS.start + 6] == 0 + 1 + 1 + 2 + 3 + 5
    elif program == "tape":
        assert rows[-1][T.COL_REGS.start + 7] == 11 + 13 + 7 and sum(r[T.COL_IS_EXT_LINE] for r in rows) == 6
    elif program == "call":
        assert rows[-1][T.COL_REGS.start + 1] == 3 * 8 * 8 and sum(1 for c in side["mem"] if c[2] in ("CALL", "RET")) == 8
    elif program == "storage":
        # the second SLOAD returned the overwriting value; 4 proofs of 256 levels; each SSTORE moved the root, no SLOAD did
        Pm = tracegen.P
        assert rows[-1][T.COL_REGS.start + 6] == ((Pm - 1) + (Pm - 4)) % Pm and [len(a) for a in side["storage"]] == [256] * 4
        roots = [(a[0]["pre_root"], a[0]["root"]) for a in side["storage"]]
        assert roots[0][0] != roots[0][1] == roots[1][0] == roots[1][1] != roots[2][1] == roots[3][1]
        leaf = side["storage"][0][255]          # lowest level: hash of the stored value and its (empty) sibling, capacity word 1
        pair = list(leaf["sib"]) + list(leaf["path"]) if leaf["bit"] else list(leaf["path"]) + list(leaf["sib"])
        assert tuple(int(x) for x in oracle.poseidon(np.array(pair + [1, 0, 0, 0], dtype=np.uint64))[:4]) == leaf["hash"]
        assert leaf["path"] == (1, 8, 15, 22) and leaf["pre_path"] == (0, 0, 0, 0)
    elif program == "heap":
        assert rows[-1][T.COL_REGS.start + 4] == 42 + 84 and sum(1 for c in side["mem"] if c[0] >= T.ADDR_HEAP_PTR) == 5
    else:       # the digest the program loads back is the sponge hash of the 16 words it stored
        words = np.array([3 * pow(5, i, tracegen.P) % tracegen.P for i in range(16)], dtype=np.uint64)
        digest = oracle.hash_no_pad(words)
        assert (rows[-1][T.COL_REGS.start + 6], rows[-1][T.COL_REGS.start + 7]) == (int(digest[0]), int(digest[3]))
    traces, params, compress = M.instance(prog, **kwargs)
    _check_all_tables(oracle, s, traces, params)
    # the constraints bite on the live rows: a wrong sum, a wrong fetched instruction
    bad = traces[0].copy()
    bad[T.COL_DST, 3] = (int(bad[T.COL_DST, 3]) + 1) % tracegen.P
    assert oracle.check_constraints(s.blob(), 0, bad) >= 0
    proof = oracle.prove_with_traces(s.blob(), traces, params, compress)
    rc, why = oracle.verify_all_proof(s.blob(), proof, params)
    assert rc == 0, why
    if program not in ("fibonacci", "mixed"):
        return      # one forged proof per kind of run is enough for the CPU suite's time budget
    # a CPU row that fetches a word the program table does not list breaks the cross-table product
    forged = [t.copy() for t in traces]
    forged[10][T.COL_PROG_EXEC_INST, 1] = (int(forged[10][T.COL_PROG_EXEC_INST, 1]) + 1) % tracegen.P
    try:
        p2 = oracle.prove_with_traces(s.blob(), forged, params, compress)
        rc, why = oracle.verify_all_proof(s.blob(), p2, params)
        assert rc != 0
    except RuntimeError:
        pass        # the prover itself may already refuse (quotient not divisible)


def test_wide_execution_satisfies_the_full_size_tables(oracle):
    """The same executor against the REAL fixed tables (2^16-entry range check, 2^18-row bitwise table): a program with
    32-bit operands whose lookups use every limb; all twelve AIRs vanish.  (Its proof is compared byte for byte with the
    GPU's in tests/test_gpu_stark.py -- the oracle prover needs minutes for these table sizes on a few cores.)"""
    from olavm_amd.air import miniexec as M
    s = T.ola_stark()
    rows, side, _ = M.execute(M.wide_program())
    assert len(side["bitwise"]) == 16 and len(side["cmp"]) == 11 and len(side["rc"]) == 16 and max(side["rc"]) > 1 << 31
    traces, params, _ = M.instance(M.wide_program(), range_bits=16, limb_bits=8)
    assert traces[2].shape[1] == 1 << 18 and traces[4].shape[1] == 1 << 16
    _check_all_tables(oracle, s, traces, params)
    bad = traces[4].copy()                      # a range-check limb that no longer matches its value
    bad[T.RC_LIMB_HI, 3] = (int(bad[T.RC_LIMB_HI, 3]) + 1) % tracegen.P
    assert oracle.check_constraints(s.blob(), 4, bad) >= 0
