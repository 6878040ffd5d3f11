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
