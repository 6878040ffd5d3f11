"""GPU parity of the multi-table STARK prover (ola_prove_with_traces): AllProof bytes identical to the oracle's and
accepted by the oracle's restatement of the reference verifier."""
import numpy as np
import pytest

from olavm_amd.air import AirSet, ola_tables as T
from tests import tracegen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from olavm_amd.backend import Backend
    b = Backend(device=0)
    yield b
    b.close()


def mini_set(range_bits):
    return AirSet([T.cmp_table(), T.rangecheck_table(range_bits)], [T.ctl_cmp_rangecheck(0, 1)])


@pytest.mark.parametrize("range_bits,n_cmp", [(4, 6), (5, 30), (7, 100)])
def test_cmp_rangecheck_proof_bytes_match_oracle(be, oracle, range_bits, n_cmp):
    rng = np.random.default_rng(range_bits * 7 + n_cmp)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, n_cmp, range_bits)
    blob = mini_set(range_bits).blob()
    got = be.prove_with_traces(blob, [cmp_t, rc_t])
    want = oracle.prove_with_traces(blob, [cmp_t, rc_t])
    assert len(got) == len(want)
    assert got == want
    rc, why = oracle.verify_all_proof(blob, got)
    assert rc == 0, why


def test_real_size_rangecheck_u16(be, oracle):
    """The reference's real RangeCheck table (2^16-row fixed table, BASE = 2^16) with a 2^10-row Cmp table: too slow for
    the oracle prover in a test, so the oracle VERIFIER checks the GPU proof (constraints at zeta, CTL, FRI)."""
    rng = np.random.default_rng(16)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, 1000, 16)
    assert rc_t.shape == (12, 1 << 16) and cmp_t.shape == (6, 1 << 10)
    blob = mini_set(16).blob()
    proof = be.prove_with_traces(blob, [cmp_t, rc_t])
    rc, why = oracle.verify_all_proof(blob, proof)
    assert rc == 0, why


def test_invalid_trace_yields_a_rejected_proof(be, oracle):
    """Cmp has quotient_degree_factor 2 = 2^qdb, so trim_to_len(n*q) (prover.rs:469-473) is vacuous exactly as in the
    reference (SURVEY F10 discusses the same effect): the prover still emits bytes, identical to the oracle's, and
    the verifier rejects them at the quotient identity."""
    rng = np.random.default_rng(3)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, 6, 4)
    cmp_t[T.COL_CMP_GTE, 2] ^= 1
    blob = mini_set(4).blob()
    proof = be.prove_with_traces(blob, [cmp_t, rc_t])
    assert proof == oracle.prove_with_traces(blob, [cmp_t, rc_t])
    rc, why = oracle.verify_all_proof(blob, proof)
    assert rc != 0 and "quotient" in why
