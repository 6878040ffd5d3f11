"""GPU parity under the reference's Blake3GoldilocksConfig (plonky2/plonky2/src/plonk/config.rs:153-161 -- the configuration of its
own full-prove tests, circuits/src/stark/ola_stark.rs:684, and of every published number): Merkle trees and the challenger on
BLAKE3 (hash/blake3.rs:166-233), proof of work on Poseidon.  Every call goes through the C ABI of a context created with
hasher = OLA_HASH_BLAKE3; the checker is the oracle switched to the same configuration (oracle/blake3.cpp, pinned by the official
BLAKE3 vectors in tests/test_blake3.py)."""
import time

import numpy as np
import pytest

from olavm_amd.air import ola_tables as T
from tests.oracle_lib import P, rand_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from olavm_amd.backend import Backend
    b = Backend(device=0, hasher="blake3")
    yield b
    b.close()


@pytest.fixture()
def b3(oracle):
    with oracle.hasher("blake3"):
        yield oracle


@pytest.mark.parametrize("row_len", [1, 3, 4, 5, 8, 9, 16, 29, 94, 127, 128, 129, 134, 256, 257, 300, 513])
def test_hash_rows(be, b3, row_len):
    """Blake3_256::hash_no_pad of every row: partial blocks, exactly one chunk, chunk + 1 word (the tree mode), three and five
    chunks; non-canonical words hash as their canonical value; small batches and batches past the 8192 switch of the launchers."""
    rng = np.random.default_rng(row_len)
    for nrows in (300, 9000):
        rows = rand_field(rng, (nrows, row_len))
        rows[0, 0] = np.uint64(P + 1)
        rows[1, row_len - 1] = np.uint64(2**64 - 1)
        got = be.hash_rows(rows)
        for i in list(range(0, 300, 13)) + [nrows - 1]:
            assert np.array_equal(got[i], b3.merkle_hash_leaf(rows[i])), (nrows, i)


@pytest.mark.parametrize("log_leaves,width,cap_h", [(4, 5, 4), (5, 3, 4), (9, 12, 4), (8, 33, 0), (6, 2, 2), (13, 7, 4), (10, 134, 3)])
def test_merkle_cap(be, b3, log_leaves, width, cap_h):
    rng = np.random.default_rng(log_leaves + width)
    leaves = rand_field(rng, (1 << log_leaves, width))
    assert np.array_equal(be.merkle_cap(leaves, cap_h), b3.merkle(leaves, cap_h))


@pytest.mark.parametrize("log_n,ncols", [(1, 1), (3, 2), (5, 3), (8, 12), (10, 9), (12, 29), (14, 6), (9, 134)])
def test_commit_values_matches_oracle(be, b3, log_n, ncols):
    """PolynomialBatch::from_values under the configuration: cap, coefficients, leaves and sibling paths."""
    rng = np.random.default_rng(log_n * 100 + ncols)
    vals = rand_field(rng, (ncols, 1 << log_n))
    b = be.commit(vals)
    ob = b3.batch(vals)
    assert np.array_equal(b.cap(), ob.cap())
    assert np.array_equal(b.coeffs(), ob.coeffs())
    N = 8 << log_n
    leaves = ob.leaves()
    for j in sorted(set([0, 1, N - 1, N // 2, int(rng.integers(0, N)), int(rng.integers(0, N))])):
        row, sib = b.leaf(j)
        assert np.array_equal(row, leaves[j])
        assert np.array_equal(sib, ob.prove(j))
    b.free()


@pytest.mark.parametrize("log_n,cols,nperm", [(5, (3, 2, 2), 0), (7, (5, 4, 4), 1), (9, (12, 6, 2), 2), (12, (9, 5, 4), 0)])
def test_open_and_prove_bytes_match_oracle_and_verify(be, b3, log_n, cols, nperm):
    """Openings + FRI with the BLAKE3 challenger (the onion permutation, digests observed as 5 elements) and BLAKE3 layer trees;
    the proof-of-work witness still comes from the Poseidon kernel (InnerHasher)."""
    from olavm_amd.backend import Challenger
    rng = np.random.default_rng(1000 + log_n)
    n = 1 << log_n
    tv, zv, qc = rand_field(rng, (cols[0], n)), rand_field(rng, (cols[1], n)), rand_field(rng, (cols[2], n))
    gt, gz, gq = be.commit(tv), be.commit(zv), be.commit(qc, from_coeffs=True)
    ch = Challenger(hasher="blake3")
    ot, oz, oq = b3.batch(tv), b3.batch(zv), b3.batch(qc, from_coeffs=True)
    och = b3.challenger()
    for b, ob in ((gt, ot), (gz, oz), (gq, oq)):
        ch.observe_cap(b.cap())
        och.observe_cap(ob.cap())
    vch = och.clone()
    g_open, g_fri = be.open_and_prove(gt, gz, gq, nperm, ch)
    zeta, o_open, o_fri = b3.open_and_prove(ot, oz, oq, nperm, och)
    assert g_open == o_open, "opening set bytes differ"
    assert g_fri == o_fri, "FRI proof bytes differ"
    assert ch.get() == och.get()
    caps = np.stack([gt.cap(), gz.cap(), gq.cap()])
    rc, why = b3.verify_opening(caps, cols, log_n, nperm, g_open + g_fri, vch)
    assert rc == 0, why
    # a Poseidon challenger is refused by a BLAKE3 context instead of producing a proof no verifier accepts
    from olavm_amd.backend import OlaGpuError
    with pytest.raises(OlaGpuError):
        be.open_and_prove(gt, gz, gq, nperm, Challenger())
    for b in (gt, gz, gq):
        b.free()


def test_twelve_table_all_proof_bytes_match_oracle(be, b3):
    """All 12 tables and 19 lookups: AllProof bytes identical to the oracle's under the same configuration (tables on the
    interpreter kernel and on the generated kernels), and the verifier accepts."""
    from tests import tracegen
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    for log_n in (3, 8):
        traces, params, compress = tracegen.empty_program_instance(log_n=log_n, live=np.random.default_rng(log_n))
        got = be.prove_with_traces(blob, traces, params, compress)
        want = b3.prove_with_traces(blob, traces, params, compress)
        assert got == want
        rc, why = b3.verify_all_proof(blob, got, params)
        assert rc == 0, why


@pytest.mark.parametrize("program", ["mixed", "hash"])
def test_real_execution_proof_bytes_match_oracle(be, oracle, program):
    """Executed programs (the Poseidon builtin and storage fill the 134-column Poseidon table: two-chunk leaves).  The traces are
    built before the oracle is switched: trace generation derives its compress challenges with a Poseidon challenger in both
    configurations (generation/builtin.rs:121, generation/prog.rs:24)."""
    from olavm_amd.air import miniexec as M
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    factory, kwargs = M.EXAMPLES[program]
    traces, params, compress = M.instance(factory(), **kwargs)
    got = be.prove_with_traces(blob, traces, params, compress)
    with oracle.hasher("blake3"):
        assert got == oracle.prove_with_traces(blob, traces, params, compress)
        rc, why = oracle.verify_all_proof(blob, got, params)
        assert rc == 0, why
    # the Poseidon configuration's verifier does not take it
    assert oracle.verify_all_proof(blob, got, params)[0] != 0


def test_memory_lean_proof_is_the_same_proof(be, b3, monkeypatch):
    from olavm_amd.air import miniexec as M
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = M.instance(M.mixed_program())
    monkeypatch.setenv("OLA_LEAN", "0")
    resident = be.prove_with_traces(blob, traces, params, compress)
    monkeypatch.setenv("OLA_LEAN", "1")
    assert be.prove_with_traces(blob, traces, params, compress) == resident


def test_full_prove_of_a_2p20_row_execution(be, oracle):
    """The reference's README workload shape (a program with a 2^20-row CPU table, Blake3GoldilocksConfig): executed program,
    full-size fixed tables, verifier of the same configuration accepts; timing printed for the record."""
    from olavm_amd.air import fastexec, miniexec as M
    blob = T.ola_stark().blob()
    traces, params, compress = fastexec.instance(M.memory_program(70000), range_bits=16, limb_bits=8, max_steps=1 << 24)
    assert traces[0].shape == (94, 1 << 20)
    be.prove_with_traces(blob, traces, params, compress)           # first call of this shape: allocations
    t0 = time.perf_counter()
    proof = be.prove_with_traces(blob, traces, params, compress)
    dt = time.perf_counter() - t0
    with oracle.hasher("blake3"):
        rc, why = oracle.verify_all_proof(blob, proof, params)
    assert rc == 0, why
    print("Blake3GoldilocksConfig, executed 2^20-row program: %d proof bytes, prove_with_traces %.3f s" % (len(proof), dt))
    be.trim()
