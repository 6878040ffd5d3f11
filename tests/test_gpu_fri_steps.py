"""The opening proof one step per call (ABI revision 7: ola_open, ola_fri_plan, ola_fri_commit_begin / next_layer / finish, ola_pow,
ola_fri_query -- SURVEY 8(b)'s layer-stepped exports) with the transcript kept by the CALLER, as a host would that keeps the reference's own
`prove_openings` / `fri_committed_trees` / `fri_proof_of_work` / `fri_prover_query_rounds` loops (fri/oracle.rs:167-241,
fri/prover.rs:20-204): the pieces reassemble to exactly the bytes of ola_open_and_prove, and the two transcripts end in the same state."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def rand_field(rng, shape):
    return rng.integers(0, P, size=shape, dtype=np.uint64)


def ext_vec(buf, at):
    (k,) = struct.unpack_from("<I", buf, at)
    v = np.frombuffer(buf, dtype="<u8", count=2 * k, offset=at + 4).reshape(k, 2)
    return v, at + 4 + 16 * k


def field_vec(buf, at):
    (k,) = struct.unpack_from("<I", buf, at)
    return np.frombuffer(buf, dtype="<u8", count=k, offset=at + 4), at + 4 + 8 * k


@pytest.mark.parametrize("hasher", ["poseidon", "blake3"])
@pytest.mark.parametrize("log_n,cols,nperm", [(3, (3, 2, 2), 0), (7, (5, 4, 4), 1), (12, (9, 5, 4), 2), (16, (7, 4, 2), 0), (18, (6, 3, 2), 1)])
def test_steps_with_the_callers_transcript_reassemble_the_fused_proof(hasher, log_n, cols, nperm):
    from olavm_amd.backend import Backend, Challenger
    be = Backend(device=0, hasher=hasher)
    rng = np.random.default_rng(7000 + log_n)
    n = 1 << log_n
    gt, gz, gq = be.commit(rand_field(rng, (cols[0], n))), be.commit(rand_field(rng, (cols[1], n))), be.commit(rand_field(rng, (cols[2], n)), from_coeffs=True)
    ch = Challenger(hasher=hasher)
    for b in (gt, gz, gq):
        ch.observe_cap(b.cap())
    mine = ch.clone()
    want_open, want_fri = be.open_and_prove(gt, gz, gq, nperm, ch)

    # ---- the caller's side of prove_single_table's tail (prover.rs:499-553) and of fri_proof
    zeta = mine.get(2)
    got_open, fri = be.open(gt, gz, gq, nperm, zeta)
    assert got_open == want_open
    local, at = ext_vec(got_open, 0)
    nxt, at = ext_vec(got_open, at)
    zs_local, at = ext_vec(got_open, at)
    zs_next, at = ext_vec(got_open, at)
    ctl_last, at = field_vec(got_open, at)
    q_local, at = ext_vec(got_open, at)
    assert at == len(got_open)
    for v in (local, zs_local, q_local, nxt, zs_next):                 # observe_openings in to_fri_openings order (proof.rs:235-265)
        mine.observe(v)
    mine.observe(np.stack([ctl_last, np.zeros_like(ctl_last)], axis=1))
    fri.begin(mine.get(2))                                              # alpha
    caps, beta = [], None
    for _ in fri.arity_bits:                                            # fri_committed_trees (fri/prover.rs:72-121)
        cap = fri.next_layer(beta)
        caps.append(cap)
        mine.observe_cap(cap)
        beta = mine.get(2)
    final_poly = fri.finish(beta)
    assert final_poly.shape == (fri.final_poly_len, 2)
    mine.observe(final_poly)
    witness = be.pow(mine.get(4), bits=16)      # fri_proof_of_work (prover.rs:126-148)
    nq = 28
    N = n << 3
    xs = np.array([int(mine.get()) % N for _ in range(nq)], dtype=np.uint64)               # fri_prover_query_rounds (prover.rs:150-204)
    queries = fri.query(xs)
    fri.free()

    # ---- FriProof in wire format (serialization.rs:305-317): caps, query rounds, final polynomial, proof-of-work witness
    out = struct.pack("<I", len(caps))
    for cap in caps:
        out += struct.pack("<I", cap.shape[0]) + cap.astype("<u8").tobytes()
    out += queries
    out += struct.pack("<I", final_poly.shape[0]) + final_poly.astype("<u8").tobytes()
    out += struct.pack("<Q", int(witness))
    assert out == want_fri
    assert mine.get() == ch.get()                                       # the two transcripts are in the same state
    for b in (gt, gz, gq):
        b.free()
    be.close()


def test_steps_refuse_calls_out_of_order():
    from olavm_amd.backend import Backend, Challenger, OlaGpuError
    be = Backend(device=0)
    rng = np.random.default_rng(5)
    n = 1 << 10
    gt, gz, gq = be.commit(rand_field(rng, (4, n))), be.commit(rand_field(rng, (3, n))), be.commit(rand_field(rng, (2, n)), from_coeffs=True)
    _, fri = be.open(gt, gz, gq, 0, [3, 5])
    with pytest.raises(OlaGpuError):
        fri.next_layer(None)                       # before begin
    fri.begin([7, 11])
    with pytest.raises(OlaGpuError):
        fri.begin([7, 11])                         # twice
    with pytest.raises(OlaGpuError):
        fri.next_layer([1, 2])                     # the first layer takes no beta
    fri.next_layer(None)
    with pytest.raises(OlaGpuError):
        fri.query(np.zeros(1, dtype=np.uint64))    # before finish
    if len(fri.arity_bits) > 1:
        with pytest.raises(OlaGpuError):
            fri.finish([1, 2])                     # layers missing
    fri.free()
    with pytest.raises(OlaGpuError):
        be.open(gt, gz, gq, 9, [3, 5])             # more permutation Zs than Z columns
    for b in (gt, gz, gq):
        b.free()
    be.close()
