"""GPU parity tests: every call goes through the C ABI (include/ola_gpu.h) into the HIP kernels and is compared
bit-for-bit with the CPU oracle on the same seeded inputs.  Run on the MI355X box with `-m gpu`."""
import json
import os

import numpy as np
import pytest

from tests.oracle_lib import EDGE, P, rand_field

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def be():
    from olavm_amd.backend import Backend
    b = Backend(device=0)
    yield b
    b.close()


def bitrev_perm(bits):
    return np.array([int(format(j, "0%db" % bits)[::-1], 2) if bits else 0 for j in range(1 << bits)])


# ---------------------------------------------------------------- field arithmetic
def test_carry_flag_reduction_equals_the_plain_form(be):
    """gl.cuh's modular reduction for general products is inline assembly with explicit carry chains and hand-placed wait
    states; ola_gpu_selftest compares it on the device with the plain C++ reduction (45 k edge-value pairs + 2^31 random pairs
    biased towards all-ones / all-zero halves).  Everything downstream (NTT, Poseidon, quotient) is checked against the oracle
    anyway; this one names the culprit if a driver or compiler update breaks the hazard spacing."""
    assert be.selftest(1 << 31) == 0


# ---------------------------------------------------------------- Poseidon / sponge / Merkle
def test_poseidon_kats_on_device(be):
    kat = json.load(open(os.path.join(HERE, "golden", "poseidon_kat.json")))["vectors"]
    out = be.poseidon(np.array([v["input"] for v in kat], dtype=np.uint64))
    for o, v in zip(out, kat):
        assert [int(x) for x in o] == v["output"]


def test_poseidon_random_and_edge_states(be, oracle):
    rng = np.random.default_rng(11)
    st = np.concatenate([rand_field(rng, (500, 12)), np.resize(EDGE, (12, 12)),
                         rng.integers(0, 2**64, size=(50, 12), dtype=np.uint64)])  # last block: non-canonical inputs
    got = be.poseidon(st)
    for i in range(st.shape[0]):
        assert np.array_equal(got[i], oracle.poseidon(st[i])), i


def test_hash_rows_large_batch(be, oracle):
    rng = np.random.default_rng(5)
    rows = rand_field(rng, (9000, 11))              # above the quad threshold: one leaf per lane
    got = be.hash_rows(rows)
    for i in (0, 17, 4444, 8999):
        assert np.array_equal(got[i], oracle.hash_no_pad(rows[i])), i
    assert np.array_equal(got[:100], be.hash_rows(rows[:100]))      # quad form on the same rows


def test_poseidon_both_device_forms_agree(be, oracle):
    """Batches above 8192 states run one state per lane, smaller ones the quad-cooperative form (4 lanes per state): same
    permutation.  The large batch is checked against the oracle on a sample and against the small-batch form everywhere."""
    rng = np.random.default_rng(77)
    st = rand_field(rng, (9001, 12))
    big = be.poseidon(st)
    for i in (0, 1, 4500, 9000):
        assert np.array_equal(big[i], oracle.poseidon(st[i])), i
    small = np.concatenate([be.poseidon(st[:5000]), be.poseidon(st[5000:])])
    assert np.array_equal(big, small)


@pytest.mark.parametrize("row_len", [1, 3, 4, 7, 8, 9, 16, 17, 29, 94, 134])
def test_hash_rows(be, oracle, row_len):
    rng = np.random.default_rng(row_len)
    rows = rand_field(rng, (300, row_len))
    got = be.hash_rows(rows)
    for i in range(0, 300, 7):
        assert np.array_equal(got[i], oracle.hash_no_pad(rows[i]))


@pytest.mark.parametrize("log_leaves,width,cap_h", [(4, 5, 4), (5, 3, 4), (9, 12, 4), (8, 33, 0), (6, 2, 2)])
def test_merkle_cap(be, oracle, log_leaves, width, cap_h):
    rng = np.random.default_rng(log_leaves + width)
    leaves = rand_field(rng, (1 << log_leaves, width))
    assert np.array_equal(be.merkle_cap(leaves, cap_h), oracle.merkle(leaves, cap_h))


def test_pow_minimal_witness(be, oracle):
    rng = np.random.default_rng(2)
    for bits in (4, 10, 16):
        h = rand_field(rng, 4)
        assert be.pow(h, bits) == oracle.fri_pow(h, bits)


# ---------------------------------------------------------------- NTT family
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 7, 8, 10, 12, 13, 14, 15, 17, 18])
def test_ntt_evaluate_interpolate(be, oracle, log_n):
    from olavm_amd.backend import OLA_NTT_EVALUATE, OLA_NTT_INTERPOLATE
    rng = np.random.default_rng(log_n)
    batch = 3 if log_n < 16 else 2
    c = rand_field(rng, (batch, 1 << log_n))
    ev = be.ntt(OLA_NTT_EVALUATE, c)
    for b in range(batch):
        assert np.array_equal(ev[b], oracle.evaluate_poly(c[b])), (log_n, b)
    back = be.ntt(OLA_NTT_INTERPOLATE, ev)
    assert np.array_equal(back, c)


@pytest.mark.parametrize("log_n", [0, 1, 3, 6, 9, 12, 13, 14, 16])
def test_coset_lde_natural_and_leaf_order(be, oracle, log_n):
    from olavm_amd.backend import OLA_NTT_COSET_LDE, OLA_NTT_COSET_LDE_LEAF_ORDER
    rng = np.random.default_rng(100 + log_n)
    c = rand_field(rng, (2, 1 << log_n))
    nat = be.ntt(OLA_NTT_COSET_LDE, c, shift=7, blowup_log=3)
    leaf = be.ntt(OLA_NTT_COSET_LDE_LEAF_ORDER, c, shift=7, blowup_log=3)
    rev = bitrev_perm(log_n + 3)
    for b in range(2):
        want = oracle.evaluate_poly_with_offset(c[b], 7, 8)
        assert np.array_equal(nat[b], want)
        assert np.array_equal(leaf[b], want[rev])


@pytest.mark.parametrize("log_n,shift", [(4, 7), (10, 7), (14, 7), (11, 49), (15, 3)])
def test_coset_fft_and_ifft_blowup1(be, oracle, log_n, shift):
    from olavm_amd.backend import OLA_NTT_COSET_LDE, OLA_NTT_COSET_INTERPOLATE
    rng = np.random.default_rng(log_n)
    c = rand_field(rng, (2, 1 << log_n))
    ev = be.ntt(OLA_NTT_COSET_LDE, c, shift=shift, blowup_log=0)
    for b in range(2):
        assert np.array_equal(ev[b], oracle.evaluate_poly_with_offset(c[b], shift, 1))
    back = be.ntt(OLA_NTT_COSET_INTERPOLATE, ev, shift=shift)
    assert np.array_equal(back, c)
    assert np.array_equal(back[0], oracle.interpolate_poly_with_offset(ev[0], shift))


def test_ntt_edge_and_noncanonical_inputs(be, oracle):
    from olavm_amd.backend import OLA_NTT_EVALUATE
    n = 1 << 9
    c = np.stack([np.resize(EDGE, n), np.full(n, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64), np.zeros(n, dtype=np.uint64)])
    ev = be.ntt(OLA_NTT_EVALUATE, c)
    for b in range(3):
        assert np.array_equal(ev[b], oracle.evaluate_poly(c[b]))
    assert ev.max() < P


@pytest.mark.parametrize("log_n,batch", [(20, 2), (22, 1)])
def test_ntt_large_roundtrip_and_spot_values(be, oracle, log_n, batch):
    """Full-size properties (the oracle is too slow for a full compare at 2^22): iNTT(NTT(x)) == x, and a few output
    points equal the naive evaluation of the polynomial."""
    from olavm_amd.backend import OLA_NTT_EVALUATE, OLA_NTT_INTERPOLATE
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    c = rand_field(rng, (batch, n))
    ev = be.ntt(OLA_NTT_EVALUATE, c)
    assert np.array_equal(be.ntt(OLA_NTT_INTERPOLATE, ev), c)
    if log_n <= 20:
        assert np.array_equal(ev[0], oracle.evaluate_poly(c[0]))
    # linearity: NTT(a) + NTT(b) == NTT(a + b)
    a, b2 = c[0], rand_field(rng, n)
    s = oracle.vec_op("add", a, b2)
    e2 = be.ntt(OLA_NTT_EVALUATE, np.stack([b2, s]))
    assert np.array_equal(oracle.vec_op("add", ev[0], e2[0]), e2[1])


# ---------------------------------------------------------------- PolynomialBatch commitment
@pytest.mark.parametrize("log_n,ncols", [(1, 1), (3, 2), (5, 3), (8, 12), (10, 9), (12, 29), (14, 6), (16, 3)])
def test_commit_values_matches_oracle(be, oracle, log_n, ncols):
    rng = np.random.default_rng(log_n * 100 + ncols)
    vals = rand_field(rng, (ncols, 1 << log_n))
    b = be.commit(vals)
    ob = oracle.batch(vals)
    assert np.array_equal(b.cap(), ob.cap())
    assert np.array_equal(b.coeffs(), ob.coeffs())
    N = 8 << log_n
    leaves = ob.leaves()
    for j in sorted(set([0, 1, N - 1, N // 2, int(rng.integers(0, N)), int(rng.integers(0, N))])):
        row, sib = b.leaf(j)
        assert np.array_equal(row, leaves[j])
        assert np.array_equal(sib, ob.prove(j))
    # get_lde_values(index, step) -- natural-order row index*step (fri/oracle.rs:131-137)
    lde0 = oracle.evaluate_poly_with_offset(ob.coeffs()[0], 7, 8)
    for idx, step in [(0, 1), (3, 2), (1, 8), ((1 << log_n) - 1, 8)]:
        if idx * step < N:
            assert int(b.lde_row(idx, step)[0]) == int(lde0[idx * step])
    b.free()


def test_commit_coeffs_matches_oracle(be, oracle):
    rng = np.random.default_rng(77)
    co = rand_field(rng, (4, 1 << 9))
    b = be.commit(co, from_coeffs=True)
    ob = oracle.batch(co, from_coeffs=True)
    assert np.array_equal(b.cap(), ob.cap())
    b.free()


def test_commit_from_device_buffer(be, oracle):
    import torch
    rng = np.random.default_rng(5)
    vals = rand_field(rng, (7, 1 << 11))
    t = torch.from_numpy(vals.view(np.int64)).cuda()
    b = be.commit_dev(t.data_ptr(), 7, 11)
    assert np.array_equal(b.cap(), oracle.batch(vals).cap())
    b.free()


# ---------------------------------------------------------------- openings + FRI (prover.rs:499-553 tail)
@pytest.mark.parametrize("log_n,cols,nperm", [(5, (3, 2, 2), 0), (7, (5, 4, 4), 1), (9, (12, 6, 2), 2), (12, (9, 5, 4), 0),
                                              (14, (6, 3, 2), 0)])
def test_open_and_prove_bytes_match_oracle_and_verify(be, oracle, log_n, cols, nperm):
    from olavm_amd.backend import Challenger
    rng = np.random.default_rng(1000 + log_n)
    n = 1 << log_n
    tv, zv, qc = rand_field(rng, (cols[0], n)), rand_field(rng, (cols[1], n)), rand_field(rng, (cols[2], n))
    # GPU path
    gt, gz, gq = be.commit(tv), be.commit(zv), be.commit(qc, from_coeffs=True)
    ch = Challenger()
    for b in (gt, gz, gq):
        ch.observe(b.cap())
    g_open, g_fri = be.open_and_prove(gt, gz, gq, nperm, ch)
    # oracle path
    ot, oz, oq = oracle.batch(tv), oracle.batch(zv), oracle.batch(qc, from_coeffs=True)
    och = oracle.challenger()
    for b in (ot, oz, oq):
        och.observe(b.cap())
    vch = och.clone()
    zeta, o_open, o_fri = oracle.open_and_prove(ot, oz, oq, nperm, och)
    assert g_open == o_open, "opening set bytes differ"
    assert g_fri == o_fri, "FRI proof bytes differ"
    # the transcript ends in the same state
    assert ch.get() == och.get()
    # and the independently written verifier accepts the GPU proof
    caps = np.stack([gt.cap(), gz.cap(), gq.cap()])
    rc, why = oracle.verify_opening(caps, cols, log_n, nperm, g_open + g_fri, vch)
    assert rc == 0, why
    for b in (gt, gz, gq):
        b.free()


def test_open_and_prove_large_verifies(be, oracle):
    """2^18 rows: too slow for the oracle prover, but the oracle VERIFIER checks the GPU proof end to end
    (Merkle paths, FRI folding consistency, final polynomial, proof of work)."""
    from olavm_amd.backend import Challenger
    rng = np.random.default_rng(18)
    log_n, cols, nperm = 18, (8, 4, 4), 0
    n = 1 << log_n
    gt = be.commit(rand_field(rng, (cols[0], n)))
    gz = be.commit(rand_field(rng, (cols[1], n)))
    gq = be.commit(rand_field(rng, (cols[2], n)), from_coeffs=True)
    ch = Challenger()
    och = oracle.challenger()
    for b in (gt, gz, gq):
        ch.observe(b.cap())
        och.observe(b.cap())
    g_open, g_fri = be.open_and_prove(gt, gz, gq, nperm, ch)
    caps = np.stack([gt.cap(), gz.cap(), gq.cap()])
    rc, why = oracle.verify_opening(caps, cols, log_n, nperm, g_open + g_fri, och)
    assert rc == 0, why


# ------------------------------------------------------------------------------------------------ coset-sharded commitment
@pytest.mark.parametrize("log_n,ncols", [(6, 5), (10, 9), (15, 3)])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_coset_sharded_commit_reassembles_the_full_commitment(be, log_n, ncols, world):
    """every rank commits its own leaf blocks (cosets); the rank-ordered concatenation of the cap slices is the
    cap of the unsharded commitment, and every leaf row / Merkle path agrees with the unsharded tree (all ranks are run
    one after the other on the single test GPU)."""
    rng = np.random.default_rng(log_n * 10 + world)
    vals = rand_field(rng, (ncols, 1 << log_n))
    full = be.commit(vals)
    N = 8 << log_n
    slices = []
    for rank in range(world):
        sh = be.commit_shard(vals, rank, world)
        assert sh.rate_bits == 3 - (world.bit_length() - 1)
        slices.append(sh.cap())
        assert np.array_equal(sh.coeffs(), full.coeffs())
        for local in (0, 1, N // world - 1, int(rng.integers(0, N // world))):
            row, sib = sh.leaf(local)
            frow, fsib = full.leaf(rank * (N // world) + local)
            assert np.array_equal(row, frow) and np.array_equal(sib, fsib)
        sh.free()
    assert np.array_equal(np.concatenate(slices), full.cap())
    full.free()


def test_shard_batches_are_refused_where_a_full_commitment_is_needed(be):
    from olavm_amd.backend import OlaGpuError
    vals = rand_field(np.random.default_rng(1), (2, 64))
    sh = be.commit_shard(vals, 1, 2)
    with pytest.raises(OlaGpuError, match="complete commitment"):
        sh.lde_row(3)
    sh.free()
    with pytest.raises(OlaGpuError):
        be.commit_shard(vals, 2, 2)
    with pytest.raises(OlaGpuError):
        be.commit_shard(vals, 0, 3)


# ---------------------------------------------------------------- three-pass sizes (2^19 and up) against the oracle
@pytest.mark.parametrize("log_n", [19, 20, 21])
def test_ntt_three_pass_sizes_match_oracle(be, oracle, log_n):
    """Every pass split the planner uses between 2^19 and 2^21 (three launches), full compare with the oracle: evaluate /
    interpolate in natural order, the x8 coset LDE in leaf order (cosets 0, 3 and 7 compared), and non-canonical input words."""
    from olavm_amd.backend import OLA_NTT_COSET_LDE_LEAF_ORDER, OLA_NTT_EVALUATE, OLA_NTT_INTERPOLATE
    rng = np.random.default_rng(3000 + log_n)
    n = 1 << log_n
    c = rand_field(rng, (3, n))
    c[2] = np.resize(EDGE, n)
    c[2, ::7] = np.uint64(0xFFFFFFFFFFFFFFFF)
    ev = be.ntt(OLA_NTT_EVALUATE, c)
    back = be.ntt(OLA_NTT_INTERPOLATE, ev)
    for b in range(3):
        assert np.array_equal(ev[b], oracle.evaluate_poly(c[b])), b
        assert np.array_equal(back[b], oracle.vec_op("add", c[b], np.zeros(n, dtype=np.uint64))), b      # canonical form of the input
    assert np.array_equal(back[0], oracle.interpolate_poly(ev[0]))
    leaf = be.ntt(OLA_NTT_COSET_LDE_LEAF_ORDER, c[:2], shift=7, blowup_log=3)
    rev = bitrev_perm(log_n + 3)
    want = oracle.evaluate_poly_with_offset(c[1], 7, 8)[rev]
    for k in (0, 3, 7):
        assert np.array_equal(leaf[1][k * n:(k + 1) * n], want[k * n:(k + 1) * n]), k
