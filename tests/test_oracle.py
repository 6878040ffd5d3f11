"""CPU tests that pin the oracle itself: the reference's known-answer vectors (Poseidon), the reference's own
property tests restated (NTT vs naive evaluation, coset round trip, divide_by_linear, Merkle prove/verify), and
protocol-constant self checks.  No GPU needed."""
import json
import os

import numpy as np
import pytest

from tests.oracle_lib import EDGE, P, rand_field

HERE = os.path.dirname(os.path.abspath(__file__))


def test_poseidon_known_answer_vectors(oracle):
    # plonky2/plonky2/src/hash/poseidon_goldilocks.rs:293-314
    kat = json.load(open(os.path.join(HERE, "golden", "poseidon_kat.json")))
    assert len(kat["vectors"]) == 4
    for v in kat["vectors"]:
        out = oracle.poseidon(np.array(v["input"], dtype=np.uint64))
        assert [int(x) for x in out] == v["output"]


def test_field_constants(oracle):
    # field_testing.rs:29-37: the 2^32-th root generator has exact order 2^32
    g = 1753635133440165772
    assert oracle.pow(g, 1 << 32) == 1 and oracle.pow(g, 1 << 31) == P - 1
    assert oracle.root_of_unity(32) == g and oracle.root_of_unity(0) == 1 and oracle.root_of_unity(1) == P - 1
    # goldilocks_extensions.rs:27: EXT_POWER_OF_TWO_GENERATOR = [0, c]; (cX)^2 = 7c^2 must be the base generator so that
    # the extension's roots of unity of order <= 2^32 are base-field elements (ext NTT = two base NTTs).
    c = 15659105665374529263
    assert (7 * c * c) % P == g
    # 2 has order 192 = 64*3 and 2^96 = -1: every 64th root of unity is a power of two (HIP butterflies rely on it)
    assert pow(2, 96, P) == P - 1 and pow(2, 192, P) == 1
    assert oracle.root_of_unity(6) in [pow(8, k, P) for k in range(1, 64, 2)]


def test_field_ops_against_python_ints(oracle):
    rng = np.random.default_rng(1)
    a = np.concatenate([np.repeat(EDGE, len(EDGE)), rand_field(rng, 500)])
    b = np.concatenate([np.tile(EDGE, len(EDGE)), rand_field(rng, 500)])
    A = [int(x) % P for x in a]
    B = [int(x) % P for x in b]
    assert [int(x) for x in oracle.vec_op("add", a, b)] == [(x + y) % P for x, y in zip(A, B)]
    assert [int(x) for x in oracle.vec_op("sub", a, b)] == [(x - y) % P for x, y in zip(A, B)]
    assert [int(x) for x in oracle.vec_op("mul", a, b)] == [(x * y) % P for x, y in zip(A, B)]
    inv = oracle.vec_op("inv", a)
    assert all((int(i) * x) % P == (1 if x else 0) for i, x in zip(inv, A))


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10])
def test_ntt_matches_naive_evaluation(oracle, log_n):
    # fft.rs:218-252 recipe: evaluate_poly == naive evaluation at w^i; interpolate_poly inverts it
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    c = rand_field(rng, n)
    ev = oracle.evaluate_poly(c)
    assert np.array_equal(ev, oracle.naive_eval(c, n, 1))
    assert np.array_equal(oracle.interpolate_poly(ev), c)


@pytest.mark.parametrize("log_n,blowup", [(0, 8), (1, 8), (4, 8), (7, 8), (6, 1), (5, 2), (9, 4)])
def test_coset_lde_matches_naive(oracle, log_n, blowup):
    # polynomial/mod.rs:494-538 recipe: coset FFT == naive evaluation on shift*<g>, coset iFFT round trip
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    c = rand_field(rng, n)
    lde = oracle.evaluate_poly_with_offset(c, 7, blowup)
    assert np.array_equal(lde, oracle.naive_eval(c, n * blowup, 7))
    if blowup == 1:
        assert np.array_equal(oracle.interpolate_poly_with_offset(lde, 7), c)


def test_ntt_edge_values(oracle):
    n = 64
    c = np.resize(EDGE, n)
    assert np.array_equal(oracle.evaluate_poly(c), oracle.naive_eval(c, n, 1))
    # non-canonical inputs (>= p) are accepted and reduced
    nc = np.full(n, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    assert np.array_equal(oracle.evaluate_poly(nc), oracle.evaluate_poly(nc - np.uint64(P)))


def test_sponge_overwrite_mode(oracle):
    # hashing.rs:84-107: a short last chunk overwrites only its own lanes; <= 4-element leaves are still permuted (F8)
    rng = np.random.default_rng(7)
    x = rand_field(rng, 11)
    st = np.zeros(12, dtype=np.uint64)
    st[:8] = x[:8]
    st = oracle.poseidon(st)
    st[:3] = x[8:]
    st = oracle.poseidon(st)
    assert np.array_equal(oracle.hash_no_pad(x), st[:4])
    one = np.zeros(12, dtype=np.uint64)
    one[0] = 5
    assert np.array_equal(oracle.hash_no_pad(np.array([5], dtype=np.uint64)), oracle.poseidon(one)[:4])
    # two_to_one = permute(l || r || 0)[0..4]
    l, r = rand_field(rng, 4), rand_field(rng, 4)
    st = np.zeros(12, dtype=np.uint64)
    st[:4], st[4:8] = l, r
    assert np.array_equal(oracle.two_to_one(l, r), oracle.poseidon(st)[:4])


@pytest.mark.parametrize("log_leaves,width,cap_h", [(4, 3, 4), (5, 9, 4), (7, 12, 4), (6, 5, 0), (6, 32, 2), (1, 4, 0)])
def test_merkle_prove_verify_all_leaves(oracle, log_leaves, width, cap_h):
    # merkle_tree/mod.rs:352-365 recipe, plus: heap-walk prove() == the reference's digest-layout formula (:273-308)
    rng = np.random.default_rng(log_leaves * 31 + width)
    leaves = rand_field(rng, (1 << log_leaves, width))
    assert oracle.merkle_selfcheck(leaves, cap_h) == 0
    cap, lh, nodes = oracle.merkle(leaves, cap_h, want_nodes=True)
    assert np.array_equal(lh[1], oracle.hash_no_pad(leaves[1]))
    if log_leaves > cap_h:
        assert np.array_equal(cap, nodes[1 << cap_h:2 << cap_h])
    else:
        assert np.array_equal(cap, lh)


def test_challenger_pops_from_the_back(oracle):
    # challenger.rs:86-100,134-153
    ch = oracle.challenger()
    ch.observe(np.arange(1, 4, dtype=np.uint64))
    st = np.zeros(12, dtype=np.uint64)
    st[:3] = [1, 2, 3]
    st = oracle.poseidon(st)
    assert [ch.get() for _ in range(8)] == [int(x) for x in st[:8][::-1]]
    st2 = oracle.poseidon(st)  # outputs exhausted -> duplex again with no new input
    assert ch.get() == int(st2[7])
    # 8 observed elements trigger an automatic duplexing
    ch2 = oracle.challenger()
    ch2.observe(np.arange(8, dtype=np.uint64))
    s = np.zeros(12, dtype=np.uint64)
    s[:8] = np.arange(8)
    assert np.array_equal(ch2.state(), oracle.poseidon(s))


def test_fri_reduction_arities(oracle):
    # reduction_strategies.rs:40-52 with ConstantArityBits(4,5), rate 3, cap 4 (SURVEY a-15)
    assert oracle.fri_arity_bits(12) == [4, 4]
    assert oracle.fri_arity_bits(20) == [4, 4, 4, 4]
    assert oracle.fri_arity_bits(22) == [4, 4, 4, 4, 4]
    assert oracle.fri_arity_bits(24) == [4, 4, 4, 4, 4]
    assert oracle.fri_arity_bits(5) == []
    assert oracle.fri_arity_bits(6) == [4]


def test_batch_commit_layout(oracle):
    # F9: leaf j of the commitment is the natural-order LDE row bitrev(j); cap entries 2c,2c+1 cover coset block c
    rng = np.random.default_rng(3)
    log_n, ncols = 5, 3
    vals = rand_field(rng, (ncols, 1 << log_n))
    b = oracle.batch(vals)
    co = b.coeffs()
    for c in range(ncols):
        assert np.array_equal(co[c], oracle.interpolate_poly(vals[c]))
    N = 8 << log_n
    leaves = b.leaves()
    rev = [int(format(j, "0%db" % (log_n + 3))[::-1], 2) for j in range(N)]
    for c in range(ncols):
        lde = oracle.evaluate_poly_with_offset(co[c], 7, 8)
        assert np.array_equal(leaves[:, c], lde[rev])
    assert np.array_equal(b.cap(), oracle.merkle(leaves, 4))


@pytest.mark.parametrize("log_n,cols,nperm", [(5, (3, 2, 2), 0), (7, (5, 4, 4), 1), (9, (4, 3, 2), 0)])
def test_open_and_prove_is_accepted_by_verifier(oracle, log_n, cols, nperm):
    # prover.rs:499-553 tail + fri/prover.rs against fri/verifier.rs (independent code paths)
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    tr = oracle.batch(rand_field(rng, (cols[0], n)))
    zs = oracle.batch(rand_field(rng, (cols[1], n)))
    qc = rand_field(rng, (cols[2], n))
    q = oracle.batch(qc, from_coeffs=True)
    ch = oracle.challenger()
    for b in (tr, zs, q):
        ch.observe(b.cap())
    vch = ch.clone()
    zeta, ob, fb = oracle.open_and_prove(tr, zs, q, nperm, ch)
    caps = np.stack([tr.cap(), zs.cap(), q.cap()])
    rc, why = oracle.verify_opening(caps, cols, log_n, nperm, ob + fb, vch)
    assert rc == 0, why
    # tamper with one byte of the final polynomial -> rejected
    bad = bytearray(ob + fb)
    bad[-20] ^= 1
    rc, why = oracle.verify_opening(caps, cols, log_n, nperm, bytes(bad), _replay(oracle, caps))
    assert rc != 0


def _replay(oracle, caps):
    ch = oracle.challenger()
    for c in caps:
        ch.observe(c)
    return ch


def test_batch_rehash_and_single_leaf(oracle):
    """The helpers the full-size GPU tests lean on: one leaf without copying all of them, and the same leaves under the other
    hash configuration == a batch built under that configuration from the start."""
    rng = np.random.default_rng(5)
    v = rand_field(rng, (7, 1 << 6))
    b = oracle.batch(v)
    leaves = b.leaves()
    for j in (0, 1, 63, 64, 300, 511):
        assert np.array_equal(b.leaf(j), leaves[j])
    cap_p = b.cap().copy()
    with oracle.hasher("blake3"):
        want = oracle.batch(v)
        b.rehash()
        assert np.array_equal(b.cap(), want.cap()) and not np.array_equal(b.cap(), cap_p)
        for j in (0, 77, 511):
            assert np.array_equal(b.prove(j), want.prove(j))
    b.rehash()
    assert np.array_equal(b.cap(), cap_p)
