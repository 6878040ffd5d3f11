"""Blake3GoldilocksConfig (plonky2/plonky2/src/plonk/config.rs:153-161), CPU side: the oracle's BLAKE3 against vectors of the
official implementation, the hasher as Merkle trees and the challenger see it (hash/blake3.rs:166-233,
hash/hash_types.rs:142-152), the product's host code (the same header the kernels are compiled from) against the oracle, and
an oracle prove -> verify round trip under the configuration."""
import ctypes as C
import json
import os

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
HERE = os.path.dirname(os.path.abspath(__file__))
LLVM_LIB = "/opt/rocm/lib/llvm/lib/libclang-cpp.so"     # part of the ROCm image (here and on the GPU box), not of the reference


def _pattern(n):
    return bytes(i % 251 for i in range(n))


def test_oracle_blake3_matches_the_official_vectors(oracle):
    """tests/golden/blake3_vectors.json (made by tests/golden/make_blake3_vectors.py with LLVM's copy of the official C code):
    the official test-vector lengths -- block, chunk and tree boundaries up to 100 chunks -- and the lengths this backend hashes."""
    d = json.load(open(os.path.join(HERE, "golden", "blake3_vectors.json")))
    assert len(d["vectors"]) >= 45
    for v in d["vectors"]:
        assert oracle.blake3(_pattern(v["len"])).hex() == v["hash"], v["len"]
    for t in d["text"]:
        assert oracle.blake3(t["ascii"].encode()).hex() == t["hash"]
    # the two digests everybody knows
    assert oracle.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert oracle.blake3(b"hello world").hex() == "d74981efa70a0c880b8d8c1985d075dbcbf679b99a5f9914e5aaf96b831a9e24"


@pytest.mark.skipif(not os.path.exists(LLVM_LIB), reason="LLVM's BLAKE3 is not on this machine")
def test_oracle_blake3_matches_llvm_blake3_on_random_inputs(oracle):
    from tests.golden.make_blake3_vectors import hasher
    h, _ = hasher()
    rng = np.random.default_rng(3)
    for n in list(range(0, 200)) + [int(x) for x in rng.integers(200, 20000, 60)]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.blake3(data) == h(data), n


def _py_onion(oracle, state):
    """Blake3Permutation::permute (hash/blake3.rs:166-201) spelled out."""
    cur = b"".join(int(x % P).to_bytes(8, "little") for x in state)
    out = []
    while len(out) < 12:
        cur = oracle.blake3(cur)
        for i in range(4):
            w = int.from_bytes(cur[8 * i:8 * i + 8], "little")
            if w < P and len(out) < 12:
                out.append(w)
    return out


def test_hasher_as_the_trees_and_the_challenger_see_it(oracle):
    rng = np.random.default_rng(5)
    with oracle.hasher("blake3"):
        # hash_no_pad: canonical little-endian words (also for <= 4 elements: new_v2 never takes the no-op, merkle_tree/mod.rs:180-201)
        for n in (1, 3, 4, 5, 16, 94, 134, 300):
            x = rng.integers(0, 2**64, n, dtype=np.uint64)
            x[0] = P + 5                      # a non-canonical word hashes as its canonical value
            want = oracle.blake3(b"".join(int(v % P).to_bytes(8, "little") for v in x.tolist()))
            assert oracle.merkle_hash_leaf(x).tobytes() == want
        # two_to_one: the 64 bytes of two digests, which are bytes and never reduced
        l = np.array([2**64 - 1, P, P + 1, 7], dtype=np.uint64)
        r = rng.integers(0, 2**64, 4, dtype=np.uint64)
        assert oracle.merkle_two_to_one(l, r).tobytes() == oracle.blake3(l.tobytes() + r.tobytes())
        # the onion
        for _ in range(5):
            s = rng.integers(0, 2**64, 12, dtype=np.uint64)
            assert oracle.blake3_permutation(s).tolist() == _py_onion(oracle, s.tolist())
        # a challenger observes a digest as 5 elements of 7 bytes
        d = rng.integers(0, 2**64, (2, 4), dtype=np.uint64)
        a, b = oracle.challenger(), oracle.challenger()
        a.observe_cap(d)
        for row in d:
            raw = row.tobytes()
            b.observe([int.from_bytes(raw[7 * c:7 * c + 7], "little") for c in range(5)])
        assert [a.get() for _ in range(4)] == [b.get() for _ in range(4)]
    # outside the block the oracle is back on Poseidon
    x = np.arange(9, dtype=np.uint64)
    assert np.array_equal(oracle.merkle_hash_leaf(x), oracle.hash_no_pad(x))


def test_product_host_code_matches_the_oracle(oracle):
    """olavm_amd/csrc/blake3.cuh is compiled for host and device from the same source: its host instantiation (digest of field
    elements, the challenger's permutation and digest observation) against the oracle, single- and multi-chunk leaves."""
    from olavm_amd.backend import Challenger, _p, load_library
    L = load_library()
    rng = np.random.default_rng(7)
    with oracle.hasher("blake3"):
        for n in [1, 2, 4, 5, 7, 8, 9, 16, 53, 94, 127, 128, 129, 134, 255, 256, 257, 300, 384, 385, 512, 513, 1000, 2049, 4096]:
            x = rng.integers(0, 2**64, n, dtype=np.uint64)
            out = np.empty(4, dtype=np.uint64)
            assert L.ola_blake3_hash_elements(_p(x), n, _p(out)) == 0
            assert np.array_equal(out, oracle.merkle_hash_leaf(x)), n
        assert L.ola_blake3_hash_elements(_p(x), 0, _p(out)) != 0 and L.ola_blake3_hash_elements(_p(x), 4097, _p(out)) != 0
        ch, och = Challenger(L, "blake3"), oracle.challenger()
        for rnd in range(30):
            e = rng.integers(0, 2**64, int(rng.integers(0, 20)), dtype=np.uint64)
            ch.observe(e); och.observe(e)
            d = rng.integers(0, 2**64, (int(rng.integers(0, 4)), 4), dtype=np.uint64)
            ch.observe_cap(d); och.observe_cap(d)
            if rnd % 7 == 3:
                ch.compact(); och.compact()
            k = int(rng.integers(1, 12))
            assert [ch.get() for _ in range(k)] == [och.get() for _ in range(k)], rnd
        assert np.array_equal(ch.state(), och.state())
    # the default challenger is the Poseidon one, and a clone keeps its hasher
    p = Challenger(L)
    assert p.c.hasher == 0 and ch.clone().c.hasher == 1
    bad = C.c_uint32(7)
    assert L.ola_challenger_init_hasher(C.byref(p.c), bad) != 0


@pytest.mark.skipif(not os.path.exists(LLVM_LIB), reason="LLVM's BLAKE3 is not on this machine")
def test_product_host_code_matches_llvm_blake3_directly():
    """The product's BLAKE3 (host instantiation of the header the kernels are compiled from) against the official C implementation
    itself, without the oracle in between: every leaf width 1..300 and a sample of wider ones, canonical little-endian words."""
    from olavm_amd.backend import _p, load_library
    from tests.golden.make_blake3_vectors import hasher
    L = load_library()
    h, _ = hasher()
    rng = np.random.default_rng(11)
    out = np.empty(4, dtype=np.uint64)
    for n in list(range(1, 301)) + [383, 384, 385, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096]:
        x = rng.integers(0, P, n, dtype=np.uint64)           # canonical already: the bytes hashed are the array's bytes
        assert L.ola_blake3_hash_elements(_p(x), n, _p(out)) == 0
        assert out.tobytes() == h(x.tobytes()), n


def test_oracle_proves_and_verifies_under_the_blake3_configuration(oracle):
    """prove_with_traces::<F, Blake3GoldilocksConfig, 2> -> verify (the configuration of the reference's own full-prove tests,
    circuits/src/stark/ola_stark.rs:684): same proof length as under Poseidon, different bytes, and neither verifier accepts
    the other configuration's proof."""
    from olavm_amd.air import ola_tables as T, tracegen
    blob = T.ola_stark().blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=5)
    pp = oracle.prove_with_traces(blob, traces, params, compress)
    with oracle.hasher("blake3"):
        pb = oracle.prove_with_traces(blob, traces, params, compress)
        assert oracle.verify_all_proof(blob, pb, params)[0] == 0
        assert oracle.verify_all_proof(blob, pp, params)[0] != 0
        tampered = bytearray(pb)
        tampered[len(pb) // 2] ^= 1
        assert oracle.verify_all_proof(blob, bytes(tampered), params)[0] != 0
    assert len(pb) == len(pp) and pb != pp
    assert oracle.verify_all_proof(blob, pp, params)[0] == 0
    assert oracle.verify_all_proof(blob, pb, params)[0] != 0
