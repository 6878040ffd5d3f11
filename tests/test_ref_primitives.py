"""The hashing, transcript and FRI-parameter code of the reference, RUN from its source (PARITY.md "prover primitives by interpretation").

tools/rust_air_eval.py --primitives interprets, from /root/reference as it lies there,
  * `Poseidon::poseidon_naive` (plonky2 hash/poseidon.rs:617: constant_layer, sbox_layer, mds_layer with the constant tables of
    poseidon_goldilocks.rs) -- it reproduces the four known answers of the reference's own test (tests/golden/poseidon_kat.json);
  * `PoseidonHash::hash_no_pad` / `two_to_one` through hashing.rs's `hash_n_to_m_no_pad` and `compress`;
  * `Challenger` (iop/challenger.rs:36-162): new, observe_elements, observe_cap, get_n_challenges, compact, duplexing;
  * `StarkConfig::standard_fast_config()` (circuits/src/stark/config.rs:18) and `fri_params(degree_bits)` through
    `FriConfig::fri_params` and `FriReductionStrategy::reduction_arity_bits`;
and tests/golden/ref_primitive_vectors.json holds what they returned.  Here the oracle and the product's host code are held to
those values; with the reference tree present, the reference's `verify_merkle_proof_to_cap` (hash/merkle_proofs.rs:46),
interpreted, is the judge of the oracle's Merkle trees and authentication paths."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIXTURE = os.path.join(HERE, "golden", "ref_primitive_vectors.json")
P = 2**64 - 2**32 + 1


@pytest.fixture(scope="module")
def vectors():
    return json.load(open(FIXTURE))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from olavm_amd.backend import load_library
    return load_library()


def u64(v):
    return np.array(v, dtype=np.uint64)


def test_interpreted_permutation_reproduces_the_references_known_answers(vectors, oracle):
    kat = json.load(open(os.path.join(HERE, "golden", "poseidon_kat.json")))["vectors"]
    by_input = {tuple(v["input"]): v["output"] for v in vectors["poseidon"]}
    # two of the reference's KAT inputs (all zeros, 0..11) are among the interpreted inputs
    hits = [k for k in kat if tuple(x % P for x in k["input"]) in by_input]
    assert len(hits) >= 2
    for k in hits:
        assert by_input[tuple(x % P for x in k["input"])] == [x % P for x in k["output"]]
    for v in vectors["poseidon"]:
        assert [int(x) for x in oracle.poseidon(u64(v["input"]))] == v["output"]


def test_sponge_hash_and_compression_equal_the_interpreted_reference(vectors, oracle):
    assert sorted(len(v["input"]) for v in vectors["hash_no_pad"]) == [0, 1, 4, 5, 7, 8, 9, 15, 16, 17, 29, 76, 135]
    for v in vectors["hash_no_pad"]:
        assert [int(x) for x in oracle.hash_no_pad(u64(v["input"]))] == v["digest"], len(v["input"])
    for v in vectors["two_to_one"]:
        assert [int(x) for x in oracle.two_to_one(u64(v["left"]), u64(v["right"]))] == v["digest"]


def replay(script, ch, get):
    outs = []
    for op in script["ops"]:
        if op[0] == "observe":
            ch.observe(u64(op[1]))
        elif op[0] == "observe_cap":
            ch.observe_cap(u64(op[1]).reshape(-1, 4))
        elif op[0] == "get":
            outs.append(get(ch, op[1]))
        else:
            ch.compact()
            outs.append([int(x) for x in ch.state()])
    return outs, [int(x) for x in ch.state()]


def test_transcripts_equal_the_interpreted_challenger(vectors, oracle, lib):
    from olavm_amd.backend import Challenger
    assert len(vectors["challenger"]) == 6
    for script in vectors["challenger"]:
        outs, state = replay(script, oracle.challenger(), lambda c, n: [c.get() for _ in range(n)])
        assert outs == script["outputs"] and state == script["state"]
        outs, state = replay(script, Challenger(lib), lambda c, n: [int(x) for x in c.get(n)])
        assert outs == script["outputs"] and state == script["state"]


def test_proving_configuration_is_standard_fast_config(vectors, oracle):
    cfg = vectors["stark_config"]
    assert cfg == {"security_bits": 100, "num_challenges": 2, "rate_bits": 3, "cap_height": 4, "proof_of_work_bits": 16, "num_query_rounds": 28,
                   "reduction_strategy": ["ConstantArityBits", 4, 5]}
    # the library's defaults (NULL config) and the oracle's
    src = open(os.path.join(ROOT, "olavm_amd", "csrc", "ola_gpu.hip")).read()
    got = {k: int(v) for k, v in re.findall(r"\bd\.(\w+) = (\d+);", src)}
    want = {"rate_bits": cfg["rate_bits"], "cap_height": cfg["cap_height"], "proof_of_work_bits": cfg["proof_of_work_bits"],
            "fri_arity_bits": cfg["reduction_strategy"][1], "fri_final_poly_bits": cfg["reduction_strategy"][2],
            "num_query_rounds": cfg["num_query_rounds"], "num_challenges": cfg["num_challenges"]}
    assert {k: got[k] for k in want} == want
    hpp = open(os.path.join(ROOT, "oracle", "oracle.hpp")).read()
    o = {k: int(v) for k, v in re.findall(r"\b(rate_bits|cap_height|proof_of_work_bits|arity_bits|final_poly_bits|num_query_rounds) = (\d+)[,;]", hpp)
         if int(v)}
    assert o == {"rate_bits": 3, "cap_height": 4, "proof_of_work_bits": 16, "arity_bits": 4, "final_poly_bits": 5, "num_query_rounds": 28}
    from olavm_amd.backend import OlaGpuConfig
    assert [f[0] for f in OlaGpuConfig._fields_][2:9] == ["rate_bits", "cap_height", "proof_of_work_bits", "fri_arity_bits", "fri_final_poly_bits",
                                                         "num_query_rounds", "num_challenges"]


def test_fri_reduction_plans_equal_the_interpreted_reference(vectors, oracle):
    assert [v["degree_bits"] for v in vectors["fri_params"]] == list(range(31))
    for v in vectors["fri_params"]:
        assert v["hiding"] is False
        assert oracle.fri_arity_bits(v["degree_bits"]) == v["reduction_arity_bits"], v["degree_bits"]
    # the plan of the baseline's 2^22-row tables: five arity-16 folds, a 2^2-coefficient final polynomial
    assert vectors["fri_params"][22]["reduction_arity_bits"] == [4, 4, 4, 4, 4]


def test_lookup_columns_equal_the_interpreted_permuted_cols(vectors, oracle):
    """`permuted_cols` (stark/lookup.rs:68-132) interpreted on 54 input / table pairs -- range-check-like lookups, few distinct inputs, inputs
    all at the top / bottom of the table, tables with repeats and inputs absent from them, inputs above / below every table value, blocks, full
    width elements: the oracle's restatement (which the device generator is held to, tests/test_gpu_lookup.py) gives the same two columns"""
    assert len(vectors["permuted_cols"]) == 54
    for v in vectors["permuted_cols"]:
        pi, pt = oracle.permuted_cols(u64(v["inputs"]), u64(v["table"]))
        assert [int(x) for x in pi] == v["permuted_inputs"] and [int(x) for x in pt] == v["permuted_table"], v["case"]


def test_poseidon_table_rows_equal_the_interpreted_generator(vectors):
    """the reference's row generator for the Poseidon table (fast partial rounds, poseidon_trace.rs:79; generate_poseidon_trace with its
    padding rows) interpreted for five permutations with each lookup filter: the executor's rows (olavm_amd/air/miniexec.py, which the native
    trace generator is held to word for word) are the same 134 words, and the padding rows are the row of the zero input"""
    from olavm_amd.air import miniexec as M
    t = vectors["poseidon_table"]
    assert (t["columns"], t["rows"]) == (134, 8) and len(t["live"]) == 5
    for row, live in zip(t["trace_rows"], t["live"]):
        assert row == M.poseidon_row(live["input"], tuple(live["filters"]))
    for row in t["trace_rows"][5:]:
        assert row == M.poseidon_row([0] * 12)


@pytest.mark.gpu
def test_device_poseidon_table_equals_the_interpreted_generator(vectors):
    from olavm_amd.backend import Backend
    t = vectors["poseidon_table"]
    be = Backend()
    inputs = np.array([v["input"] for v in t["live"]] + [[0] * 12] * 3, dtype=np.uint64).T.copy()
    filters = np.array([v["filters"] for v in t["live"]] + [[0] * 4] * 3, dtype=np.uint64).T.copy()
    got = be.generate_poseidon_trace(inputs, filters)
    assert got.shape == (134, 8) and got.T.tolist() == t["trace_rows"]
    be.close()


@pytest.mark.gpu
def test_device_lookup_columns_equal_the_interpreted_permuted_cols(vectors):
    from olavm_amd.backend import Backend
    be = Backend()
    for v in vectors["permuted_cols"]:
        pi, pt = be.permuted_cols(u64(v["inputs"]), u64(v["table"]))
        assert [int(x) for x in pi] == v["permuted_inputs"] and [int(x) for x in pt] == v["permuted_table"], v["case"]
    be.close()


@pytest.mark.gpu
def test_device_hashes_equal_the_interpreted_reference(vectors):
    """the permutation, the row hash and the Merkle compression on the device against the outputs of the reference's own code"""
    from olavm_amd.backend import Backend
    be = Backend()
    st = np.array([v["input"] for v in vectors["poseidon"]], dtype=np.uint64)
    assert be.poseidon(st).tolist() == [v["output"] for v in vectors["poseidon"]]
    for v in vectors["hash_no_pad"]:
        if len(v["input"]) >= 1:                                  # ola_hash_rows is hash_no_pad of each row
            assert [int(x) for x in be.hash_rows(u64(v["input"]).reshape(1, -1))[0]] == v["digest"], len(v["input"])
    be.close()


reference = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")


@pytest.fixture(scope="module")
def interp():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rust_air_eval as R
    sys.setrecursionlimit(20000)
    return R, R.plonky2_interp("/root/reference")


@reference
def test_vectors_are_what_the_interpreter_computes_today(interp):
    R, _ = interp
    assert json.dumps(R.primitives("/root/reference"), separators=(",", ":")) + "\n" == open(FIXTURE).read()


@reference
def test_the_references_merkle_verifier_accepts_the_oracles_paths(interp, oracle):
    """`verify_merkle_proof_to_cap` as written in hash/merkle_proofs.rs:46-75, interpreted, on trees and authentication paths the
    oracle built (the product's are byte-equal to those: tests/test_gpu_parity.py) -- leaf order, sibling order, cap index."""
    from tests.oracle_lib import rand_field
    R, it = interp
    rng = np.random.default_rng(11)
    for ncols, log_n, cap_height, leaves in ((7, 4, 2, (0, 127)), (29, 3, 4, (5,)), (5, 2, 0, (31,))):
        b = oracle.batch(rand_field(rng, (ncols, 1 << log_n)), rate_bits=3, cap_height=cap_height)
        cap = b.cap()
        for idx in leaves:
            leaf, path = b.leaf(idx), b.prove(idx)
            assert len(path) == log_n + 3 - cap_height
            assert R.verify_merkle_proof_to_cap(it, leaf, idx, cap, path)
            # teeth: a changed sibling, a changed leaf element, a neighbouring index
            bad = path.copy()
            bad[len(bad) // 2, 1] ^= np.uint64(1)
            assert not R.verify_merkle_proof_to_cap(it, leaf, idx, cap, bad)
            wrong = leaf.copy()
            wrong[-1] = (int(wrong[-1]) + 1) % P
            assert not R.verify_merkle_proof_to_cap(it, wrong, idx, cap, path)
            assert not R.verify_merkle_proof_to_cap(it, leaf, idx ^ 1, cap, path)
