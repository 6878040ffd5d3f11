"""The reference's verifier, run from its source, on a proof of this repository (PARITY.md "the reference's verifier").

tests/golden/ref_verified/wide_program[_blake3].proof are proofs of the oracle prover for the reference's own AIR set under
PoseidonGoldilocksConfig and Blake3GoldilocksConfig (see tests/make_ref_verdict.py for the instance); the .json beside each records what tools/ref_verifier.py -- the reference's
`verify_proof`, `AllProof::get_challenges` and `Buffer::write_all_proof`, interpreted from /root/reference -- made of it:
the writer gives back the bytes, the verifier returns Ok(()), on nineteen one-bit corruptions it stops where recorded, and the
reference's PROVER (`prove_single_table`, interpreted) writes the same bytes for ten of the twelve tables.

  * everywhere: the record and the proof file belong together; the oracle's verifier agrees with the reference's on the proof and
    on all nineteen corruptions; the product's host transcript (the C library's challenger) re-derives the challenges the
    reference's `get_challenges` derived;
  * where the reference tree is present: the interpretation is repeated (encode, verify, two corruptions, all challenges);
  * -m gpu: the GPU prover's bytes for the same instance are the file's bytes -- so the reference's verifier has accepted exactly
    what `ola_prove_with_traces` returns."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DIR = os.path.join(HERE, "golden", "ref_verified")
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "integration", "pin"))


STEM = {"poseidon": "wide_program", "blake3": "wide_program_blake3"}


@pytest.fixture(scope="module", params=["poseidon", "blake3"])
def config(request):
    """PoseidonGoldilocksConfig / Blake3GoldilocksConfig (the configuration of the reference's end-to-end tests, ola_stark.rs:684)"""
    return request.param


@pytest.fixture(scope="module")
def record(config):
    return json.load(open(os.path.join(DIR, STEM[config] + ".json")))


@pytest.fixture(scope="module")
def raw(config):
    return open(os.path.join(DIR, STEM[config] + ".proof"), "rb").read()


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from olavm_amd.backend import load_library
    return load_library()


def test_record_belongs_to_the_proof_and_says_accepted(record, raw, config):
    import compare_with_dump as CD
    assert record["config"] == {"poseidon": "PoseidonGoldilocksConfig", "blake3": "Blake3GoldilocksConfig"}[config]
    assert len(raw) == record["proof_bytes"] and hashlib.sha256(raw).hexdigest() == record["proof_sha256"]
    assert record["verify_proof"] == "Ok(())" and record["write_all_proof_reproduces_the_bytes"] is True
    spans = CD.parse_all_proof(raw)
    assert sum(n.endswith("trace_cap") for n, _, _ in spans) == 12
    # the instance: the reference's full-size fixed tables, and every table but tape / sccall with real rows
    shapes = record["trace_shapes"]
    assert shapes[2][1] == 1 << 18 and shapes[4][1] == 1 << 16 and shapes[0][1] >= 128 and shapes[5][1] >= 1024 and shapes[7][1] >= 256
    assert len(record["tampered"]) == 19
    stops = {t["reference"] for t in record["tampered"]}
    # the corruptions reach the quotient identity, the proof of work, the Merkle paths and the FRI consistency checks
    assert {"Err verifier.rs:295", "Err verifier.rs:52", "Err merkle_proofs.rs:71"} <= stops


def test_oracle_verifier_agrees_with_the_reference_verifier_on_every_case(record, raw, oracle, config):
    from olavm_amd.air import ola_tables as T
    blob = T.ola_stark().blob()
    compress = np.frombuffer(raw[-96:], dtype="<u8")              # the proof's last field: the twelve compress challenges
    params = [int(compress[2]), int(compress[10])]                # the AIR parameters: the bitwise and the program table's
    with oracle.hasher(config):
        assert oracle.verify_all_proof(blob, raw, params) == (0, "")
        for t in record["tampered"]:
            bad = bytearray(raw)
            bad[t["byte"]] ^= 1 << t["bit"]
            # a corrupted compress challenge reaches the oracle's verifier as the AIR parameter it is
            c = np.frombuffer(bytes(bad[-96:]), dtype="<u8")
            rc, _ = oracle.verify_all_proof(blob, bytes(bad), [int(c[2]), int(c[10])])
            assert (rc == 0) == t["oracle_accepts"], t["span"]
            assert (rc == 0) == (t["reference"] == "Ok(())"), t["span"]
    with oracle.hasher({"poseidon": "blake3", "blake3": "poseidon"}[config]):
        assert oracle.verify_all_proof(blob, raw, params)[0] != 0          # the other configuration's verifier does not take it
    assert sum(t["reference"] != "Ok(())" for t in record["tampered"]) >= 18


def cap_array(cap):
    """digests as the C ABI takes them: 32 bytes = 4 little-endian words each, under either configuration"""
    if "elements" in cap[0][0]:
        return np.array([[e.v for e in h["elements"]] for h in cap[0]], dtype=np.uint64)
    return np.frombuffer(b"".join(bytes(h[0]) for h in cap[0]), dtype="<u8").reshape(-1, 4).copy()


def ext_words(values):
    out = []
    for e in values:
        out += [e.a, e.b]
    return np.array(out, dtype=np.uint64)


def test_host_transcript_rederives_the_challenges_the_reference_derived(record, raw, lib, oracle, config):
    """get_challenges.rs:18-150 replayed with the product's host challenger over the decoded proof: lookup challenges, then per table
    compact / permutation challenge sets / alphas / zeta / FRI alpha, betas, proof-of-work response, query indices."""
    import ref_verifier as V
    from olavm_amd.air import ola_tables as T
    from olavm_amd.backend import Challenger
    proof = V.decode_all_proof(raw, config)
    want = record["challenges"]
    tables = T.ola_stark().tables
    ch = Challenger(lib, hasher=config)
    for sp in proof["stark_proofs"]:
        ch.observe_cap(cap_array(sp["trace_cap"]))
    got = [int(x) for x in ch.get(4)]
    assert got == [want["ctl_challenges"]["challenges"][0]["beta"], want["ctl_challenges"]["challenges"][0]["gamma"],
                   want["ctl_challenges"]["challenges"][1]["beta"], want["ctl_challenges"]["challenges"][1]["gamma"]]
    for t, sp in enumerate(proof["stark_proofs"]):
        w = want["stark_challenges"][t]
        ch.compact()
        sets = w["permutation_challenge_sets"]
        if sets is not None:
            for s in sets:
                for c in s["challenges"]:
                    assert [int(x) for x in ch.get(2)] == [c["beta"], c["gamma"]]
        ch.observe_cap(cap_array(sp["permutation_ctl_zs_cap"]))
        assert [int(x) for x in ch.get(2)] == w["stark_alphas"]
        ch.observe_cap(cap_array(sp["quotient_polys_cap"]))
        assert [int(x) for x in ch.get(2)] == w["stark_zeta"]
        op = sp["openings"]
        ch.observe(ext_words(op["local_values"] + op["permutation_ctl_zs"] + op["quotient_polys"]))
        ch.observe(ext_words(op["next_values"] + op["permutation_ctl_zs_next"]))
        last = np.zeros(2 * len(op["ctl_zs_last"]), dtype=np.uint64)
        last[0::2] = np.array([e.v for e in op["ctl_zs_last"]], dtype=np.uint64)
        ch.observe(last)
        f = w["fri_challenges"]
        fri = sp["opening_proof"]
        assert [int(x) for x in ch.get(2)] == f["fri_alpha"]
        assert len(f["fri_betas"]) == len(fri["commit_phase_merkle_caps"])
        for cap, beta in zip(fri["commit_phase_merkle_caps"], f["fri_betas"]):
            ch.observe_cap(cap_array(cap))
            assert [int(x) for x in ch.get(2)] == beta
        ch.observe(ext_words(fri["final_poly"]["coeffs"]))
        h = [int(x) for x in ch.get(4)]
        # the proof-of-work response (challenges.rs:55-66): hash of the transcript hash and the witness; 16 leading zero bits
        assert int(oracle.hash_no_pad(np.array(h + [fri["pow_witness"].v], dtype=np.uint64))[0]) == f["fri_pow_response"] < 1 << 48
        degree_bits = len(fri["query_round_proofs"][0]["initial_trees_proof"]["evals_proofs"][0][1]["siblings"]) + 4 - 3
        idx = [int(x) % (1 << (degree_bits + 3)) for x in ch.get(28)]
        assert idx == f["fri_query_indices"], tables[t].name
        assert len(h) == 4


reference = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")


@reference
def test_the_references_writer_and_verifier_today(record, raw, config):
    import ref_verifier as V
    rv = V.RefVerifier("/root/reference", hasher=config)
    proof = V.decode_all_proof(raw, config)
    assert rv.encode(proof) == raw
    assert rv.challenges(proof) == record["challenges"]
    assert rv.verify(proof) == (True, None)
    for t in record["tampered"][:1] + [x for x in record["tampered"] if x["span"] == "table 0: openings.quotient_polys"]:
        bad = bytearray(raw)
        bad[t["byte"]] ^= 1 << t["bit"]
        ok, where = rv.verify(bytes(bad))
        assert (("Ok(())" if ok else "Err " + where) == t["reference"]), t["span"]


def test_record_of_the_references_prover(record):
    """tests/make_ref_verdict.py ran the reference's `prove_single_table` (interpreted, full proof-of-work search from 0) on ten of the twelve
    tables -- all but the two with 2^16 / 2^18-row fixed tables: the StarkProof bytes it wrote are the proof's, and the transcript it left is the
    verifier side's"""
    rows = {r["table"]: r for r in record["prove_single_table"]}
    assert sorted(rows) == [0, 1, 3, 5, 6, 7, 8, 9, 10, 11]
    assert (rows[5]["columns"], rows[5]["rows"]) == (134, 1024)         # the Poseidon table: two FRI reduction layers
    assert all(r["equal"] and r["transcript_after_equal"] for r in rows.values())
    assert (rows[0]["columns"], rows[0]["rows"]) == (94, 128)           # the CPU table: 251 constraints
    assert rows[10]["rows"] == 128 and rows[3]["rows"] == 16            # the program table has permutation arguments; cmp is looked up by the CPU table
    assert rows[7]["rows"] == 256                                       # storage access: the concurrent Merkle build, one FRI reduction
    assert sum(r["bytes"] for r in rows.values()) == 371_224            # of the proof's 586 616 bytes


@reference
def test_the_references_prover_today(record, raw, config):
    """`prove_single_table` (prover.rs:330-567) interpreted again for the cmp table (16 rows of live comparisons): from the trace, the interpreted `cross_table_lookup_data` and the transcript state of `get_challenger_states` to the
    bytes of `write_proof` -- equal to the table's bytes in the proof.  The proof-of-work search is replaced by a check of the proof's
    witness and of 64 smaller candidates (the search from 0 is in the record)."""
    import hashlib as H
    import ref_verifier as V
    from tests.make_ref_verdict import instance
    traces, _, _ = instance()
    rp = V.RefProver("/root/reference", hasher=config)
    by = {r["table"]: r for r in record["prove_single_table"]}
    for k in (3,):
        got, state_ok = rp.prove_table(raw, traces, k, full_pow_search=False)
        a, b = V.table_span(raw, k)
        assert got == raw[a:b] and state_ok, k
        assert H.sha256(got).hexdigest() == by[k]["sha256"]


@reference
def test_a_changed_reference_prover_is_noticed(tmp_path):
    """teeth of the byte comparison: a copy of the reference with one line of its prover changed no longer writes the proof's bytes for the
    cmp table -- (a) the order of the opening batches the transcript observes swapped (stark/proof.rs), (b) the quotient read one LDE step off
    (stark/prover.rs: `next_step`).  (Dropping the multiplication of the final polynomial by X, fri/oracle.rs:218, makes the changed prover
    panic in log2_strict: tried by hand.)"""
    import shutil
    import ref_verifier as V
    import rust_air_eval as R
    from tests.make_ref_verdict import instance
    raw = open(os.path.join(DIR, "wide_program.proof"), "rb").read()
    traces, _, _ = instance()
    a, b = V.table_span(raw, 3)
    edits = [("circuits/src/stark/proof.rs", ".chain(&self.permutation_ctl_zs)\n                .chain(&self.quotient_polys)",
              ".chain(&self.quotient_polys)\n                .chain(&self.permutation_ctl_zs)"),
             ("circuits/src/stark/prover.rs", "let next_step = 1 << quotient_degree_bits;", "let next_step = 2 << quotient_degree_bits;")]
    for n, (rel, old, new) in enumerate(edits):
        ref = tmp_path / ("ref%d" % n)
        for sub in ("circuits/src", "core/src", "plonky2/field/src", "plonky2/plonky2/src", "plonky2/util/src"):
            shutil.copytree(os.path.join("/root/reference", sub), ref / sub)
        path = ref / rel
        text = path.read_text()
        assert text.count(old) >= 1, rel
        path.write_text(text.replace(old, new, 1))
        R.X.Src.cache.clear()
        try:
            got, state_ok = V.RefProver(str(ref)).prove_table(raw, traces, 3, full_pow_search=False)
            noticed = got != raw[a:b] or not state_ok
        except (R.RustError, ZeroDivisionError, IndexError, AssertionError):
            noticed = True                     # the changed prover panics, fails its own checks, or the proof's witness no longer fits its transcript
        assert noticed, rel
    R.X.Src.cache.clear()


@pytest.mark.gpu
def test_gpu_prover_returns_the_bytes_the_reference_verifier_accepted(record, raw, config):
    from olavm_amd.air import ola_tables as T
    from olavm_amd.backend import Backend
    from tests.make_ref_verdict import instance
    traces, params, compress = instance()
    assert [[int(x) for x in tr.shape] for tr in traces] == record["trace_shapes"]
    be = Backend(hasher=config)
    got = bytes(be.prove_with_traces(T.ola_stark().blob(), traces, params, compress))
    assert hashlib.sha256(got).hexdigest() == record["proof_sha256"]
    assert got == raw
    # and from separately allocated columns (ola_prove_with_traces_cols)
    cols = [[np.ascontiguousarray(tr[c]) for c in range(tr.shape[0])] for tr in traces]
    assert bytes(be.prove_with_traces(T.ola_stark().blob(), cols, params, compress)) == raw
    be.close()
