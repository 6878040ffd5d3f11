#!/usr/bin/env python3
"""Writes tests/golden/ref_verified/: a proof of the ORACLE prover and what the REFERENCE's verifier, run from its source, said of it.

    python -m tests.make_ref_verdict [poseidon] [blake3] [--prover-only]     (this container only: needs /root/reference; about 20 minutes each)

The instance is `miniexec.wide_program()` with its program-hash proof, against the reference's own AIR set (ola_stark(): range-check
table of 2^16 rows, bitwise table of 2^18): 32-bit operands through AND / OR / XOR / GTE / RC kept in memory between uses, the
program hashed by Poseidon and looked up in the state tree -- all twelve tables active except tape and sccall (padding only).
The oracle proves it (StarkConfig::standard_fast_config) under PoseidonGoldilocksConfig and under Blake3GoldilocksConfig -- the
configuration of the reference's own end-to-end tests (stark/ola_stark.rs:684); for each, tools/ref_verifier.py
  * re-encodes the decoded proof with the interpreted `Buffer::write_all_proof` (serialization.rs:377): the same bytes;
  * runs the interpreted `verify_proof` (verifier.rs:35): Ok(());
  * runs it again on the proof with ONE BIT flipped in each of a list of spans, recording where the reference's verifier stops;
  * records every challenge `AllProof::get_challenges` derives;
  * runs the interpreted PROVER, `prove_single_table` (prover.rs:330), on ten of the twelve tables -- from the table's trace, the lookup Z columns of
    the interpreted `cross_table_lookup_data` and the transcript state of `get_challenger_states` -- and compares its `write_proof` bytes
    with the table's bytes in the oracle's proof: equal, proof-of-work witness included.
wide_program[_blake3].proof is the oracle's proof, wide_program[_blake3].json the record.  tests/test_ref_verifier.py replays part of this where the
reference is present, checks the oracle's own verifier against the record everywhere, and (-m gpu) holds the GPU prover's bytes
for the same instance to wide_program.proof."""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "golden", "ref_verified")

TAMPER = ["table 0: trace_cap", "table 1: permutation_ctl_zs_cap", "table 5: quotient_polys_cap", "table 4: openings.local_values", "table 0: openings.next_values",
          "table 2: openings.permutation_ctl_zs", "table 10: openings.permutation_ctl_zs_next", "table 2: openings.ctl_zs_last", "table 0: openings.quotient_polys",
          "table 10: fri.commit_phase_merkle_caps[0]", "table 2: fri.query[0].initial_trees_proof[1].leaf", "table 2: fri.query[3].initial_trees_proof[0].path",
          "table 4: fri.query[27].initial_trees_proof[2].path", "table 2: fri.query[5].step[1].evals", "table 0: fri.query[9].step[0].path", "table 4: fri.final_poly",
          "table 7: fri.pow_witness", "compress_challenges[2]", "compress_challenges[10]"]


def instance():
    from olavm_amd.air import miniexec as M
    return M.instance(M.wide_program(), range_bits=16, limb_bits=8, prove_program_hash=True)


def tamper(raw, spans, name):
    """one flipped bit in the middle word of the span (for compress_challenges[k]: in word k)"""
    if name.startswith("compress_challenges["):
        a, _ = spans["compress_challenges"]
        off = a + 8 * int(name[len("compress_challenges["):-1])
    else:
        key = [n for n in spans if n.startswith(name)][0]
        a, b = spans[key]
        off = a + ((b - a) // 2 // 8) * 8
    out = bytearray(raw)
    out[off] ^= 1
    return bytes(out), off


def main():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "integration", "pin"))
    import compare_with_dump as CD
    import ref_verifier as V
    from olavm_amd.air import ola_tables as T
    from tests import oracle_lib
    oracle = oracle_lib.load()
    traces, params, compress = instance()
    blob = T.ola_stark().blob()
    for hasher in [a for a in sys.argv[1:] if a in ("poseidon", "blake3")] or ["poseidon", "blake3"]:
        with oracle.hasher(hasher):
            one(oracle, V, CD, blob, traces, params, compress, hasher)


# cpu (94 columns x 128 rows, 251 constraints), memory (32 rows), cmp (16 rows), poseidon (134 columns x 1024 rows: 37 minutes), poseidon_chunk,
# storage access (256 rows), tape, sccall, program (128 rows, permutation arguments), prog_chunk: every table but the two with fixed tables of
# 2^16 / 2^18 rows
PROVE_TABLES = [0, 1, 3, 5, 6, 7, 8, 9, 10, 11]


def prover_section(V, raw, traces, hasher, have=()):
    """the reference's prove_single_table, interpreted (tools/ref_verifier.py RefProver), on the small tables: its StarkProof bytes against the
    proof's, and the transcript state it leaves against the verifier side's"""
    rp = V.RefProver("/root/reference", hasher=hasher)
    out = [r for r in have if r["table"] in PROVE_TABLES]
    for k in PROVE_TABLES:
        if any(r["table"] == k for r in out):
            continue
        t = time.time()
        got, state_ok = rp.prove_table(raw, traces, k)
        a, b = V.table_span(raw, k)
        out.append({"table": k, "rows": int(traces[k].shape[1]), "columns": int(traces[k].shape[0]), "bytes": b - a, "equal": got == raw[a:b],
                    "transcript_after_equal": bool(state_ok), "sha256": hashlib.sha256(got).hexdigest()})
        print("%s: prove_single_table of table %d (%d x %d): %d bytes, equal %s, transcript after equal %s (%.0f s)" % (
            hasher, k, traces[k].shape[0], traces[k].shape[1], b - a, got == raw[a:b], state_ok, time.time() - t), flush=True)
    assert all(x["equal"] and x["transcript_after_equal"] for x in out)
    return sorted(out, key=lambda r: r["table"])


def one(oracle, V, CD, blob, traces, params, compress, hasher):
    stem = "wide_program" if hasher == "poseidon" else "wide_program_" + hasher
    if "--prover-only" in sys.argv:
        raw = open(os.path.join(OUT, stem + ".proof"), "rb").read()
        record = json.load(open(os.path.join(OUT, stem + ".json")))
        record["prove_single_table"] = prover_section(V, raw, traces, hasher, record.get("prove_single_table", ()))
        open(os.path.join(OUT, stem + ".json"), "w").write(json.dumps(record, indent=1) + "\n")
        return
    t = time.time()
    raw = oracle.prove_with_traces(blob, traces, params, compress)
    print("%s: oracle proof: %d bytes, %.0f s" % (hasher, len(raw), time.time() - t), flush=True)
    assert oracle.verify_all_proof(blob, raw, params) == (0, "")
    rv = V.RefVerifier("/root/reference", hasher=hasher)
    proof = V.decode_all_proof(raw, hasher)
    assert rv.encode(proof) == raw, "write_all_proof does not reproduce the bytes"
    challenges = rv.challenges(proof)
    t = time.time()
    ok, where = rv.verify(proof)
    assert rv.challenges(proof) == challenges and rv.encode(proof) == raw          # the verifier left the proof as it was
    print("reference verify_proof: %s (%.0f s)" % ("Ok(())" if ok else "Err at " + where, time.time() - t), flush=True)
    assert ok
    spans = {n: (a, b) for n, a, b in CD.parse_all_proof(raw)}
    record = {"generated_by": "python -m tests.make_ref_verdict",
              "instance": "miniexec.wide_program(), range_bits=16, limb_bits=8, prove_program_hash=True; ola_stark(); standard_fast_config",
              "config": {"poseidon": "PoseidonGoldilocksConfig", "blake3": "Blake3GoldilocksConfig"}[hasher],
              "trace_shapes": [[int(x) for x in tr.shape] for tr in traces],
              "proof_bytes": len(raw), "proof_sha256": hashlib.sha256(raw).hexdigest(),
              "write_all_proof_reproduces_the_bytes": True, "verify_proof": "Ok(())",
              "challenges": challenges, "tampered": []}
    for name in TAMPER:
        bad, off = tamper(raw, spans, name)
        t = time.time()
        ok, where = rv.verify(bad)
        # the oracle's verifier takes the two AIR parameters beside the proof: hand it the (possibly corrupted) words of the proof, as the
        # reference's verifier reads them (verifier.rs:81-90)
        cc = np.frombuffer(bad[-96:], dtype="<u8")
        o_rc, o_why = oracle.verify_all_proof(blob, bad, [int(cc[2]), int(cc[10])])
        print("%-55s reference: %-24s oracle: %s (%.0f s)" % (name, "Ok(())" if ok else "Err " + where, "accept" if o_rc == 0 else "reject", time.time() - t), flush=True)
        record["tampered"].append({"span": name, "byte": off, "bit": 0, "reference": "Ok(())" if ok else "Err " + where, "oracle_accepts": o_rc == 0})
    record["prove_single_table"] = prover_section(V, raw, traces, hasher)
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, stem + ".proof"), "wb").write(raw)
    open(os.path.join(OUT, stem + ".json"), "w").write(json.dumps(record, indent=1) + "\n")
    print("wrote", OUT, stem)


if __name__ == "__main__":
    main()
