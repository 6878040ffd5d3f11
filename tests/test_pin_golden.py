"""Golden proofs of the real Rust prover (tests/golden/pin/*.traces + *.proof, produced by integration/pin/pin_dump.rs): the oracle
prover (here) and the GPU prover (-m gpu) must reproduce them byte for byte except for the pow_witness words.  Skipped while no
dump is committed (the build image has no Rust toolchain); the reader / comparer themselves are tested on a self-made dump."""
import importlib.util
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PIN = os.path.join(HERE, "golden", "pin")


def _tool():
    spec = importlib.util.spec_from_file_location("compare_with_dump", os.path.join(ROOT, "integration", "pin", "compare_with_dump.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _dumps():
    return sorted(f[:-7] for f in os.listdir(PIN) if f.endswith(".traces"))


def test_dump_format_round_trip_and_comparer(tmp_path, oracle):
    """The Python half of the harness on a dump written here: reader == writer, and the comparer accepts a proof that differs
    from the 'reference' one only in pow_witness-sized fields and rejects one that differs elsewhere."""
    from olavm_amd.air import miniexec as M, ola_tables as T
    m = _tool()
    traces, params, compress = M.instance(M.fibonacci(5))
    p = str(tmp_path / "fib.traces")
    m.write_traces(p, traces, compress)
    t2, c2 = m.read_traces(p)
    assert c2 == [int(c) for c in compress] and all(np.array_equal(a, b) for a, b in zip(traces, t2))
    proof = oracle.prove_with_traces(T.ola_stark(range_bits=4, limb_bits=2).blob(), traces, params, compress)
    assert m.compare("fib", proof, proof, "self")
    other = bytearray(proof)
    other[-8 * 13 - 4] ^= 1                                   # inside the last table's pow_witness
    assert m.compare("fib", bytes(other), proof, "one field")
    other[100] ^= 1
    other[300] ^= 1
    for k in range(12):
        other[1000 + 50 * k] ^= 1
    assert not m.compare("fib", bytes(other), proof, "many fields")


def test_wire_format_parser_names_the_first_differing_phase(tmp_path, oracle):
    """parse_all_proof tiles an AllProof (serialization.rs:349-393) span by span -- 12 tables, three caps, six opening vectors, the
    FRI caps, 28 query rounds, final polynomial, pow_witness each -- and first_difference turns a byte difference into the name of
    the phase that produced it; the .diag reader and the transcript check (CTL challenges from the trace caps through this
    backend's host challenger) agree with the oracle's challenger."""
    from olavm_amd.air import miniexec as M, ola_tables as T
    m = _tool()
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = M.instance(M.fibonacci(5))
    proof = oracle.prove_with_traces(blob, traces, params, compress)
    spans = m.parse_all_proof(proof)
    names = [n for n, _, _ in spans]
    assert sum(n.endswith("trace_cap") for n in names) == 12 and sum(n.endswith("pow_witness") for n in names) == 12
    assert names[-1] == "compress_challenges" and spans[-1][2] == len(proof)
    assert all(a <= b for _, a, b in spans) and all(spans[i][2] <= spans[i + 1][1] for i in range(len(spans) - 1))
    assert m.first_difference(proof, proof) is None
    by_name = {n: (a, b) for n, a, b in spans}
    for want in ("table 3: permutation_ctl_zs_cap", "table 0: openings.next_values", "table 7: fri.query[5].step[0].evals", "table 4: fri.query[27].initial_trees_proof[2].path",
                     "table 11: fri.final_poly"):
        found = [n for n in names if n.startswith(want)]
        if not found or by_name[found[0]][0] == by_name[found[0]][1]:
            continue                                          # a table too small to have that span in this instance
        key = found[0]
        a, b = by_name[key]
        other = bytearray(proof)
        other[a + (b - a) // 2] ^= 0x40
        other[-3] ^= 1                                        # a later difference must not mask the first one
        assert m.first_difference(bytes(other), proof) == key
    powd = bytearray(proof)
    a, b = by_name["table 2: fri.pow_witness"]
    powd[a] ^= 1
    assert m.first_difference(bytes(powd), proof) is None       # grinding nonces may differ
    # a proof whose shape differs (one opening fewer in a vector: the count word of span 4 decremented, eight... sixteen bytes cut)
    a4, b4 = spans[4][1], spans[4][2]
    (n4,) = struct.unpack_from("<I", proof, a4 - 4)
    if n4 > 0:
        shorter = proof[:a4 - 4] + struct.pack("<I", n4 - 1) + proof[a4:b4 - 16] + proof[b4:]
        assert "shape differs" in m.first_difference(shorter, proof)
    assert m.first_difference(proof[:spans[3][2]] + proof[spans[4][2]:], proof).startswith("(")      # unparsable: reported, not raised
    # transcript: challenges recomputed from the caps == the oracle challenger's
    och = oracle.challenger()
    for n, a, b in spans:
        if n.endswith("trace_cap"):
            och.observe(np.frombuffer(proof[a:b], dtype="<u8"))
    want_ch = [(och.get(), och.get()) for _ in range(2)]
    assert m.transcript_challenges(proof) == want_ch
    diag = tmp_path / "x.diag"
    diag.write_text("OLADIAG01\n" + "".join("ctl_challenge %d %d\n" % c for c in want_ch) + "table 0 trace_cap " + (struct.pack("<I", 16) + proof[by_name["table 0: trace_cap"][0]:by_name["table 0: trace_cap"][1]]).hex() + "\n")   # write_merkle_cap: count, digests
    dg = m.read_diag(str(diag))
    assert dg["ctl_challenges"] == want_ch and len(dg["caps"][(0, "trace_cap")]) == 16 * 32


@pytest.mark.skipif(not _dumps(), reason="no dump of the Rust prover committed (see tests/golden/pin/README.md)")
@pytest.mark.parametrize("name", _dumps() or ["none"])
def test_oracle_prover_reproduces_the_rust_proof(oracle, name):
    from olavm_amd.air import ola_tables as T
    m = _tool()
    traces, compress = m.read_traces(os.path.join(PIN, name + ".traces"))
    want = open(os.path.join(PIN, name + ".proof"), "rb").read()
    blob, params = T.ola_stark().blob(), [compress[2], compress[10]]
    assert oracle.verify_all_proof(blob, want, params)[0] == 0
    assert m.compare(name, oracle.prove_with_traces(blob, traces, params, compress), want, "oracle vs reference")
    b3 = os.path.join(PIN, name + ".blake3.proof")            # the same traces under Blake3GoldilocksConfig, when dumped
    if os.path.exists(b3):
        want3 = open(b3, "rb").read()
        with oracle.hasher("blake3"):
            assert oracle.verify_all_proof(blob, want3, params)[0] == 0
            assert m.compare(name, oracle.prove_with_traces(blob, traces, params, compress), want3, "oracle vs reference, Blake3")


@pytest.mark.gpu
@pytest.mark.skipif(not _dumps(), reason="no dump of the Rust prover committed (see tests/golden/pin/README.md)")
@pytest.mark.parametrize("name", _dumps() or ["none"])
def test_gpu_prover_reproduces_the_rust_proof(oracle, name):
    from olavm_amd.air import ola_tables as T
    from olavm_amd.backend import Backend
    m = _tool()
    traces, compress = m.read_traces(os.path.join(PIN, name + ".traces"))
    want = open(os.path.join(PIN, name + ".proof"), "rb").read()
    blob, params = T.ola_stark().blob(), [compress[2], compress[10]]
    be = Backend(device=0)
    try:
        assert m.compare(name, be.prove_with_traces(blob, traces, params, compress), want, "GPU vs reference")
    finally:
        be.close()
    b3 = os.path.join(PIN, name + ".blake3.proof")
    if os.path.exists(b3):
        be = Backend(device=0, hasher="blake3")
        try:
            assert m.compare(name, be.prove_with_traces(blob, traces, params, compress), open(b3, "rb").read(), "GPU vs reference, Blake3")
        finally:
            be.close()
