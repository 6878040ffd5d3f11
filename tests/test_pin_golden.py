"""Golden proofs of the real Rust prover (tests/golden/pin/*.traces + *.proof, produced by integration/pin/pin_dump.rs): the oracle
prover (here) and the GPU prover (-m gpu) must reproduce them byte for byte except for the pow_witness words.  Skipped while no
dump is committed (the build image has no Rust toolchain); the reader / comparer themselves are tested on a self-made dump."""
import importlib.util
import os

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
