"""Runs tests/host_api_check.cpp -- the reference's unit tests for this path written against the C++ mirror of its
interfaces (include/ola_host.hpp) -- on the GPU, and compares the AllProof bytes the C++ program produced with the
oracle's."""
import json
import os
import subprocess

import numpy as np
import pytest

from olavm_amd.air import ola_tables as T
from tests import host_api_build

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def write_fixture(path, sections):
    with open(path, "wb") as f:
        np.array([0x4F4C41484F5354, len(sections)], dtype=np.uint64).tofile(f)
        for s in sections:
            s = np.ascontiguousarray(s, dtype=np.uint64).reshape(-1)
            np.array([s.size], dtype=np.uint64).tofile(f)
            s.tofile(f)


def test_cpp_host_layer_passes_the_reference_style_checks_and_proves_the_same_bytes(tmp_path, oracle):
    from olavm_amd.air import miniexec as M
    kat = json.load(open(os.path.join(HERE, "golden", "poseidon_kat.json")))["vectors"]
    kat_words = [w for v in kat for w in v["input"] + v["output"]]
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    traces, params, compress = M.instance(M.mixed_program())
    logs = [int(t.shape[1]).bit_length() - 1 for t in traces]
    fixture, proof_path = tmp_path / "fixture.bin", tmp_path / "proof.bin"
    n_params = [t.n_params for t in s.tables]
    write_fixture(fixture, [kat_words, blob, logs, params, compress, n_params] + list(traces))
    exe = host_api_build.build(tmp_path)
    r = subprocess.run([exe, str(fixture), str(proof_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    proof = open(proof_path, "rb").read()
    assert proof == oracle.prove_with_traces(blob, traces, params, compress)
    rc, why = oracle.verify_all_proof(blob, proof, params)
    assert rc == 0, why


def test_c99_program_proves_from_separately_allocated_columns(tmp_path, oracle):
    """tests/host_c_abi_check.c (C99, linked against libola_gpu.so alone) holds the traces the way the reference does -- every column its
    own malloc (prover.rs:79-83, polynomial/mod.rs:24-26) -- and proves through ola_prove_with_traces_cols, then through
    ola_prove_with_traces from the contiguous tables: the program compares the two, this test compares with the oracle prover."""
    from olavm_amd.air import miniexec as M
    from olavm_amd.backend import lib_path
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    traces, params, compress = M.instance(M.mixed_program())
    logs = [int(t.shape[1]).bit_length() - 1 for t in traces]
    fixture, proof_path = tmp_path / "fixture.bin", tmp_path / "proof.bin"
    write_fixture(fixture, [[0], blob, logs, params, compress, [t.n_params for t in s.tables]] + list(traces))
    lib = os.path.dirname(lib_path())
    exe = os.path.join(str(tmp_path), "host_c_abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", os.path.join(HERE, "host_c_abi_check.c"), "-o", exe,
                           "-L" + lib, "-lola_gpu", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"])
    r = subprocess.run([exe, str(fixture), str(proof_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "per-column == contiguous" in r.stdout, r.stdout + r.stderr
    proof = open(proof_path, "rb").read()
    assert proof == oracle.prove_with_traces(blob, traces, params, compress)
