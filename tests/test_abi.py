"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that include/ola_gpu.h
declares, argument validation and error reporting work without a GPU, and the host-side transcript matches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from olavm_amd.backend import load_library
    return load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ola_gpu.h")).read()
    declared = set(re.findall(r"\b(ola_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    from olavm_amd.backend import EXPORTS
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from olavm_amd.backend import Backend, OlaGpuError
    with pytest.raises(OlaGpuError) as e:
        Backend()
    assert e.value.code == -2 and "no HIP device" in str(e.value)


def test_null_arguments_are_rejected(lib):
    assert lib.ola_gpu_init(None, None) == -1
    assert b"out_ctx" in lib.ola_gpu_last_error()
    assert lib.ola_challenger_init(None) == -1


def test_host_challenger_matches_oracle(lib, oracle):
    from olavm_amd.backend import Challenger
    rng = np.random.default_rng(5)
    ch, och = Challenger(lib), oracle.challenger()
    for step in range(40):
        k = int(rng.integers(0, 12))
        e = rng.integers(0, 2**64, size=k, dtype=np.uint64)  # includes non-canonical inputs
        ch.observe(e)
        och.observe(e)
        if step % 3 == 0:
            n = int(rng.integers(1, 11))
            assert [int(x) for x in ch.get(n)] == [och.get() for _ in range(n)]
        if step % 7 == 0:
            ch.compact()
            och.compact()
            assert np.array_equal(ch.state(), och.state())
