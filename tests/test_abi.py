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


def test_abi_revision_and_struct_sizes(lib):
    """ola_gpu_abi_version: the loaded library, the header and the ctypes structs agree (INTEGRATION.md "ABI revision")."""
    from olavm_amd.backend import OlaChallenger, OlaGpuConfig
    hdr = open(os.path.join(ROOT, "include", "ola_gpu.h")).read()
    want = int(re.search(r"#define OLA_GPU_ABI_VERSION (\d+)", hdr).group(1))
    a, b = C.c_size_t(), C.c_size_t()
    assert lib.ola_gpu_abi_version(C.byref(a), C.byref(b)) == want == 7
    assert a.value == C.sizeof(OlaChallenger) == 240 and b.value == C.sizeof(OlaGpuConfig)
    assert lib.ola_gpu_abi_version(None, None) == want
    # OlaScopeTime (revision 5): char[64], 4 x 32 bits, 3 doubles -- the layout integration/rust/ola_gpu_sys.rs declares
    from olavm_amd.backend import OlaScopeTime
    assert C.sizeof(OlaScopeTime) == 104 and OlaScopeTime.start_ms.offset == 80


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


def test_specialised_kernels_are_paired_with_the_blob_by_signature(lib):
    """Host-only entry point: the library hashes each table of the AIR-set blob it is handed (description + the static part of
    its lookups) and finds the kernel the code generator printed for the same signature -- for the three OlaStark variants
    of air/codegen.py; any change to a table (here: one more constraint) falls back to the interpreter kernel."""
    from olavm_amd.air import AirSet, ola_tables as T

    def available(airset):
        blob = np.ascontiguousarray(airset.blob(), dtype=np.uint64)
        flags = (C.c_uint8 * len(airset.tables))()
        rc = lib.ola_air_kernels_available(blob.ctypes.data_as(C.POINTER(C.c_uint64)), blob.size, flags, len(airset.tables))
        assert rc == 0, lib.ola_gpu_last_error()
        return [bool(x) for x in flags]

    for variant in (T.ola_stark(), T.ola_stark(range_bits=8, limb_bits=8), T.ola_stark(range_bits=4, limb_bits=2)):
        assert all(available(variant))
    s = T.ola_stark(range_bits=4, limb_bits=2)
    s.tables[8].constraint(s.tables[8].local(0) * s.tables[8].local(1))          # tape table, modified
    got = available(s)
    assert got == [i != 8 for i in range(12)]
    other = T.ola_stark(range_bits=5, limb_bits=2)                                 # a range-check table nobody generated
    assert available(other) == [i != 4 for i in range(12)]


def test_checked_in_airset_blob_is_the_ola_stark_and_fully_specialised():
    """include/ola_airset.bin is what the Rust side `include_bytes!`s: it must be exactly
    ola_stark().blob() -- the reference's 12 tables and 19 lookups at full size -- and every table must have its generated
    straight-line quotient kernel in this build."""
    import ctypes as C
    import os
    import numpy as np
    from olavm_amd.air import ola_tables as T
    from olavm_amd.backend import load_library, U64P
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ola_airset.bin")
    disk = np.fromfile(path, dtype="<u8")
    blob = np.ascontiguousarray(T.ola_stark().blob(), dtype=np.uint64)
    assert disk.shape == blob.shape and np.array_equal(disk, blob), "regenerate include/ola_airset.bin (olavm_amd/air/ola_tables.py changed)"
    lib = load_library()
    flags = (C.c_uint8 * 12)()
    assert lib.ola_air_kernels_available(disk.ctypes.data_as(U64P), disk.size, flags, 12) == 0
    assert all(flags), [bool(f) for f in flags]
