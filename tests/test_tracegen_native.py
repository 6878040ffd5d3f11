"""The native trace generator (include/ola_tracegen.h) against the Python executor it restates: all twelve tables, word for
word, on every example program made of the instructions both support -- and the oracle's constraint check on its output."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from olavm_amd.air import fastexec, miniexec as M, ola_tables as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()


def test_library_exports_what_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "ola_tracegen.h")).read()
    declared = set(re.findall(r"\b(ola_tracegen_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(fastexec.EXPORTS)
    lib = fastexec.load_library()
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("program", ["fibonacci", "mixed", "memory", "hash", "call", "heap", "tape", "storage", "storage_heavy"])
def test_native_generator_reproduces_the_python_executor(program):
    factory, kwargs = M.EXAMPLES[program]
    want, params, compress = M.instance(factory(), **kwargs)
    got, gparams, gcompress = fastexec.instance(factory(), **kwargs)
    assert (params, compress) == (gparams, gcompress)
    for t, (a, b) in enumerate(zip(want, got)):
        assert a.shape == b.shape, (program, t, a.shape, b.shape)
        assert np.array_equal(a, b), (program, t, np.argwhere(a != b)[:5].tolist())


def test_native_generator_with_full_size_tables_and_a_long_run(oracle):
    """ola_stark() table sizes and 32-bit operands: identical to the Python executor; then a 2^18-row run whose tables
    satisfy every AIR (the Python executor needs a quarter of a minute for this one, the native one a fraction of a second)."""
    want, _, _ = M.instance(M.wide_program(), range_bits=16, limb_bits=8)
    got, params, _ = fastexec.instance(M.wide_program(), range_bits=16, limb_bits=8)
    for t, (a, b) in enumerate(zip(want, got)):
        assert np.array_equal(a, b), t
    traces, params, _ = fastexec.instance(M.memory_program(18000), range_bits=16, limb_bits=8, max_steps=1 << 22)
    assert traces[0].shape == (T.NUM_CPU_COLS, 1 << 18)
    s = T.ola_stark()
    from tests.test_oracle_stark import _per_table_params
    for i, (tr, pr) in enumerate(zip(traces, _per_table_params(s, params))):
        assert oracle.check_constraints(s.blob(), i, tr, pr) == -1, s.tables[i].name


def test_unsupported_and_faulty_programs_are_reported():
    p = M.Program()
    p.add("TSTORE", op0=1, op1=("imm", 1)).add("END")            # memory cell 0 was never written
    with pytest.raises(RuntimeError, match="never written"):
        fastexec.instance(p)
    q = M.Program()
    q.add("JMP", op1=("imm", 0))                   # never reaches END
    with pytest.raises(RuntimeError, match="does not terminate"):
        fastexec.instance(q, max_steps=100)
