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


def random_program(rng, length=60):
    """A random straight-line program (with a few forward jumps) over the instructions whose operands need no set-up, kept
    valid by shadowing the register file: bitwise / comparison / range-check operands stay below 2^8 (miniature bitwise table),
    loads only touch written cells."""
    Pm = 0xFFFFFFFF00000001
    p = M.Program()
    regs = [0] * 10
    written = {}
    blocks, unknown, calls = [0], [2**50], [0]
    small = lambda r: regs[r] < 256

    def emit(op, dst=None, op0=None, op1=None, val=None):
        p.add(op, dst=dst, op0=op0, op1=op1)
        if dst is not None and val is not None:
            regs[dst] = val % Pm
    for _ in range(length):
        kind = rng.choice(["mov", "add", "mul", "not", "eq", "bit", "gte", "rc", "store", "load", "assert", "skip", "block", "call"])
        d, a, b = (int(x) for x in rng.integers(0, 9, 3))          # r9 (fp) stays 0
        if kind == "mov":
            v = int(rng.integers(0, 256)) if rng.integers(0, 2) else int(rng.integers(0, 2**63)) % Pm
            emit("MOV", d, None, ("imm", v), v)
        elif kind == "add":
            if rng.integers(0, 2):
                v = int(rng.integers(0, 2**62))
                emit("ADD", d, a, ("imm", v), regs[a] + v)
            else:
                emit("ADD", d, a, b, regs[a] + regs[b])
        elif kind == "mul":
            emit("MUL", d, a, b, regs[a] * regs[b])
        elif kind == "not":
            emit("NOT", d, None, b, Pm - 1 - regs[b])
        elif kind == "eq":
            op = "EQ" if rng.integers(0, 2) else "NEQ"
            emit(op, d, a, b, int((regs[a] == regs[b]) == (op == "EQ")))
        elif kind == "bit" and small(a) and small(b):
            op = str(rng.choice(["AND", "OR", "XOR"]))
            emit(op, d, a, b, {"AND": regs[a] & regs[b], "OR": regs[a] | regs[b], "XOR": regs[a] ^ regs[b]}[op])
        elif kind == "gte" and small(a) and small(b):
            emit("GTE", d, a, b, int(regs[a] >= regs[b]))
        elif kind == "rc" and small(b):
            emit("RC", None, None, b)
        elif kind == "store" and small(a):
            off = int(rng.integers(1, 40))
            emit("MSTORE", d, a, ("imm", off), regs[d])
            written[regs[a] + off] = regs[d]
        elif kind == "load" and written:
            addr = int(rng.choice(sorted(written)))
            emit("MOV", a, None, ("imm", 0), 0)
            emit("MLOAD", d, a, ("imm", addr), written[addr])
        elif kind == "assert":
            emit("EQ", d, a, a, 1)
            emit("ASSERT", None, None, d)
        elif kind == "block" and blocks[0] < 6:
            # eight fresh cells, hashed with the POSEIDON builtin, three of them appended to the tape and read back: the
            # digest and the loaded words are not shadowed exactly, so their registers / cells are marked "large"
            base = 1000 + 64 * blocks[0]
            blocks[0] += 1
            emit("MOV", 0, None, ("imm", base), base)
            for k in range(8):
                v = int(rng.integers(0, 2**40))
                emit("MOV", 1, None, ("imm", v), v)
                emit("MSTORE", 1, 0, ("imm", k), v)
                written[base + k] = v
            emit("MOV", 2, None, ("imm", base + 16), base + 16)
            emit("POSEIDON", 2, 0, ("imm", 8), base + 16)
            emit("MLOAD", 3, 2, ("imm", int(rng.integers(0, 4))), unknown[0])
            unknown[0] += 1
            emit("TSTORE", None, 0, ("imm", 3))
            emit("MOV", 4, None, ("imm", base + 32), base + 32)
            emit("MOV", 5, None, ("imm", 1), 1)
            emit("TLOAD", 4, 5, ("imm", 2), base + 32)
            emit("MLOAD", 6, 4, ("imm", 1), written[base + 2])       # tape[tp-1] = the third stored word
            if blocks[0] == 1:
                # the first block also goes through storage: slot key = its cells 0..3, value = cells 4..7, read back elsewhere
                emit("MOV", 7, None, ("imm", base + 4), base + 4)
                emit("SSTORE", None, 0, 7)
                emit("MOV", 8, None, ("imm", base + 40), base + 40)
                emit("SLOAD", None, 0, 8)
                emit("MLOAD", 3, 8, ("imm", 3), written[base + 7])
        elif kind == "call" and calls[0] < 3:
            # a frame at fp, the caller's frame pointer (0) saved at [fp - 2], a routine of one ADD behind a jump, CALL / RET
            fp = 600 + 8 * calls[0]
            calls[0] += 1
            emit("MOV", 9, None, ("imm", fp), fp)
            emit("MOV", 8, None, ("imm", 0), 0)
            emit("MSTORE", 8, 9, ("imm", Pm - 2), 0)
            here = len(p.words()[0])
            emit("CALL", None, None, ("imm", here + 4))
            emit("JMP", None, None, ("imm", here + 7))
            emit("ADD", 7, 7, ("imm", 1), regs[7] + 1)
            emit("RET")
            regs[9] = 0                                                   # RET restored the saved frame pointer
            written[fp - 2], written[fp - 1] = 0, here + 2
        elif kind == "skip":                                              # a taken conditional jump over one instruction
            emit("EQ", d, a, a, 1)
            here = len(p.words()[0])
            emit("CJMP", None, d, ("imm", here + 2 + 2))
            p.add("MOV", dst=b, op1=("imm", 12345))                       # skipped: the shadow register keeps its value
    p.add("END")
    return p


def test_random_programs_native_equals_python_and_all_airs_vanish(oracle):
    from tests.test_oracle_stark import _per_table_params
    s = T.ola_stark(range_bits=8, limb_bits=2)          # 16-bit range checks: the blocks live at addresses around 1000
    rng = np.random.default_rng(4242)
    for trial in range(8):
        prog = random_program(rng)
        want, params, _ = M.instance(prog, range_bits=8, limb_bits=2)
        got, _, _ = fastexec.instance(prog, range_bits=8, limb_bits=2)
        for t, (a, b) in enumerate(zip(want, got)):
            assert np.array_equal(a, b), (trial, t)
        for i, (tr, pr) in enumerate(zip(got, _per_table_params(s, params))):
            assert oracle.check_constraints(s.blob(), i, tr, pr) == -1, (trial, s.tables[i].name)


def test_compress_challenges_come_from_the_generators_transcript():
    """The bitwise and program compress challenges are Fiat-Shamir outputs (generation/builtin.rs:120-131, generation/prog.rs:
    23-29), not caller inputs: both generators derive the same pair, it changes with the run, and the explicit-beta test path
    still reproduces given values."""
    from olavm_amd.air import fastexec, miniexec as M
    from olavm_amd.backend import Challenger
    a = fastexec.instance(M.mixed_program())
    b = fastexec.instance(M.hash_program(2))
    assert a[1] != b[1] and a[2][2] == a[1][0] and a[2][10] == a[1][1]
    # the program challenge restated here: start / end state roots, limb by limb
    tree = M.StorageTree()
    start = tree.root()
    M.execute(M.mixed_program(), tree=tree)
    ch = Challenger()
    for x, y in zip(start, tree.root()):
        ch.observe([x, y])
    assert ch.get() == a[1][1]
    # the bitwise challenge restated here: the twelve limb columns in order
    bw = a[0][2]
    ch = Challenger()
    from olavm_amd.air import ola_tables as T
    for cols in (T.BW_OP0_LIMBS, T.BW_OP1_LIMBS, T.BW_RES_LIMBS):
        for i in range(4):
            ch.observe(bw[cols.start + i])
    assert ch.get() == a[1][0]
    c = fastexec.instance(M.mixed_program(), bitwise_beta=12345, program_beta=67890)
    assert c[1] == [12345, 67890]
