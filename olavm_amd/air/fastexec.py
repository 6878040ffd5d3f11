"""ctypes binding of the native trace generator (include/ola_tracegen.h, olavm_amd/csrc/host/tracegen.cpp): the same
instance() as olavm_amd/air/miniexec.py -- 12 traces, params, compress challenges -- at native speed (a 2^22-row execution in seconds instead of minutes)."""
import ctypes as C
import os

import numpy as np

from . import ola_tables as T
from .dsl import P

_lib = None


class OlaInstr(C.Structure):
    _fields_ = [("op", C.c_uint32), ("dst", C.c_int32), ("op0", C.c_int32), ("op1", C.c_int32), ("op1_is_imm", C.c_uint32), ("imm", C.c_uint64)]


EXPORTS = ["ola_tracegen_run", "ola_tracegen_table", "ola_tracegen_cpu_rows", "ola_tracegen_free", "ola_tracegen_last_error", "ola_tracegen_betas"]
OLA_TRACEGEN_PROVE_PROGRAM_HASH, OLA_TRACEGEN_EXPLICIT_BETAS, OLA_TRACEGEN_REFERENCE_QUIRKS = 1, 2, 4


def lib_path():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libola_tracegen.so")


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path()):
            raise RuntimeError("libola_tracegen.so is missing: run `python __graft_entry__.py` (build) first")
        L = C.CDLL(lib_path())
        L.ola_tracegen_run.argtypes = [C.POINTER(OlaInstr), C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32,
                                       C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]
        L.ola_tracegen_table.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint64))]
        L.ola_tracegen_cpu_rows.argtypes = [C.c_void_p]
        L.ola_tracegen_cpu_rows.restype = C.c_uint64
        L.ola_tracegen_free.argtypes = [C.c_void_p]
        L.ola_tracegen_last_error.restype = C.c_char_p
        L.ola_tracegen_betas.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def encode(prog):
    """miniexec.Program -> array of OlaInstr."""
    arr = (OlaInstr * len(prog.ins))()
    for k, (op, dst, op0, op1) in enumerate(prog.ins):
        imm = isinstance(op1, tuple)
        arr[k] = OlaInstr(T.OPCODE_SHIFT[op], -1 if dst is None else dst, -1 if op0 is None else op0,
                          -1 if (op1 is None or imm) else op1, int(imm), (int(op1[1]) % P) if imm else 0)
    return arr


def instance(prog, range_bits=4, limb_bits=2, bitwise_beta=None, program_beta=None, prove_program_hash=False, max_steps=1 << 16, reference_quirks=False):
    """Same contract as miniexec.instance(prog, ...).  -> (traces, params, compress).  Betas left at None are derived by the
    generator's own Fiat-Shamir transcript, as the reference does; explicit values (both or neither) are for tests."""
    assert (bitwise_beta is None) == (program_beta is None), "give both compress challenges or neither"
    explicit = bitwise_beta is not None
    L = load_library()
    ins = encode(prog)
    code = (C.c_uint64 * 4)(*prog.code_addr)
    stor = (C.c_uint64 * 4)(*prog.storage_addr)
    handle = C.c_void_p()
    flags = ((OLA_TRACEGEN_PROVE_PROGRAM_HASH if prove_program_hash else 0) | (OLA_TRACEGEN_EXPLICIT_BETAS if explicit else 0) |
             (OLA_TRACEGEN_REFERENCE_QUIRKS if reference_quirks else 0))
    rc = L.ola_tracegen_run(ins, len(prog.ins), code, stor, range_bits, limb_bits, bitwise_beta if explicit else 0, program_beta if explicit else 0,
                            max_steps, flags, C.byref(handle))
    if rc != 0:
        raise RuntimeError("ola_tracegen_run: " + L.ola_tracegen_last_error().decode())
    try:
        traces = []
        for t in range(12):
            ncols, log_n, data = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint64)()
            assert L.ola_tracegen_table(handle, t, C.byref(ncols), C.byref(log_n), C.byref(data)) == 0
            n = 1 << log_n.value
            traces.append(np.ctypeslib.as_array(data, shape=(ncols.value, n)).copy())
        betas = (C.c_uint64 * 2)()
        assert L.ola_tracegen_betas(handle, betas) == 0
        bitwise_beta, program_beta = int(betas[0]), int(betas[1])
    finally:
        L.ola_tracegen_free(handle)
    return traces, [bitwise_beta, program_beta], [0, 0, bitwise_beta, 0, 0, 0, 0, 0, 0, 0, program_beta, 0]
