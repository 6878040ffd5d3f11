"""ctypes binding of include/ola_gpu.h.  Plumbing only -- no arithmetic happens in Python."""
import ctypes as C
import os

import numpy as np

U64P = C.POINTER(C.c_uint64)
_HERE = os.path.dirname(os.path.abspath(__file__))

OLA_NTT_EVALUATE = 0
OLA_NTT_INTERPOLATE = 1
OLA_NTT_COSET_LDE = 2
OLA_NTT_COSET_INTERPOLATE = 3
OLA_NTT_COSET_LDE_LEAF_ORDER = 4


class OlaGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ola_gpu error {code}: {msg}")
        self.code = code


class OlaGpuConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("stream", C.c_void_p), ("rate_bits", C.c_uint32), ("cap_height", C.c_uint32),
                ("proof_of_work_bits", C.c_uint32), ("fri_arity_bits", C.c_uint32), ("fri_final_poly_bits", C.c_uint32),
                ("num_query_rounds", C.c_uint32), ("num_challenges", C.c_uint32), ("hasher", C.c_uint32)]


class OlaScopeTime(C.Structure):
    """include/ola_gpu.h OlaScopeTime: one `timed!` scope of the last proof with device times."""
    _fields_ = [("name", C.c_char * 64), ("depth", C.c_uint32), ("ref_depth", C.c_uint32), ("table", C.c_int32),
                ("is_reference_scope", C.c_uint32), ("start_ms", C.c_double), ("ms", C.c_double), ("sharded_ms", C.c_double)]


class OlaPassTime(C.Structure):
    """include/ola_gpu.h OlaPassTime: the launches of one transform-pass kernel instantiation, summed."""
    _fields_ = [("kernel", C.c_char * 48), ("launches", C.c_uint32), ("reserved", C.c_uint32), ("total_ms", C.c_double), ("elements", C.c_double)]


class OlaChallenger(C.Structure):
    _fields_ = [("sponge_state", C.c_uint64 * 12), ("input_buffer", C.c_uint64 * 8), ("output_buffer", C.c_uint64 * 8),
                ("input_len", C.c_uint32), ("output_len", C.c_uint32), ("hasher", C.c_uint32), ("reserved", C.c_uint32)]


# GenericConfig::Hasher (plonk/config.rs:112-161): PoseidonGoldilocksConfig / Blake3GoldilocksConfig
OLA_HASH_POSEIDON, OLA_HASH_BLAKE3 = 0, 1
HASHERS = {"poseidon": OLA_HASH_POSEIDON, "blake3": OLA_HASH_BLAKE3, OLA_HASH_POSEIDON: OLA_HASH_POSEIDON, OLA_HASH_BLAKE3: OLA_HASH_BLAKE3}


def lib_path():
    return os.path.join(_HERE, "lib", "libola_gpu.so")


_lib = None


def load_library():
    """Load libola_gpu.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise OlaGpuError(-7, f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the backend has no CPU fallback)")
    # torch bundles its own libamdhip64.so.7 + HSA runtime; two HIP runtimes in one process fight over the device, so
    # let torch's load first (same SONAME -> the dynamic loader then binds our library to the already-loaded one).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(p)
    L.ola_gpu_last_error.restype = C.c_char_p
    L.ola_gpu_init.argtypes = [C.POINTER(OlaGpuConfig), C.POINTER(C.c_void_p)]
    L.ola_gpu_abi_version.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    a, b = C.c_size_t(), C.c_size_t()
    if L.ola_gpu_abi_version(C.byref(a), C.byref(b)) != 7 or a.value != C.sizeof(OlaChallenger) or b.value != C.sizeof(OlaGpuConfig):
        raise OlaGpuError(-7, "libola_gpu.so and olavm_amd/backend.py disagree on the ABI revision or struct sizes: rebuild the library")
    L.ola_gpu_init_multi.argtypes = [C.POINTER(OlaGpuConfig), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]
    L.ola_gpu_device_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.ola_gpu_collective.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]
    L.ola_gpu_all_gather_check.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.ola_gpu_proof_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
    L.ola_gpu_phase_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_uint32]
    L.ola_gpu_free.argtypes = [C.c_void_p]
    L.ola_gpu_sync.argtypes = [C.c_void_p]
    L.ola_ntt_batch.argtypes = [C.c_void_p, C.c_int32, U64P, U64P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    L.ola_ntt_batch_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint64, C.c_uint32]
    L.ola_poseidon_permute.argtypes = [C.c_void_p, U64P, C.c_size_t]
    L.ola_hash_rows.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_size_t, U64P]
    L.ola_merkle_cap.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_size_t, C.c_uint32, U64P]
    L.ola_pow.argtypes = [C.c_void_p, U64P, C.c_uint32, U64P]
    for f in ("ola_commit_values", "ola_commit_coeffs"):
        getattr(L, f).argtypes = [C.c_void_p, C.POINTER(U64P), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), U64P]
    for f in ("ola_commit_values_dev", "ola_commit_coeffs_dev"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), U64P]
    L.ola_batch_free.argtypes = [C.c_void_p, C.c_void_p]
    L.ola_batch_shape.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ola_batch_get_coeffs.argtypes = [C.c_void_p, C.c_void_p, U64P]
    L.ola_batch_get_leaf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, U64P, U64P]
    L.ola_batch_get_lde_row.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, U64P]
    L.ola_challenger_init.argtypes = [C.POINTER(OlaChallenger)]
    L.ola_challenger_init_hasher.argtypes = [C.POINTER(OlaChallenger), C.c_uint32]
    L.ola_challenger_observe_cap.argtypes = [C.POINTER(OlaChallenger), U64P, C.c_size_t]
    L.ola_blake3_hash_elements.argtypes = [U64P, C.c_size_t, U64P]
    L.ola_challenger_observe.argtypes = [C.POINTER(OlaChallenger), U64P, C.c_size_t]
    L.ola_challenger_get.argtypes = [C.POINTER(OlaChallenger), U64P, C.c_size_t]
    L.ola_challenger_compact.argtypes = [C.POINTER(OlaChallenger)]
    L.ola_open_and_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(OlaChallenger),
                                     C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ola_prove_with_traces.argtypes = [C.c_void_p, U64P, C.c_size_t, C.POINTER(U64P), C.POINTER(C.c_uint32), U64P, U64P,
                                        C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ola_prove_with_traces_cols.argtypes = [C.c_void_p, U64P, C.c_size_t, C.POINTER(C.POINTER(U64P)), C.POINTER(C.c_uint32), U64P, U64P,
                                             C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ola_gpu_scope_times.argtypes = [C.c_void_p, C.c_int32, C.POINTER(OlaScopeTime), C.c_uint32, C.POINTER(C.c_uint32)]
    L.ola_gpu_upload_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.ola_gpu_warmup.argtypes = [C.c_int32, C.c_uint32, U64P, C.c_size_t]
    L.ola_gpu_warmup_wait.argtypes = [C.POINTER(C.c_double)]
    L.ola_gpu_ntt_pass_times.argtypes = [C.c_void_p, C.c_int32, C.POINTER(OlaPassTime), C.c_uint32, C.POINTER(C.c_uint32)]
    L.ola_commit_values_shard.argtypes = [C.c_void_p, C.POINTER(U64P), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_void_p), U64P]
    L.ola_commit_values_shard_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.POINTER(C.c_void_p), U64P]
    L.ola_generate_poseidon_trace.argtypes = [C.c_void_p, U64P, U64P, C.c_size_t, U64P]
    L.ola_prove_single_table.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_uint32, C.POINTER(U64P), C.c_void_p, U64P, U64P, U64P,
                                         C.POINTER(OlaChallenger), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ola_take_pending_proof.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ola_permuted_cols.argtypes = [C.c_void_p, U64P, U64P, C.c_size_t, U64P, U64P]
    L.ola_permuted_cols_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ola_set_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, ALL_GATHER_FN, C.c_void_p]
    L.ola_set_shard_options.argtypes = [C.c_void_p, C.c_uint32]
    L.ola_gpu_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.ola_table_shape.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
    L.ola_perm_z.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(U64P), U64P, U64P]
    L.ola_ctl_z.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(U64P), U64P, U64P]
    L.ola_quotient.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, U64P, U64P, U64P, U64P, U64P]
    L.ola_gpu_memory_stats.argtypes = [C.c_void_p, U64P, C.c_int32]
    L.ola_gpu_selftest.argtypes = [C.c_void_p, C.c_uint64, U64P]
    L.ola_gpu_reserve.argtypes = [C.c_void_p, U64P, C.c_size_t, C.POINTER(C.c_uint32)]
    L.ola_air_kernels_available.argtypes = [U64P, C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t]
    _lib = L
    return L


EXPORTS = [
    "ola_gpu_init", "ola_gpu_free", "ola_gpu_last_error", "ola_gpu_sync", "ola_ntt_batch", "ola_ntt_batch_dev",
    "ola_poseidon_permute", "ola_hash_rows", "ola_merkle_cap", "ola_commit_values", "ola_commit_coeffs",
    "ola_commit_values_dev", "ola_commit_coeffs_dev", "ola_batch_free", "ola_batch_shape", "ola_batch_get_coeffs",
    "ola_batch_get_leaf", "ola_batch_get_lde_row", "ola_challenger_init", "ola_challenger_observe",
    "ola_challenger_get", "ola_challenger_compact", "ola_challenger_init_hasher", "ola_challenger_observe_cap", "ola_blake3_hash_elements", "ola_open_and_prove", "ola_pow", "ola_prove_with_traces",
    "ola_air_kernels_available", "ola_commit_values_shard", "ola_commit_values_shard_dev", "ola_set_shard", "ola_gpu_trim", "ola_generate_poseidon_trace",
    "ola_permuted_cols", "ola_permuted_cols_dev", "ola_prove_single_table", "ola_take_pending_proof", "ola_gpu_memory_stats", "ola_gpu_selftest", "ola_gpu_reserve",
    "ola_table_shape", "ola_perm_z", "ola_ctl_z", "ola_quotient", "ola_set_shard_options", "ola_gpu_get_stream",
    "ola_gpu_abi_version", "ola_gpu_init_multi", "ola_gpu_device_count", "ola_gpu_proof_stats", "ola_gpu_phase_stats",
    "ola_gpu_collective", "ola_gpu_all_gather_check", "ola_prove_with_traces_cols", "ola_gpu_scope_times", "ola_gpu_upload_stats",
    "ola_gpu_warmup", "ola_gpu_warmup_wait", "ola_gpu_ntt_pass_times",
    "ola_open", "ola_fri_plan", "ola_fri_commit_begin", "ola_fri_commit_next_layer", "ola_fri_commit_finish", "ola_fri_query", "ola_fri_free",
]


OLA_WARMUP_PINNED_RING = 1


def warmup(device=-1, pinned_ring=True, airset=None):
    """ola_gpu_warmup: start the HIP runtime, open the device, load the code objects and -- with an AIR-set blob -- prime a context
    with a throw-away proof, all on a helper thread; returns at once (the reference's early hook: OlaStark::default() ->
    init_gpu(), circuits/src/stark/ola_stark.rs:47)."""
    L = load_library()
    if airset is not None:
        a = np.ascontiguousarray(airset, dtype=np.uint64)
        rc = L.ola_gpu_warmup(int(device), OLA_WARMUP_PINNED_RING if pinned_ring else 0, _p(a), a.size)
    else:
        rc = L.ola_gpu_warmup(int(device), OLA_WARMUP_PINNED_RING if pinned_ring else 0, None, 0)
    if rc != 0:
        raise OlaGpuError(rc, (L.ola_gpu_last_error() or b"").decode())


def warmup_wait():
    """ola_gpu_warmup_wait -> milliseconds the warm-up thread ran."""
    L = load_library()
    ms = C.c_double()
    rc = L.ola_gpu_warmup_wait(C.byref(ms))
    if rc != 0:
        raise OlaGpuError(rc, (L.ola_gpu_last_error() or b"").decode())
    return ms.value


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


def _p(a):
    return a.ctypes.data_as(U64P)


class _DeviceBytes:
    """View of raw device memory for torch (CUDA array interface): lets torch.distributed operate on the library's buffers."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


class Challenger:
    """Host-side Fiat-Shamir transcript (iop/challenger.rs:36-162), state lives in an OlaChallenger struct."""

    def __init__(self, lib=None, hasher="poseidon"):
        self.lib = lib or load_library()
        self.c = OlaChallenger()
        rc = self.lib.ola_challenger_init_hasher(C.byref(self.c), HASHERS[hasher])
        if rc != 0:
            raise OlaGpuError(rc, self.lib.ola_gpu_last_error().decode())

    def observe_cap(self, digests):
        """observe_cap: 4 elements per Poseidon digest, 5 (7 bytes each) per Blake3 digest."""
        d = np.ascontiguousarray(digests, dtype=np.uint64).reshape(-1, 4)
        self.lib.ola_challenger_observe_cap(C.byref(self.c), _p(d), d.shape[0])

    def observe(self, elems):
        e = np.ascontiguousarray(elems, dtype=np.uint64).ravel()
        self.lib.ola_challenger_observe(C.byref(self.c), _p(e), e.size)

    def get(self, n=None):
        out = np.empty(1 if n is None else n, dtype=np.uint64)
        self.lib.ola_challenger_get(C.byref(self.c), _p(out), out.size)
        return int(out[0]) if n is None else out

    def compact(self):
        self.lib.ola_challenger_compact(C.byref(self.c))

    def state(self):
        return np.array(list(self.c.sponge_state), dtype=np.uint64)

    def clone(self):
        o = Challenger(self.lib, int(self.c.hasher))
        C.memmove(C.byref(o.c), C.byref(self.c), C.sizeof(OlaChallenger))
        return o


class FriSteps:
    """An OlaFri: the opening proof of one table, one step per call (ola_fri_*; include/ola_gpu.h "one step per call")."""

    def __init__(self, be, handle):
        self.be, self.h = be, handle
        n, fl = C.c_uint32(), C.c_uint32()
        ab = (C.c_uint32 * 64)()
        be._chk(be.lib.ola_fri_plan(self.h, ab, 64, C.byref(n), C.byref(fl)))
        self.arity_bits, self.final_poly_len = [int(ab[i]) for i in range(n.value)], fl.value

    def begin(self, alpha):
        a = np.ascontiguousarray(alpha, dtype=np.uint64)
        self.be._chk(self.be.lib.ola_fri_commit_begin(self.h, _p(a)))

    def next_layer(self, beta=None):
        """-> the layer's cap (2^cap_height x 4 words); beta = the PREVIOUS layer's challenge, None for the first layer"""
        cap = np.empty((1 << self.be.cap_height, 4), dtype=np.uint64)
        b = None if beta is None else np.ascontiguousarray(beta, dtype=np.uint64)
        self.be._chk(self.be.lib.ola_fri_commit_next_layer(self.h, None if b is None else _p(b), _p(cap)))
        return cap

    def finish(self, beta=None):
        """-> final polynomial as (len, 2) words"""
        out = np.empty((max(self.final_poly_len, 1), 2), dtype=np.uint64)
        n = C.c_size_t(0)
        b = None if beta is None else np.ascontiguousarray(beta, dtype=np.uint64)
        self.be._chk(self.be.lib.ola_fri_commit_finish(self.h, None if b is None else _p(b), _p(out), out.shape[0], C.byref(n)))
        return out[:n.value]

    def query(self, x_index):
        x = np.ascontiguousarray(x_index, dtype=np.uint64)
        need = C.c_size_t(0)
        cap = 1 << 18
        while True:
            buf = C.create_string_buffer(cap)
            rc = self.be.lib.ola_fri_query(self.h, _p(x), x.size, buf, cap, C.byref(need))
            if rc != 0 and need.value > cap:
                cap = need.value
                continue
            self.be._chk(rc)
            return bytes(buf.raw[:need.value])

    def free(self):
        if self.h:
            self.be.lib.ola_fri_free(self.h)
            self.h = None


class Batch:
    """A committed PolynomialBatch resident in HBM (fri/oracle.rs:31-39)."""

    def __init__(self, be, handle, cap, shard_log_world=0):
        self.be, self.h, self._cap, self.shard_log_world = be, handle, cap, shard_log_world
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        be._chk(be.lib.ola_batch_shape(handle, C.byref(a), C.byref(b), C.byref(c)))
        self.ncols, self.log_n, self.rate_bits = a.value, b.value, c.value

    def cap(self):
        return self._cap

    def coeffs(self):
        out = np.empty((self.ncols, 1 << self.log_n), dtype=np.uint64)
        self.be._chk(self.be.lib.ola_batch_get_coeffs(self.be.ctx, self.h, _p(out)))
        return out

    def leaf(self, index):
        depth = self.log_n + self.rate_bits + self.shard_log_world - self.be.cap_height   # rate_bits is the local one
        row = np.empty(self.ncols, dtype=np.uint64)
        sib = np.empty((max(depth, 0), 4), dtype=np.uint64)
        self.be._chk(self.be.lib.ola_batch_get_leaf(self.be.ctx, self.h, index, _p(row), _p(sib) if depth > 0 else None))
        return row, sib

    def lde_row(self, index, step=1):
        row = np.empty(self.ncols, dtype=np.uint64)
        self.be._chk(self.be.lib.ola_batch_get_lde_row(self.be.ctx, self.h, index, step, _p(row)))
        return row

    def free(self):
        if self.h:
            self.be.lib.ola_batch_free(self.be.ctx, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Backend:
    """One OlaCtx.  `stream` may be a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).
    devices=[d0, d1, ...] makes it ONE context spanning those GPUs (ola_gpu_init_multi): prove_with_traces then runs on the
    coset partition inside the library, one call, one process; entries may repeat (logical ranks sharing a GPU)."""

    def __init__(self, device=-1, stream=None, devices=None, **cfg):
        self.lib = load_library()
        c = OlaGpuConfig(device, stream, cfg.get("rate_bits", 3), cfg.get("cap_height", 4),
                         cfg.get("proof_of_work_bits", 16), cfg.get("fri_arity_bits", 4),
                         cfg.get("fri_final_poly_bits", 5), cfg.get("num_query_rounds", 28), cfg.get("num_challenges", 2),
                         HASHERS[cfg.get("hasher", "poseidon")])
        self.hasher = int(c.hasher)
        self.cap_height = c.cap_height
        self.rate_bits = c.rate_bits
        self.ctx = C.c_void_p()
        if devices is not None:
            dv = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            # collective="rccl" | "peer": the library reads OLA_COLLECTIVE when the context is created (ola_gpu_collective tells what it got)
            want, old = cfg.get("collective"), os.environ.get("OLA_COLLECTIVE")
            if want is not None:
                os.environ["OLA_COLLECTIVE"] = want
            try:
                self._chk(self.lib.ola_gpu_init_multi(C.byref(c), dv, len(devices), C.byref(self.ctx)))
            finally:
                if want is not None:
                    if old is None:
                        del os.environ["OLA_COLLECTIVE"]
                    else:
                        os.environ["OLA_COLLECTIVE"] = old
        else:
            self._chk(self.lib.ola_gpu_init(C.byref(c), C.byref(self.ctx)))

    PHASES = ("leaf_hash", "merkle_levels", "fri_fold", "lde", "intt", "quotient", "open_eval")

    def phase_stats(self):
        """ola_gpu_phase_stats of the last proof (accounting must be on): {phase: (ms, units0, units1)}."""
        n = len(self.PHASES)
        out = (C.c_double * (6 * n))()
        self._chk(self.lib.ola_gpu_phase_stats(self.ctx, out, 2 * n))
        st = {name: (out[3 * i], out[3 * i + 1], out[3 * i + 2]) for i, name in enumerate(self.PHASES)}
        # rows n + p: the phase's dominant scope (most bytes): (ms, bytes)
        self.phase_top = {name: (out[3 * (n + i)], out[3 * (n + i) + 1]) for i, name in enumerate(self.PHASES)}
        return st

    def device_count(self):
        n = C.c_uint32()
        self._chk(self.lib.ola_gpu_device_count(self.ctx, C.byref(n)))
        return n.value

    def collective(self):
        """ola_gpu_collective: who carries this context's exchanges -> {"carrier": "none" | "peer" | "rccl", "ranks": n, "note": str}"""
        carrier, ranks = C.c_uint32(), C.c_uint32()
        note = C.create_string_buffer(512)
        self._chk(self.lib.ola_gpu_collective(self.ctx, C.byref(carrier), C.byref(ranks), note, 512))
        return {"carrier": ("none", "peer", "rccl")[carrier.value], "ranks": ranks.value, "note": note.value.decode()}

    def all_gather_check(self, carrier, bytes_per_rank, reps=10):
        """ola_gpu_all_gather_check through "peer" or "rccl": -> (ms per gather on the slowest rank, wrong bytes over all ranks)"""
        ms, bad = C.c_double(), C.c_uint64()
        self._chk(self.lib.ola_gpu_all_gather_check(self.ctx, {"peer": 1, "rccl": 2}[carrier], bytes_per_rank, reps, C.byref(ms), C.byref(bad)))
        return ms.value, bad.value

    def proof_stats(self, enable=None):
        """ola_gpu_proof_stats: switch the accounting (True / False / None = leave) and return the last proof's figures."""
        out = (C.c_double * 8)()
        self._chk(self.lib.ola_gpu_proof_stats(self.ctx, -1 if enable is None else int(bool(enable)), out))
        return {"wall_ms": out[0], "sharded_ms_upto2": out[1], "sharded_ms_upto4": out[2], "sharded_ms_upto8": out[3],
                "exchange_bytes": int(out[4]), "exchanges": int(out[5]), "peer_exchanges": int(out[6]), "peer_bytes_moved": int(out[7])}

    def scope_times(self, enable=None):
        """ola_gpu_scope_times: switch the recording (True / False / None = leave); -> the last proof's `timed!` scopes as dicts."""
        n = C.c_uint32()
        self._chk(self.lib.ola_gpu_scope_times(self.ctx, -1 if enable is None else int(bool(enable)), None, 0, C.byref(n)))
        if n.value == 0:
            return []
        out = (OlaScopeTime * n.value)()
        self._chk(self.lib.ola_gpu_scope_times(self.ctx, -1, out, n.value, C.byref(n)))
        return [{"name": o.name.decode(), "depth": o.depth, "ref_depth": o.ref_depth, "table": o.table, "reference": bool(o.is_reference_scope),
                 "start_ms": o.start_ms, "ms": o.ms, "sharded_ms": o.sharded_ms} for o in out]

    def ntt_pass_times(self, enable=None):
        """ola_gpu_ntt_pass_times: switch the per-launch events of the transform passes (True / False / None = leave); -> what was
        recorded since the last call as {kernel: {"launches", "total_ms", "avg_ms", "elements"}}."""
        n = C.c_uint32()
        out = (OlaPassTime * 64)()
        self._chk(self.lib.ola_gpu_ntt_pass_times(self.ctx, -1 if enable is None else int(bool(enable)), out, 64, C.byref(n)))
        return {o.kernel.decode(): {"launches": o.launches, "total_ms": o.total_ms, "avg_ms": o.total_ms / max(o.launches, 1), "elements": o.elements}
                for o in out[:n.value]}

    def upload_stats(self):
        """ola_gpu_upload_stats of the last whole proof."""
        out = (C.c_double * 8)()
        self._chk(self.lib.ola_gpu_upload_stats(self.ctx, out))
        return {"waited_ms": out[0], "total_ms": out[1], "first_group_ms": out[2], "bytes": int(out[3]),
                "mode": ("staged", "pageable")[int(out[4])], "threads": int(out[5]), "link_bytes": int(out[6])}

    def _chk(self, rc):
        if rc != 0:
            raise OlaGpuError(rc, (self.lib.ola_gpu_last_error() or b"").decode())

    def close(self):
        if self.ctx:
            self.lib.ola_gpu_free(self.ctx)
            self.ctx = None

    def sync(self):
        self._chk(self.lib.ola_gpu_sync(self.ctx))

    # ---- NTT (host arrays, shape (batch, n)) ----
    def ntt(self, op, data, shift=7, blowup_log=0):
        d = np.ascontiguousarray(data, dtype=np.uint64)
        if d.ndim == 1:
            d = d[None, :]
        batch, n = d.shape
        log_n = int(n).bit_length() - 1
        grows = op in (OLA_NTT_COSET_LDE, OLA_NTT_COSET_LDE_LEAF_ORDER)
        out = np.empty((batch, n << blowup_log if grows else n), dtype=np.uint64)
        self._chk(self.lib.ola_ntt_batch(self.ctx, op, _p(d), _p(out), log_n, batch, shift, blowup_log))
        return out

    def ntt_dev(self, op, in_ptr, out_ptr, log_n, batch, shift=7, blowup_log=0, scratch_ptr=None):
        self._chk(self.lib.ola_ntt_batch_dev(self.ctx, op, in_ptr, out_ptr, scratch_ptr, log_n, batch, shift, blowup_log))

    # ---- hashing ----
    def poseidon(self, states):
        s = np.array(states, dtype=np.uint64).reshape(-1, 12)
        self._chk(self.lib.ola_poseidon_permute(self.ctx, _p(s), s.shape[0]))
        return s

    def hash_rows(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty((r.shape[0], 4), dtype=np.uint64)
        self._chk(self.lib.ola_hash_rows(self.ctx, _p(r), r.shape[0], r.shape[1], _p(out)))
        return out

    def merkle_cap(self, leaves, cap_height):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64)
        out = np.empty((1 << cap_height, 4), dtype=np.uint64)
        self._chk(self.lib.ola_merkle_cap(self.ctx, _p(lv), lv.shape[0], lv.shape[1], cap_height, _p(out)))
        return out

    def pow(self, h4, bits=16):
        h = np.ascontiguousarray(h4, dtype=np.uint64)
        w = np.zeros(1, dtype=np.uint64)
        self._chk(self.lib.ola_pow(self.ctx, _p(h), bits, _p(w)))
        return int(w[0])

    # ---- commitments ----
    def commit(self, cols, from_coeffs=False):
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        ncols, n = cols.shape
        ptrs = (U64P * ncols)(*[cols[i].ctypes.data_as(U64P) for i in range(ncols)])
        h = C.c_void_p()
        cap = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        f = self.lib.ola_commit_coeffs if from_coeffs else self.lib.ola_commit_values
        self._chk(f(self.ctx, ptrs, ncols, int(n).bit_length() - 1, C.byref(h), _p(cap)))
        return Batch(self, h, cap)

    def commit_dev(self, dev_ptr, ncols, log_n, from_coeffs=False):
        h = C.c_void_p()
        cap = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        f = self.lib.ola_commit_coeffs_dev if from_coeffs else self.lib.ola_commit_values_dev
        self._chk(f(self.ctx, dev_ptr, ncols, log_n, C.byref(h), _p(cap)))
        return Batch(self, h, cap)

    def commit_shard(self, cols, rank, world, dev_ptr=None, ncols=None, log_n=None):
        """This GPU's share of a commitment under the coset partition (ola_commit_values_shard): -> Batch whose cap() is
        the slice [rank*16/world, (rank+1)*16/world) of the full Merkle cap.  Pass `cols` (host array) or dev_ptr/ncols/
        log_n (device-resident column-major values)."""
        lw = int(world).bit_length() - 1
        h = C.c_void_p()
        cap = np.empty(((1 << self.cap_height) >> lw, 4), dtype=np.uint64)
        if dev_ptr is not None:
            self._chk(self.lib.ola_commit_values_shard_dev(self.ctx, dev_ptr, ncols, log_n, rank, world, C.byref(h), _p(cap)))
        else:
            cols = np.ascontiguousarray(cols, dtype=np.uint64)
            ncols, n = cols.shape
            ptrs = (U64P * ncols)(*[cols[i].ctypes.data_as(U64P) for i in range(ncols)])
            self._chk(self.lib.ola_commit_values_shard(self.ctx, ptrs, ncols, int(n).bit_length() - 1, rank, world, C.byref(h), _p(cap)))
        return Batch(self, h, cap, shard_log_world=lw)

    def generate_poseidon_trace(self, inputs, filters=None):
        """Poseidon STARK table (134 x n) from permutation inputs (12 x n) and optional lookup filters (4 x n)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
        assert inputs.shape[0] == 12
        n = inputs.shape[1]
        f = None if filters is None else np.ascontiguousarray(filters, dtype=np.uint64)
        out = np.empty((134, n), dtype=np.uint64)
        self._chk(self.lib.ola_generate_poseidon_trace(self.ctx, _p(inputs), None if f is None else _p(f), n, _p(out)))
        return out

    def permuted_cols(self, inputs, table):
        """lookup.rs permuted_cols on the device: -> (sorted canonical inputs, permuted table), both uint64 arrays."""
        a = np.ascontiguousarray(inputs, dtype=np.uint64)
        b = np.ascontiguousarray(table, dtype=np.uint64)
        assert a.ndim == 1 and a.shape == b.shape
        pi, pt = np.empty_like(a), np.empty_like(a)
        self._chk(self.lib.ola_permuted_cols(self.ctx, _p(a), _p(b), a.shape[0], _p(pi), _p(pt)))
        return pi, pt

    def permuted_cols_dev(self, in_ptr, table_ptr, n, out_in_ptr, out_table_ptr):
        self._chk(self.lib.ola_permuted_cols_dev(self.ctx, in_ptr, table_ptr, n, out_in_ptr, out_table_ptr))

    def trim(self):
        """Return the context's cached device buffers to the driver (ola_gpu_trim)."""
        self._chk(self.lib.ola_gpu_trim(self.ctx))

    # ---- per-phase entry points (one `timed!` scope of prove_single_table at a time) ----
    def table_shape(self, airset_blob, table):
        """-> dict(ncols, n_params, perm_zs, ctl_zs, quotient_degree_factor, permutation_batch_size)."""
        blob = np.ascontiguousarray(airset_blob, dtype=np.uint64)
        out = (C.c_uint32 * 6)()
        self._chk(self.lib.ola_table_shape(self.ctx, _p(blob), blob.size, table, out))
        return dict(zip(("ncols", "n_params", "perm_zs", "ctl_zs", "quotient_degree_factor", "permutation_batch_size"), (int(x) for x in out)))

    def _zs_phase(self, fn, airset_blob, table, trace, challenges, count):
        blob = np.ascontiguousarray(airset_blob, dtype=np.uint64)
        tr = np.ascontiguousarray(trace, dtype=np.uint64)
        ptrs = (U64P * tr.shape[0])(*[_p(tr[c]) for c in range(tr.shape[0])])
        cc = np.ascontiguousarray(np.array(challenges, dtype=np.uint64).reshape(-1))
        out = np.zeros((count, tr.shape[1]), dtype=np.uint64)
        self._chk(fn(self.ctx, _p(blob), blob.size, table, int(tr.shape[1]).bit_length() - 1, ptrs, _p(cc), _p(out) if count else _p(np.zeros(1, dtype=np.uint64))))
        return out

    def perm_z(self, airset_blob, table, trace, perm_challenges):
        """Permutation Z columns (values): perm_challenges = [batch_size][num_challenges] pairs (beta, gamma)."""
        return self._zs_phase(self.lib.ola_perm_z, airset_blob, table, trace, perm_challenges, self.table_shape(airset_blob, table)["perm_zs"])

    def ctl_z(self, airset_blob, table, trace, ctl_challenges):
        """CTL Z columns (values): ctl_challenges = [num_challenges] pairs (beta, gamma)."""
        return self._zs_phase(self.lib.ola_ctl_z, airset_blob, table, trace, ctl_challenges, self.table_shape(airset_blob, table)["ctl_zs"])

    def quotient(self, airset_blob, table, trace_batch, zs_batch, perm_challenges, ctl_challenges, alphas, params, n):
        """Coefficients of the 2 * quotient_degree_factor quotient chunk polynomials, [chunks][n]."""
        blob = np.ascontiguousarray(airset_blob, dtype=np.uint64)
        q = self.table_shape(airset_blob, table)["quotient_degree_factor"]
        pc = None if perm_challenges is None else np.ascontiguousarray(np.array(perm_challenges, dtype=np.uint64).reshape(-1))
        cc = np.ascontiguousarray(np.array(ctl_challenges, dtype=np.uint64).reshape(-1))
        al = np.ascontiguousarray(alphas, dtype=np.uint64)
        pr = None if params is None or len(params) == 0 else np.ascontiguousarray(params, dtype=np.uint64)
        out = np.zeros((2 * q, n), dtype=np.uint64)
        self._chk(self.lib.ola_quotient(self.ctx, _p(blob), blob.size, table, trace_batch.h, zs_batch.h, None if pc is None else _p(pc), _p(cc), _p(al),
                                        None if pr is None else _p(pr), _p(out)))
        return out

    def reserve(self, airset_blob, log_ns):
        """Start allocating the buffers of a coming proof in the background (ola_gpu_reserve); returns at once."""
        blob = np.ascontiguousarray(airset_blob, dtype=np.uint64)
        logs = (C.c_uint32 * len(log_ns))(*[int(x) for x in log_ns])
        self._chk(self.lib.ola_gpu_reserve(self.ctx, _p(blob), blob.size, logs))

    def selftest(self, pairs=1 << 28):
        """Device field-arithmetic self-test (carry-flag reduction against the C++ form): number of mismatches."""
        out = np.zeros(1, dtype=np.uint64)
        self._chk(self.lib.ola_gpu_selftest(self.ctx, int(pairs), _p(out)))
        return int(out[0])

    def memory_stats(self, reset=False):
        """Device memory of the context's pool in bytes: dict(live, live_peak, reserved, reserved_peak)."""
        out = np.zeros(4, dtype=np.uint64)
        self._chk(self.lib.ola_gpu_memory_stats(self.ctx, _p(out), 1 if reset else 0))
        return dict(zip(("live", "live_peak", "reserved", "reserved_peak"), (int(x) for x in out)))

    def set_shard(self, rank, world, group=None):
        """Coset-partitioned proving (ola_set_shard): this context is rank `rank` of `world` GPUs.  The all-gather the
        library asks for runs through torch.distributed on `group` -- RCCL when the process group is "nccl" (device
        buffers are handed over as they are), host staging for "gloo" (tests).  world = 1 switches back."""
        if world == 1:
            self._chk(self.lib.ola_set_shard(self.ctx, 0, 1, ALL_GATHER_FN(0), None))
            self._shard_cb = None
            return
        import torch
        import torch.distributed as dist
        on_device = dist.get_backend(group) == "nccl"

        self.shard_calls = 0          # exchanges performed so far (observability / tests)

        # RCCL path: the collective is launched on the context's own stream (torch ExternalStream), so it is ordered with the
        # library's kernels by the stream itself -- no host synchronisation per exchange (OLA_SHARD_STREAM_ORDERED).
        ext = None
        if on_device:
            sp = C.c_void_p()
            self._chk(self.lib.ola_gpu_get_stream(self.ctx, C.byref(sp)))
            ext = torch.cuda.ExternalStream(sp.value)

        def all_gather(_user, send, recv, nbytes):
            self.shard_calls += 1
            try:
                src = torch.as_tensor(_DeviceBytes(send, nbytes), device="cuda")
                dst = torch.as_tensor(_DeviceBytes(recv, nbytes * world), device="cuda")
                if on_device:
                    with torch.cuda.stream(ext):
                        dist.all_gather_into_tensor(dst, src, group=group)
                else:
                    parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(parts, src.cpu(), group=group)
                    dst.copy_(torch.cat(parts))
                    torch.cuda.synchronize()
                return 0
            except Exception as e:          # noqa: BLE001 -- must not unwind through the C frame
                import sys
                print("ola all_gather callback failed:", repr(e), file=sys.stderr)
                return 1

        self._shard_cb = ALL_GATHER_FN(all_gather)      # keep the trampoline alive as long as the context uses it
        self._chk(self.lib.ola_set_shard(self.ctx, rank, world, self._shard_cb, None))
        self._chk(self.lib.ola_set_shard_options(self.ctx, 1 if on_device else 0))

    def prove_with_traces(self, airset_blob, traces, params=None, compress=None, cap=8 << 20):
        """AllProof bytes for the multi-table STARK described by `airset_blob` (olavm_amd.air.AirSet.blob()).

        A table is a 2-d numpy array (host, one column-major block), a contiguous 64-bit torch tensor resident on this GPU, or a
        LIST of 1-d uint64 arrays -- every column its own allocation, the reference's Vec<PolynomialValues<F>> (prover.rs:79-83).
        As soon as one table is a list the call goes through ola_prove_with_traces_cols (blocks become their column pointers)."""
        blob = np.ascontiguousarray(airset_blob, dtype=np.uint64)
        pr = None if params is None else np.ascontiguousarray(params, dtype=np.uint64)
        cc = None if compress is None else np.ascontiguousarray(compress, dtype=np.uint64)
        need = C.c_size_t(0)
        buf = C.create_string_buffer(cap)
        if any(isinstance(t, (list, tuple)) for t in traces):
            keep, tabs, logs = [], [], []
            for t in traces:
                if isinstance(t, (list, tuple)):
                    cols = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1) for c in t]
                    n = cols[0].size
                    if any(c.size != n for c in cols):
                        raise ValueError("columns of one table differ in length")
                    addrs = [c.ctypes.data for c in cols]
                elif hasattr(t, "data_ptr"):
                    if not (t.is_contiguous() and t.element_size() == 8):
                        raise ValueError("device-resident tables must be contiguous 64-bit tensors")
                    cols, n = t, int(t.shape[1])
                    addrs = [t.data_ptr() + 8 * n * c for c in range(int(t.shape[0]))]
                else:
                    cols = np.ascontiguousarray(t, dtype=np.uint64)
                    n = cols.shape[1]
                    addrs = [cols.ctypes.data + 8 * n * c for c in range(cols.shape[0])]
                arr = (U64P * len(addrs))(*[C.cast(C.c_void_p(a), U64P) for a in addrs])
                keep.append((cols, arr))
                tabs.append(arr)
                logs.append(n.bit_length() - 1)
            ptrs = (C.POINTER(U64P) * len(tabs))(*[C.cast(a, C.POINTER(U64P)) for a in tabs])
            logs = (C.c_uint32 * len(logs))(*logs)
            rc = self.lib.ola_prove_with_traces_cols(self.ctx, _p(blob), blob.size, ptrs, logs, None if pr is None else _p(pr),
                                                     None if cc is None else _p(cc), buf, cap, C.byref(need))
        else:
            tr = [t if hasattr(t, "data_ptr") else np.ascontiguousarray(t, dtype=np.uint64) for t in traces]
            for t in tr:
                if hasattr(t, "data_ptr") and not (t.is_contiguous() and t.element_size() == 8):
                    raise ValueError("device-resident tables must be contiguous 64-bit tensors")
            ptrs = (U64P * len(tr))(*[C.cast(C.c_void_p(t.data_ptr()), U64P) if hasattr(t, "data_ptr") else _p(t) for t in tr])
            logs = (C.c_uint32 * len(tr))(*[int(t.shape[1]).bit_length() - 1 for t in tr])
            rc = self.lib.ola_prove_with_traces(self.ctx, _p(blob), blob.size, ptrs, logs, None if pr is None else _p(pr),
                                                None if cc is None else _p(cc), buf, cap, C.byref(need))
        if rc != 0 and need.value > cap:            # the proof is kept in the context: fetch it, do not prove again
            buf = C.create_string_buffer(need.value)
            rc = self.lib.ola_take_pending_proof(self.ctx, buf, need.value, C.byref(need))
        self._chk(rc)
        return bytes(buf.raw[:need.value])

    def prove_single_table(self, airset_blob, table, trace, batch, ctl_challenges, params, challenger):
        """StarkProof bytes of one table (ola_prove_single_table): `batch` is the table's trace commitment, `challenger` the
        shared transcript (a Challenger, advanced in place), ctl_challenges = [(beta, gamma)] * num_challenges."""
        blob = np.ascontiguousarray(airset_blob, dtype=np.uint64)
        tr = np.ascontiguousarray(trace, dtype=np.uint64)
        ptrs = (U64P * tr.shape[0])(*[_p(tr[c]) for c in range(tr.shape[0])])
        cc = np.ascontiguousarray(np.array(ctl_challenges, dtype=np.uint64).reshape(-1))
        cap_words = np.ascontiguousarray(batch.cap(), dtype=np.uint64).reshape(-1)
        pr = None if params is None or len(params) == 0 else np.ascontiguousarray(params, dtype=np.uint64)
        need = C.c_size_t(0)
        cap = 8 << 20          # larger than any proof of the supported sizes: a too-small buffer costs a second full proof
        while True:
            buf = C.create_string_buffer(cap)
            rc = self.lib.ola_prove_single_table(self.ctx, _p(blob), blob.size, table, ptrs, batch.h, _p(cap_words), _p(cc),
                                                 None if pr is None else _p(pr), C.byref(challenger.c), buf, cap, C.byref(need))
            if rc != 0 and need.value > cap:
                cap = need.value
                continue
            self._chk(rc)
            return bytes(buf.raw[:need.value])

    def air_kernels_available(self, airset_blob, ntables):
        """-> list of bool: which tables of the AIR set have a specialised quotient kernel in this build."""
        blob = np.ascontiguousarray(airset_blob, dtype=np.uint64)
        flags = (C.c_uint8 * ntables)()
        self._chk(self.lib.ola_air_kernels_available(_p(blob), blob.size, flags, ntables))
        return [bool(x) for x in flags]

    def open(self, trace, zs, quot, num_permutation_zs, zeta):
        """ola_open: StarkOpeningSet::new at `zeta` = (a, b) -> (opening-set bytes in wire format, FriSteps for the rest of the opening proof)"""
        z = np.ascontiguousarray(zeta, dtype=np.uint64)
        need, h = C.c_size_t(0), C.c_void_p()
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            rc = self.lib.ola_open(self.ctx, trace.h, zs.h, quot.h, num_permutation_zs, _p(z), buf, cap, C.byref(need), C.byref(h))
            if rc != 0 and need.value > cap:
                cap = need.value
                continue
            self._chk(rc)
            return bytes(buf.raw[:need.value]), FriSteps(self, h)

    def open_and_prove(self, trace, zs, quot, num_permutation_zs, challenger):
        need, olen = C.c_size_t(0), C.c_size_t(0)
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            rc = self.lib.ola_open_and_prove(self.ctx, trace.h, zs.h, quot.h, num_permutation_zs, C.byref(challenger.c), buf, cap,
                                             C.byref(need), C.byref(olen))
            if rc != 0 and need.value > cap:
                cap = need.value
                continue
            self._chk(rc)
            return bytes(buf.raw[:olen.value]), bytes(buf.raw[olen.value:need.value])
