"""ctypes binding of oracle/liboracle.so -- the CPU restatement used ONLY as the checker in tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFF00000001
U64P = C.POINTER(C.c_uint64)


def build():
    srcs = [f for f in os.listdir(ODIR) if f.endswith((".cpp", ".hpp"))]
    so = os.path.join(ODIR, "liboracle.so")
    newest = max(os.path.getmtime(os.path.join(ODIR, f)) for f in srcs)
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["make", "-C", ODIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def ptr(a):
    return a.ctypes.data_as(U64P)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.oracle_root_of_unity.restype = C.c_uint64
        L.oracle_root_of_unity.argtypes = [C.c_int]
        L.oracle_gl_pow.restype = C.c_uint64
        L.oracle_gl_pow.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_gl_vec_op.argtypes = [C.c_int, U64P, U64P, U64P, C.c_size_t]
        L.oracle_evaluate_poly.argtypes = [U64P, C.c_size_t]
        L.oracle_interpolate_poly.argtypes = [U64P, C.c_size_t]
        L.oracle_evaluate_poly_with_offset.argtypes = [U64P, C.c_size_t, C.c_uint64, C.c_size_t, U64P]
        L.oracle_interpolate_poly_with_offset.argtypes = [U64P, C.c_size_t, C.c_uint64]
        L.oracle_naive_eval.argtypes = [U64P, C.c_size_t, C.c_size_t, C.c_uint64, U64P]
        L.oracle_eval_at_points.argtypes = [U64P, C.c_size_t, U64P, C.c_size_t, U64P]
        L.oracle_poseidon.argtypes = [U64P]
        L.oracle_permuted_cols.argtypes = [U64P, U64P, C.c_size_t, U64P, U64P]
        L.oracle_hash_no_pad.argtypes = [U64P, C.c_size_t, U64P]
        L.oracle_two_to_one.argtypes = [U64P, U64P, U64P]
        L.oracle_merkle.argtypes = [U64P, C.c_size_t, C.c_size_t, C.c_int, U64P, U64P, U64P]
        L.oracle_merkle_selfcheck.argtypes = [U64P, C.c_size_t, C.c_size_t, C.c_int]
        L.oracle_merkle_selfcheck.restype = C.c_int
        for f in ("oracle_batch_from_values", "oracle_batch_from_coeffs"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [U64P, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_batch_free.argtypes = [C.c_void_p]
        for f in ("oracle_batch_cap", "oracle_batch_coeffs", "oracle_batch_leaves"):
            getattr(L, f).argtypes = [C.c_void_p, U64P]
        L.oracle_batch_leaf.argtypes = [C.c_void_p, C.c_size_t, U64P]
        L.oracle_batch_leaf.restype = None
        L.oracle_batch_rehash.argtypes = [C.c_void_p]
        L.oracle_batch_rehash.restype = None
        L.oracle_batch_prove.argtypes = [C.c_void_p, C.c_size_t, U64P]
        L.oracle_batch_prove.restype = C.c_int
        L.oracle_challenger_new.restype = C.c_void_p
        L.oracle_challenger_free.argtypes = [C.c_void_p]
        L.oracle_challenger_observe.argtypes = [C.c_void_p, U64P, C.c_size_t]
        L.oracle_challenger_get.argtypes = [C.c_void_p]
        L.oracle_challenger_get.restype = C.c_uint64
        L.oracle_challenger_compact.argtypes = [C.c_void_p]
        L.oracle_challenger_state.argtypes = [C.c_void_p, U64P]
        L.oracle_fri_reduction_arity_bits.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_fri_reduction_arity_bits.restype = C.c_int
        L.oracle_fri_pow.argtypes = [U64P, C.c_int]
        L.oracle_fri_pow.restype = C.c_uint64
        L.oracle_open_and_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                            U64P, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.oracle_open_and_prove.restype = C.c_size_t
        L.oracle_verify_opening.argtypes = [U64P, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                            C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.oracle_verify_opening.restype = C.c_int
        L.oracle_prove_with_traces.argtypes = [U64P, C.c_size_t, C.POINTER(U64P), C.POINTER(C.c_uint32), U64P, U64P,
                                               C.POINTER(C.c_int), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.oracle_prove_with_traces.restype = C.c_size_t
        L.oracle_verify_all_proof.argtypes = [U64P, C.c_size_t, U64P, C.POINTER(C.c_int), C.c_char_p, C.c_size_t, C.c_char_p,
                                              C.c_size_t]
        L.oracle_verify_all_proof.restype = C.c_int
        L.oracle_check_constraints.argtypes = [U64P, C.c_size_t, C.c_int, U64P, C.c_uint32, U64P]
        L.oracle_check_constraints.restype = C.c_long
        L.oracle_set_hasher.argtypes = [C.c_int]
        L.oracle_get_hasher.restype = C.c_int
        L.oracle_blake3.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.oracle_blake3_permutation.argtypes = [U64P]
        L.oracle_merkle_hash_leaf.argtypes = [U64P, C.c_size_t, U64P]
        L.oracle_merkle_two_to_one.argtypes = [U64P, U64P, U64P]
        L.oracle_challenger_observe_cap.argtypes = [C.c_void_p, U64P, C.c_size_t]
        L.oracle_ext_mul.argtypes = [U64P, U64P, U64P]
        L.oracle_ext_inv.argtypes = [U64P, U64P]

    # ---- field ----
    def vec_op(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.empty_like(a)
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
        self.lib.oracle_gl_vec_op({"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], ptr(a), None if bb is None else ptr(bb),
                                  ptr(out), a.size)
        return out

    def root_of_unity(self, k):
        return int(self.lib.oracle_root_of_unity(k))

    def pow(self, b, e):
        return int(self.lib.oracle_gl_pow(b, e))

    # ---- ntt (1-D arrays) ----
    def evaluate_poly(self, coeffs):
        v = np.array(coeffs, dtype=np.uint64)
        self.lib.oracle_evaluate_poly(ptr(v), v.size)
        return v

    def interpolate_poly(self, values):
        v = np.array(values, dtype=np.uint64)
        self.lib.oracle_interpolate_poly(ptr(v), v.size)
        return v

    def evaluate_poly_with_offset(self, coeffs, shift=7, blowup=8):
        c = np.ascontiguousarray(coeffs, dtype=np.uint64)
        out = np.empty(c.size * blowup, dtype=np.uint64)
        self.lib.oracle_evaluate_poly_with_offset(ptr(c), c.size, shift, blowup, ptr(out))
        return out

    def interpolate_poly_with_offset(self, values, shift=7):
        v = np.array(values, dtype=np.uint64)
        self.lib.oracle_interpolate_poly_with_offset(ptr(v), v.size, shift)
        return v

    def naive_eval(self, coeffs, domain, shift=1):
        c = np.ascontiguousarray(coeffs, dtype=np.uint64)
        out = np.empty(domain, dtype=np.uint64)
        self.lib.oracle_naive_eval(ptr(c), c.size, domain, shift, ptr(out))
        return out

    def eval_at_points(self, coeffs, points):
        """Horner value of the polynomial `coeffs` (low degree first) at each of `points`."""
        c = np.ascontiguousarray(coeffs, dtype=np.uint64)
        x = np.ascontiguousarray(points, dtype=np.uint64)
        out = np.empty(x.size, dtype=np.uint64)
        self.lib.oracle_eval_at_points(ptr(c), c.size, ptr(x), x.size, ptr(out))
        return out

    # ---- poseidon ----
    def permuted_cols(self, inputs, table):
        """lookup.rs:68-132 -> (sorted canonical inputs, permuted table)."""
        a = np.ascontiguousarray(inputs, dtype=np.uint64)
        b = np.ascontiguousarray(table, dtype=np.uint64)
        assert a.ndim == 1 and a.shape == b.shape
        pi, pt = np.empty_like(a), np.empty_like(a)
        self.lib.oracle_permuted_cols(ptr(a), ptr(b), a.shape[0], ptr(pi), ptr(pt))
        return pi, pt

    def poseidon(self, state):
        s = np.array(state, dtype=np.uint64)
        self.lib.oracle_poseidon(ptr(s))
        return s

    def hash_no_pad(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.empty(4, dtype=np.uint64)
        self.lib.oracle_hash_no_pad(ptr(d), d.size, ptr(out))
        return out

    def two_to_one(self, l, r):
        l = np.ascontiguousarray(l, dtype=np.uint64)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.empty(4, dtype=np.uint64)
        self.lib.oracle_two_to_one(ptr(l), ptr(r), ptr(out))
        return out

    # ---- the configuration's hasher: "poseidon" (PoseidonGoldilocksConfig) or "blake3" (Blake3GoldilocksConfig) ----
    HASHERS = {"poseidon": 0, "blake3": 1}

    def hasher(self, kind):
        """Context manager: Merkle trees, challengers, provers and verifiers of the oracle use `kind` inside the block.
        (Trace generation derives its compress challenges with a Poseidon challenger in both configurations --
        generation/builtin.rs:121, generation/prog.rs:24 -- so build the traces outside.)"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = self.lib.oracle_get_hasher()
            self.lib.oracle_set_hasher(self.HASHERS[kind])
            try:
                yield self
            finally:
                self.lib.oracle_set_hasher(old)
        return cm()

    def blake3(self, data: bytes) -> bytes:
        out = C.create_string_buffer(32)
        self.lib.oracle_blake3(data, len(data), out)
        return out.raw

    def blake3_permutation(self, state):
        s = np.array(state, dtype=np.uint64)
        self.lib.oracle_blake3_permutation(ptr(s))
        return s

    def merkle_hash_leaf(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.empty(4, dtype=np.uint64)
        self.lib.oracle_merkle_hash_leaf(ptr(d), d.size, ptr(out))
        return out

    def merkle_two_to_one(self, l, r):
        l = np.ascontiguousarray(l, dtype=np.uint64)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.empty(4, dtype=np.uint64)
        self.lib.oracle_merkle_two_to_one(ptr(l), ptr(r), ptr(out))
        return out

    # ---- merkle ----
    def merkle(self, leaves, cap_height, want_nodes=False):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64)
        n, w = lv.shape
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        lh = np.empty((n, 4), dtype=np.uint64)
        nodes = np.empty((max(n, 1), 4), dtype=np.uint64)
        self.lib.oracle_merkle(ptr(lv), n, w, cap_height, ptr(cap), ptr(lh), ptr(nodes))
        return (cap, lh, nodes) if want_nodes else cap

    def merkle_selfcheck(self, leaves, cap_height):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64)
        return self.lib.oracle_merkle_selfcheck(ptr(lv), lv.shape[0], lv.shape[1], cap_height)

    # ---- PolynomialBatch ----
    def batch(self, cols, rate_bits=3, cap_height=4, from_coeffs=False):
        return Batch(self, cols, rate_bits, cap_height, from_coeffs)

    def challenger(self):
        return OracleChallenger(self)

    def fri_arity_bits(self, degree_bits, cfg=None):
        out = (C.c_int * 16)()
        c = None if cfg is None else (C.c_int * 6)(*cfg)
        k = self.lib.oracle_fri_reduction_arity_bits(degree_bits, c, out)
        return list(out[:k])

    def fri_pow(self, h4, bits=16):
        h = np.ascontiguousarray(h4, dtype=np.uint64)
        return int(self.lib.oracle_fri_pow(ptr(h), bits))

    def open_and_prove(self, trace, zs, quot, num_perm_zs, challenger, cfg=None):
        c = None if cfg is None else (C.c_int * 6)(*cfg)
        zeta = np.empty(2, dtype=np.uint64)
        olen = C.c_size_t(0)
        probe = challenger.clone()  # keep alive across the sizing call
        need = self.lib.oracle_open_and_prove(trace.h, zs.h, quot.h, num_perm_zs, probe.h, c, ptr(zeta), None, 0,
                                              C.byref(olen))
        buf = C.create_string_buffer(need)
        got = self.lib.oracle_open_and_prove(trace.h, zs.h, quot.h, num_perm_zs, challenger.h, c, ptr(zeta), buf, need,
                                             C.byref(olen))
        assert got == need
        return zeta, bytes(buf.raw[:olen.value]), bytes(buf.raw[olen.value:need])

    def verify_opening(self, caps, num_polys, degree_bits, num_perm_zs, proof_bytes, challenger, cfg=None):
        c = None if cfg is None else (C.c_int * 6)(*cfg)
        caps = np.ascontiguousarray(caps, dtype=np.uint64)
        msg = C.create_string_buffer(256)
        rc = self.lib.oracle_verify_opening(ptr(caps), (C.c_int * 3)(*num_polys), degree_bits, num_perm_zs, proof_bytes,
                                            len(proof_bytes), challenger.h, c, msg, 256)
        return rc, msg.value.decode()


    # ---- multi-table STARK (oracle/stark.cpp) ----
    @staticmethod
    def _stark_args(blob, traces, params, compress):
        blob = np.ascontiguousarray(blob, dtype=np.uint64)
        tr = [np.ascontiguousarray(t, dtype=np.uint64) for t in traces]
        ptrs = (U64P * len(tr))(*[ptr(t) for t in tr])
        logs = (C.c_uint32 * len(tr))(*[int(t.shape[1]).bit_length() - 1 for t in tr])
        pr = np.ascontiguousarray(params if params is not None else [], dtype=np.uint64)
        cc = np.ascontiguousarray(compress if compress is not None else np.zeros(len(tr)), dtype=np.uint64)
        return blob, tr, ptrs, logs, pr, cc

    def prove_with_traces(self, blob, traces, params=None, compress=None, cfg=None):
        blob, tr, ptrs, logs, pr, cc = self._stark_args(blob, traces, params, compress)
        c = None if cfg is None else (C.c_int * 6)(*cfg)
        err = C.create_string_buffer(512)
        cap = 1 << 22
        while True:
            buf = C.create_string_buffer(cap)
            n = self.lib.oracle_prove_with_traces(ptr(blob), blob.size, ptrs, logs, ptr(pr) if pr.size else None, ptr(cc), c, buf,
                                                  cap, err, 512)
            if n == 0:
                raise RuntimeError("oracle prover: " + err.value.decode())
            if n <= cap:
                return bytes(buf.raw[:n])
            cap = n

    def verify_all_proof(self, blob, proof_bytes, params=None, cfg=None):
        blob = np.ascontiguousarray(blob, dtype=np.uint64)
        pr = np.ascontiguousarray(params if params is not None else [], dtype=np.uint64)
        c = None if cfg is None else (C.c_int * 6)(*cfg)
        msg = C.create_string_buffer(512)
        rc = self.lib.oracle_verify_all_proof(ptr(blob), blob.size, ptr(pr) if pr.size else None, c, proof_bytes, len(proof_bytes),
                                              msg, 512)
        return rc, msg.value.decode()

    def check_constraints(self, blob, table, trace, params=None):
        blob = np.ascontiguousarray(blob, dtype=np.uint64)
        tr = np.ascontiguousarray(trace, dtype=np.uint64)
        pr = np.ascontiguousarray(params if params is not None else [0], dtype=np.uint64)
        return int(self.lib.oracle_check_constraints(ptr(blob), blob.size, table, ptr(tr), int(tr.shape[1]).bit_length() - 1, ptr(pr)))


class Batch:
    def __init__(self, o, cols, rate_bits, cap_height, from_coeffs):
        self.o = o
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        self.ncols, self.n = cols.shape
        self.log_n = int(self.n).bit_length() - 1
        self.rate_bits, self.cap_height = rate_bits, cap_height
        f = o.lib.oracle_batch_from_coeffs if from_coeffs else o.lib.oracle_batch_from_values
        self.h = f(ptr(cols), self.ncols, self.log_n, rate_bits, cap_height)

    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        self.o.lib.oracle_batch_cap(self.h, ptr(out))
        return out

    def coeffs(self):
        out = np.empty((self.ncols, self.n), dtype=np.uint64)
        self.o.lib.oracle_batch_coeffs(self.h, ptr(out))
        return out

    def leaves(self):
        out = np.empty((self.n << self.rate_bits, self.ncols), dtype=np.uint64)
        self.o.lib.oracle_batch_leaves(self.h, ptr(out))
        return out

    def leaf(self, index):
        out = np.empty(self.ncols, dtype=np.uint64)
        self.o.lib.oracle_batch_leaf(self.h, index, ptr(out))
        return out

    def rehash(self):
        """Rebuild the Merkle tree over the same leaves under the hasher selected now (`with oracle.hasher(...)`)."""
        self.o.lib.oracle_batch_rehash(self.h)

    def prove(self, leaf):
        out = np.empty((64, 4), dtype=np.uint64)
        k = self.o.lib.oracle_batch_prove(self.h, leaf, ptr(out))
        return out[:k].copy()

    def __del__(self):
        try:
            self.o.lib.oracle_batch_free(self.h)
        except Exception:
            pass


class OracleChallenger:
    def __init__(self, o, h=None):
        self.o = o
        self.h = h if h is not None else o.lib.oracle_challenger_new()
        self.log = []  # replay log so the state can be cloned

    def observe(self, elems):
        e = np.ascontiguousarray(elems, dtype=np.uint64).ravel()
        self.log.append(("o", e.copy()))
        self.o.lib.oracle_challenger_observe(self.h, ptr(e), e.size)

    def observe_cap(self, digests):
        """observe_cap (challenger.rs:75-84): 4 elements per Poseidon digest, 5 (7-byte chunks) per Blake3 digest."""
        d = np.ascontiguousarray(digests, dtype=np.uint64).reshape(-1, 4)
        self.log.append(("h", d.copy()))
        self.o.lib.oracle_challenger_observe_cap(self.h, ptr(d), d.shape[0])

    def get(self):
        self.log.append(("g", None))
        return int(self.o.lib.oracle_challenger_get(self.h))

    def compact(self):
        self.log.append(("c", None))
        self.o.lib.oracle_challenger_compact(self.h)

    def state(self):
        out = np.empty(12, dtype=np.uint64)
        self.o.lib.oracle_challenger_state(self.h, ptr(out))
        return out

    def clone(self):
        c = OracleChallenger(self.o)
        for k, v in self.log:
            if k == "o":
                c.observe(v)
            elif k == "h":
                c.observe_cap(v)
            elif k == "g":
                c.get()
            else:
                c.compact()
        return c

    def __del__(self):
        try:
            self.o.lib.oracle_challenger_free(self.h)
        except Exception:
            pass


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(C.CDLL(build()))
    return _cached


def rand_field(rng, shape):
    """Uniform canonical field elements."""
    v = rng.integers(0, P, size=shape, dtype=np.uint64, endpoint=False)
    return v


EDGE = np.array([0, 1, P - 1, 2**32 - 1, 2**32, 0xFFFFFFFF00000000, P - 2, 2**63, 7], dtype=np.uint64)
