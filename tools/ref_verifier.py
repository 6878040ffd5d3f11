#!/usr/bin/env python3
"""The reference's VERIFIER, run from its source on a proof of this repository's provers.

The reference is Rust and this image has no Rust toolchain, so no binary of it can judge a proof here.  tools/rust_air_eval.py
interprets the subset of Rust the reference is written in; this driver hands an AllProof (the bytes `ola_prove_with_traces` /
the oracle prover return, reference wire format) to
  * `Buffer::write_all_proof` (circuits/src/stark/serialization.rs:377): the decoded proof, re-encoded by the reference's own
    writer, must give back the bytes;
  * `verify_proof` (circuits/src/stark/verifier.rs:35): `AllProof::get_challenges` (get_challenges.rs:18, the whole Fiat-Shamir
    transcript), `CtlCheckVars::from_proofs`, per table `verify_stark_proof_with_challenges` (proof shape, `eval_vanishing_poly`
    of the table's AIR over the extension field at zeta, the quotient identity, `verify_fri_proof` of plonky2 with its Merkle
    paths, reduced openings, folding and final polynomial), and `verify_cross_table_lookups`.
The only pieces not interpreted are the Goldilocks / quadratic-extension arithmetic (Fe, Fe2 of rust_air_eval.py) and, for speed,
the Poseidon permutation inside Merkle paths: a direct Python permutation that is first checked against the interpreted
`Poseidon::poseidon_naive` on fresh inputs.

    python tools/ref_verifier.py PROOF [--reference /root/reference]      (exit status 0 = the reference's verifier returned Ok(()))
"""
import argparse
import os
import re
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rust_air_eval as R  # noqa: E402
from rust_air_eval import Enum, Fe, Fe2, P, Struct  # noqa: E402


# ------------------------------------------------------------------------------------------------------------------------------
# the proof as the structs of circuits/src/stark/proof.rs and plonky2 fri/proof.rs
def hash_out(raw, off):
    return Struct({"__name__": "HashOut", "elements": [Fe(x) for x in struct.unpack_from("<4Q", raw, off)]})


def decode_all_proof(raw, hasher="poseidon"):
    """bytes -> AllProof { stark_proofs: [StarkProof; 12], compress_challenges }.  Canonical encodings only: a word >= p is an error
    (the reference's reader would reduce it; a prover must not emit it)."""
    off = 0

    def u32():
        nonlocal off
        (v,) = struct.unpack_from("<I", raw, off)
        off += 4
        return v

    def u8():
        nonlocal off
        v = raw[off]
        off += 1
        return v

    def field():
        nonlocal off
        (v,) = struct.unpack_from("<Q", raw, off)
        off += 8
        if v >= P:
            raise ValueError(f"non-canonical field element at byte {off - 8}")
        return Fe(v)

    def ext():
        a = field()
        b = field()
        return Fe2(a.v, b.v)

    def digest():
        nonlocal off
        h = hash_out(raw, off) if hasher == "poseidon" else bytes_hash(raw[off:off + 32])
        if hasher == "poseidon" and any(x.v != w for x, w in zip(h["elements"], struct.unpack_from("<4Q", raw, off))):
            raise ValueError(f"non-canonical digest word at byte {off}")
        off += 32
        return h

    def cap():
        return Struct({"__name__": "MerkleCap", 0: [digest() for _ in range(u32())]})

    def merkle_proof():
        return Struct({"__name__": "MerkleProof", "siblings": [digest() for _ in range(u8())]})

    def ext_vec():
        return [ext() for _ in range(u32())]

    def field_vec():
        return [field() for _ in range(u32())]

    proofs = []
    for _ in range(u32()):
        trace_cap, zs_cap, quot_cap = cap(), cap(), cap()
        openings = Struct({"__name__": "StarkOpeningSet", "local_values": ext_vec(), "next_values": ext_vec(), "permutation_ctl_zs": ext_vec(),
                           "permutation_ctl_zs_next": ext_vec(), "ctl_zs_last": field_vec(), "quotient_polys": ext_vec()})
        caps = [cap() for _ in range(u32())]
        rounds = []
        for _ in range(u32()):
            evals_proofs = [(field_vec(), merkle_proof()) for _ in range(u32())]
            steps = []
            for _ in range(u32()):
                evals = [ext() for _ in range(u32())]
                steps.append(Struct({"__name__": "FriQueryStep", "evals": evals, "merkle_proof": merkle_proof()}))
            rounds.append(Struct({"__name__": "FriQueryRound", "initial_trees_proof": Struct({"__name__": "FriInitialTreeProof", "evals_proofs": evals_proofs}),
                                  "steps": steps}))
        final_poly = Struct({"__name__": "PolynomialCoeffs", "coeffs": ext_vec()})
        pow_witness = field()
        fri = Struct({"__name__": "FriProof", "commit_phase_merkle_caps": caps, "query_round_proofs": rounds, "final_poly": final_poly, "pow_witness": pow_witness})
        proofs.append(Struct({"__name__": "StarkProof", "trace_cap": trace_cap, "permutation_ctl_zs_cap": zs_cap, "quotient_polys_cap": quot_cap,
                              "openings": openings, "opening_proof": fri}))
    compress = field_vec()
    if off != len(raw):
        raise ValueError(f"{len(raw) - off} trailing bytes")
    return Struct({"__name__": "AllProof", "stark_proofs": proofs, "compress_challenges": compress, "public_values": Struct({"__name__": "PublicValues"})})


# ------------------------------------------------------------------------------------------------------------------------------
def poseidon_tables(reference):
    """MDS rows and round constants read from the reference's poseidon_goldilocks.rs / poseidon.rs through the interpreter's constant lookup"""
    it = R.plonky2_interp(reference)
    src = R.X.Src.get(os.path.join(it.plonky2, "hash", "poseidon.rs"))
    gsrc = R.X.Src.get(os.path.join(it.plonky2, "hash", "poseidon_goldilocks.rs"))
    circ = it.const_value("MDS_MATRIX_CIRC", gsrc)
    diag = it.const_value("MDS_MATRIX_DIAG", gsrc)
    rc = it.const_value("ALL_ROUND_CONSTANTS", src)
    return it, [int(x) for x in circ], [int(x) for x in diag], [int(x) for x in rc]


class FastPoseidon:
    """x^7 S-box, the circulant + diagonal MDS layer, 4 + 22 + 4 rounds: the permutation of poseidon.rs written directly, for the tens of
    thousands of Merkle-path hashes of a verification.  `check` compares it with the interpreted `poseidon_naive`."""

    def __init__(self, circ, diag, rc):
        self.circ, self.diag, self.rc = circ, diag, rc
        # row r of the MDS matrix: out[r] = sum_i state[(i + r) % 12] * circ[i] + state[r] * diag[r]
        self.rows = [[circ[(j - r) % 12] + (diag[r] if j == r else 0) for j in range(12)] for r in range(12)]

    def permute(self, state):
        s = [x.v for x in state]
        rows, rc = self.rows, self.rc
        r = 0
        for phase, rounds in (("full", 4), ("partial", 22), ("full", 4)):
            for _ in range(rounds):
                k = 12 * r
                if phase == "full":
                    s = [pow(s[i] + rc[k + i], 7, P) for i in range(12)]
                else:
                    s = [s[i] + rc[k + i] for i in range(12)]
                    s[0] = pow(s[0], 7, P)
                s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11 = s
                s = [(m[0] * s0 + m[1] * s1 + m[2] * s2 + m[3] * s3 + m[4] * s4 + m[5] * s5 + m[6] * s6 + m[7] * s7 + m[8] * s8 + m[9] * s9
                      + m[10] * s10 + m[11] * s11) % P for m in rows]
                r += 1
        return [Fe(x) for x in s]

    def check(self, it, count=6):
        pos = os.path.join(it.plonky2, "hash", "poseidon.rs")
        for k in range(count):
            v = [Fe(x) for x in ([P - 1] * 12 if k == 0 else R.stream_for(7700 + k, 9, 12))]
            want = it.call_assoc("Poseidon", "poseidon_naive", [list(v)], pos)
            if [x.v for x in want] != [x.v for x in self.permute(v)]:
                raise SystemExit("the direct Poseidon permutation disagrees with the interpreted poseidon_naive")


# ------------------------------------------------------------------------------------------------------------------------------
# BLAKE3 (the `blake3` crate the reference's Blake3_256 hasher calls is not part of the reference tree): the published algorithm, hash mode,
# any length; checked against tests/golden/blake3_vectors.json (the official C implementation's outputs) before use
B3_IV = (0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19)
B3_PERM = (2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
M32 = 0xFFFFFFFF


def _b3_compress(cv, block_words, counter, block_len, flags):
    v = list(cv) + [B3_IV[0], B3_IV[1], B3_IV[2], B3_IV[3], counter & M32, (counter >> 32) & M32, block_len, flags]
    m = list(block_words)

    def g(a, b, c, d, x, y):
        va, vb, vc, vd = v[a], v[b], v[c], v[d]
        va = (va + vb + x) & M32
        vd ^= va
        vd = ((vd >> 16) | (vd << 16)) & M32
        vc = (vc + vd) & M32
        vb ^= vc
        vb = ((vb >> 12) | (vb << 20)) & M32
        va = (va + vb + y) & M32
        vd ^= va
        vd = ((vd >> 8) | (vd << 24)) & M32
        vc = (vc + vd) & M32
        vb ^= vc
        vb = ((vb >> 7) | (vb << 25)) & M32
        v[a], v[b], v[c], v[d] = va, vb, vc, vd

    for r in range(7):
        g(0, 4, 8, 12, m[0], m[1])
        g(1, 5, 9, 13, m[2], m[3])
        g(2, 6, 10, 14, m[4], m[5])
        g(3, 7, 11, 15, m[6], m[7])
        g(0, 5, 10, 15, m[8], m[9])
        g(1, 6, 11, 12, m[10], m[11])
        g(2, 7, 8, 13, m[12], m[13])
        g(3, 4, 9, 14, m[14], m[15])
        if r < 6:
            m = [m[i] for i in B3_PERM]
    return [v[i] ^ v[i + 8] for i in range(8)]


def blake3(data):
    """32-byte BLAKE3 hash of `data`"""
    CHUNK_START, CHUNK_END, PARENT, ROOT = 1, 2, 4, 8
    data = bytes(data)
    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]

    def chunk_cv(chunk, counter, root):
        cv = list(B3_IV)
        blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
        for i, blk in enumerate(blocks):
            flags = (CHUNK_START if i == 0 else 0) | (CHUNK_END if i == len(blocks) - 1 else 0)
            if root and i == len(blocks) - 1:
                flags |= ROOT
            words = struct.unpack("<16I", blk.ljust(64, b"\0"))
            cv = _b3_compress(cv, words, counter, len(blk), flags)
        return cv

    if len(chunks) == 1:
        return struct.pack("<8I", *chunk_cv(chunks[0], 0, True))
    cvs = [chunk_cv(c, i, False) for i, c in enumerate(chunks)]

    def merge(nodes, root):
        # left subtree: the largest power of two of chunks strictly less than the count
        if len(nodes) == 1:
            return nodes[0]
        left = 1 << ((len(nodes) - 1).bit_length() - 1)
        l, r = merge(nodes[:left], False), merge(nodes[left:], False)
        return _b3_compress(B3_IV, l + r, 0, 64, PARENT | (ROOT if root else 0))

    return struct.pack("<8I", *merge(cvs, True))


def check_blake3():
    import json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "blake3_vectors.json")))
    for v in fx["vectors"]:
        n = int(v["len"])
        if blake3(bytes(i % 251 for i in range(n))).hex() != v["hash"]:
            raise SystemExit("BLAKE3 restatement fails the official vector of length %d" % n)
    for t in fx["text"]:
        if blake3(t["ascii"].encode()).hex() != t["hash"]:
            raise SystemExit("BLAKE3 restatement fails the official text vector")


def bytes_hash(b):
    return Struct({"__name__": "BytesHash", 0: [R.TInt(x, 8) for x in b]})


def blake3_hooks(it):
    """`Blake3_256::<32>::hash_no_pad / two_to_one` and `Blake3Permutation::permute` (hash/blake3.rs:167-235): the three functions of the
    reference that call the external crate, restated.  hash_no_pad hashes the elements' memory -- 8 little-endian bytes each (a proof's
    elements are canonical); two_to_one hashes left || right; the permutation is the 'hash onion': hash the 96 state bytes, then keep hashing
    the previous output, cut the outputs into little-endian u64 words, drop words >= p, take twelve."""
    def hash_no_pad(args):
        return bytes_hash(blake3(b"".join(struct.pack("<Q", x.v) for x in args[0])))

    def two_to_one(args):
        return bytes_hash(blake3(bytes(args[0][0]) + bytes(args[1][0])))

    def permute(state):
        cur = b"".join(struct.pack("<Q", x.v) for x in state)
        out = []
        while len(out) < 12:
            cur = blake3(cur)
            for w in struct.unpack("<4Q", cur):
                if w < P and len(out) < 12:
                    out.append(Fe(w))
        return out

    it.assoc_hooks[("Blake3_256", "hash_no_pad")] = hash_no_pad
    it.assoc_hooks[("Blake3_256", "two_to_one")] = two_to_one
    return permute


def verifier_interp(reference, fast_hash=True, hasher="poseidon"):
    it, circ, diag, rc = poseidon_tables(reference)
    base = it.plonky2
    it.extra_files += [os.path.join(base, p) for p in (
        "fri/verifier.rs", "fri/challenges.rs", "fri/structure.rs", "fri/proof.rs", "fri/validate_shape.rs", "util/reducing.rs", "plonk/plonk_common.rs",
        "hash/hash_types.rs", "hash/merkle_tree/mod.rs", "util/mod.rs")]
    it.extra_files += [os.path.join(reference, "plonky2", p) for p in (
        "util/src/lib.rs", "field/src/interpolation.rs", "field/src/polynomial/mod.rs", "field/src/extension/mod.rs")]
    it.generics.update({"C": ["PoseidonGoldilocksConfig"], "InnerHasher": ["PoseidonHash", "Hasher"], "Hasher": ["PoseidonHash", "Hasher"]})
    it.extension = True
    gf = open(os.path.join(reference, "plonky2", "field", "src", "goldilocks_field.rs")).read()
    for name in ("TWO_ADICITY", "POWER_OF_TWO_GENERATOR", "MULTIPLICATIVE_GROUP_GENERATOR"):
        m = re.search(r"const %s: \w+ = (?:Self\()?(\d+)\)?;" % name, gf)
        it.field_consts[name] = int(m.group(1))
    if re.search(r"const W: Self = Self\((\d+)\);", open(os.path.join(reference, "plonky2", "field", "src", "goldilocks_extensions.rs")).read()).group(1) != str(Fe2.W):
        raise SystemExit("the quadratic extension's W is not 7")
    # Hasher::zero_hash (plonk/config.rs:72-78): HASH_SIZE zero bytes through Hash::from_bytes
    it.assoc_hooks[("Hasher", "zero_hash")] = (lambda args: hash_out(bytes(32), 0)) if hasher == "poseidon" else (lambda args: bytes_hash(bytes(32)))
    poseidon = it.permutation_hook
    if fast_hash:
        fp = FastPoseidon(circ, diag, rc)
        fp.check(it)
        poseidon = lambda st, segs=None: fp.permute(st)        # noqa: E731
    it.permutation_hook = poseidon
    if hasher == "blake3":
        # Blake3GoldilocksConfig (plonk/config.rs:155-161): Hasher = Blake3_256<32>, InnerHasher = PoseidonHash (the proof of work)
        check_blake3()
        onion = blake3_hooks(it)
        it.extra_files.append(os.path.join(base, "hash", "blake3.rs"))
        it.generics.update({"C": ["Blake3GoldilocksConfig"], "Hasher": ["Blake3_256", "Hasher"], "H": ["Blake3_256", "Hasher"], "OH": ["Blake3_256", "Hasher"],
                            "Hash": ["BytesHash"]})
        # `H::Permutation::permute` (the challenger's sponge) is the config hasher's; `P::permute` inside hashing.rs is PoseidonHash's own
        it.permutation_hook = lambda st, segs=None: onion(st) if (segs and "Permutation" in segs) else poseidon(st)
    elif hasher != "poseidon":
        raise ValueError(hasher)
    return it


STARK_FIELDS = ["cpu_stark", "memory_stark", "bitwise_stark", "cmp_stark", "rangecheck_stark", "poseidon_stark", "poseidon_chunk_stark", "storage_access_stark",
                "tape_stark", "sccall_stark", "program_stark", "prog_chunk_stark"]


class RefVerifier:
    """The reference's verifier over one reference tree.  `verify(raw)` -> (True, None) or (False, "file.rs:line" of the `ensure!` that failed)."""

    def __init__(self, reference="/root/reference", fast_hash=True, hasher="poseidon"):
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
        self.reference = reference
        self.hasher = hasher
        self.it = verifier_interp(reference, fast_hash, hasher)
        self.stark_dir = os.path.join(reference, "circuits", "src", "stark")
        self.config = self.it.call_assoc("StarkConfig", "standard_fast_config", [], os.path.join(self.stark_dir, "config.rs"))

    def ola_stark(self):
        """what `OlaStark::default()` (ola_stark.rs:45-64) builds, minus its init_gpu() call: twelve default table structs (compress challenges unset,
        as `#[derive(Default)]` leaves them) and `all_cross_table_lookups()`"""
        ola = Struct({"__name__": "OlaStark"})
        for f, (_, rel) in zip(STARK_FIELDS, R.X.TABLES):
            st = R.stark_struct(self.it, self.it.ref, rel)
            st["compress_challenge"] = None
            ola[f] = st
        ola["cross_table_lookups"] = self.it.call_fn(os.path.join(self.stark_dir, "ola_stark.rs"), "all_cross_table_lookups", [])
        return ola

    @staticmethod
    def _verdict(r):
        if isinstance(r, Enum):
            if r.variant != "Err":
                raise R.RustError("unexpected result " + r.variant)
            return False, r.payload[0] if r.payload else "?"
        return True, None

    def verify(self, raw):
        """`verify_proof(OlaStark::default(), all_proof, &StarkConfig::standard_fast_config())`, verifier.rs:35"""
        proof = raw if isinstance(raw, Struct) else decode_all_proof(raw, self.hasher)
        try:
            r = self.it.call_free(os.path.join(self.stark_dir, "verifier.rs"), "verify_proof", [self.ola_stark(), proof, self.config])
        except ZeroDivisionError as e:           # an inverse of zero: Rust would panic
            return False, "panic: %s" % e
        return self._verdict(r)

    def encode(self, proof):
        """`Buffer::write_all_proof` (serialization.rs:377) of a decoded proof -> bytes"""
        buf = Struct({"__name__": "Buffer", 0: R.Cursor()})
        r = self.it.call_assoc("Buffer", "write_all_proof", [buf, proof], os.path.join(self.stark_dir, "serialization.rs"))
        if isinstance(r, Enum):
            raise R.RustError("write_all_proof returned " + r.variant)
        return bytes(buf[0].data)

    def challenges(self, raw):
        """`AllProof::get_challenges` (get_challenges.rs:18): every Fiat-Shamir challenge of the proof, as plain integers"""
        proof = raw if isinstance(raw, Struct) else decode_all_proof(raw, self.hasher)
        ch = self.it.call_assoc("AllProof", "get_challenges", [proof, self.ola_stark(), self.config], os.path.join(self.stark_dir, "get_challenges.rs"))

        def plain(v):
            if isinstance(v, R.Opt):
                return plain(v.v)
            if isinstance(v, Fe):
                return v.v
            if isinstance(v, Fe2):
                return [v.a, v.b]
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items() if k != "__name__"}
            if isinstance(v, (list, tuple)):
                return [plain(x) for x in v]
            return v
        return plain(ch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("proof")
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--hasher", default="poseidon", choices=("poseidon", "blake3"), help="PoseidonGoldilocksConfig or Blake3GoldilocksConfig")
    a = ap.parse_args()
    raw = open(a.proof, "rb").read()
    proof = decode_all_proof(raw, a.hasher)
    print("decoded: %d tables, %d bytes" % (len(proof["stark_proofs"]), len(raw)))
    rv = RefVerifier(a.reference, hasher=a.hasher)
    if rv.encode(proof) != raw:
        raise SystemExit("write_all_proof of the decoded proof does not give back the bytes")
    print("write_all_proof (serialization.rs:377), interpreted, reproduces the %d bytes" % len(raw))
    ok, where = rv.verify(proof)
    print("verify_proof (verifier.rs:35), interpreted: %s" % ("Ok(())" if ok else "Err at " + where))
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()


# ------------------------------------------------------------------------------------------------------------------------------
class RefProver(RefVerifier):
    """The reference's PROVER for one table, run from its source: `prove_single_table` (circuits/src/stark/prover.rs:330-567) with everything
    under it -- `compute_permutation_z_polys`, `PolynomialBatch::from_values / from_coeffs` (ifft, the x8 coset LDE of cfft, transpose,
    bit-reversal, `MerkleTree::new_v2`), `compute_quotient_polys`, `StarkOpeningSet::new`, `PolynomialBatch::prove_openings`, `fri_proof`
    (commit phase, proof of work, query rounds) -- and `cross_table_lookup_data` for the table's lookup Z columns.  Only tables small enough for
    an interpreter (up to a few hundred rows); the transcript state before the table comes from the reference's own
    `AllProof::get_challenger_states` over the whole proof.  Where the reference leaves a choice, the driver takes the one this repository's
    provers take: packing width 1 (`<F as Packable>::Packing` is a SIMD width, not a value) and the SMALLEST proof-of-work witness
    (`find_any` over 0..p may return any)."""

    def __init__(self, reference="/root/reference", hasher="poseidon"):
        super().__init__(reference, True, hasher)
        it, ref = self.it, reference
        cf = os.path.join(ref, "plonky2", "field", "src")
        pl = os.path.join(ref, "plonky2", "plonky2", "src")
        it.extra_files += [os.path.join(cf, "cfft", "mod.rs"), os.path.join(cf, "cfft", "serial.rs"), os.path.join(cf, "cfft", "concurrent.rs"),
                           os.path.join(cf, "polynomial", "division.rs"), os.path.join(cf, "zero_poly_coset.rs"), os.path.join(cf, "types.rs"),
                           os.path.join(pl, "fri", "oracle.rs"), os.path.join(pl, "fri", "prover.rs"), os.path.join(pl, "hash", "merkle_tree", "concurrent.rs")]
        it.features = {"parallel"}
        it.packing_width = 1
        oracle_rs = os.path.join(pl, "fri", "oracle.rs")

        def get_lde_values_packed(args):
            # fri/oracle.rs:138-161 for P::WIDTH = 1: the one row get_lde_values returns, each element its own 'pack'
            return list(it.call_assoc("PolynomialBatch", "get_lde_values", [args[0], args[1], args[2]], oracle_rs))
        it.assoc_hooks[("PolynomialBatch", "get_lde_values_packed")] = get_lde_values_packed
        zpc = os.path.join(cf, "zero_poly_coset.rs")
        # zero_poly_coset.rs:45-53 for P::WIDTH = 1
        it.assoc_hooks[("ZeroPolyOnCoset", "eval_inverse_packed")] = lambda args: it.call_assoc("ZeroPolyOnCoset", "eval_inverse", [args[0], args[1]], zpc)
        # plonky2_field batch_util.rs:29-65: the SIMD-packed elementwise product / sum of two slices, in place -- for a packing of width 1
        def batch_multiply_inplace(args):
            out, a = args
            if len(out) != len(a):
                raise R.RustError("both arrays must have the same length")
            out[:] = [it.binop("*", x, y, None, 0) for x, y in zip(out, a)]

        def batch_add_inplace(args):
            out, a = args
            if len(out) != len(a):
                raise R.RustError("both arrays must have the same length")
            out[:] = [it.binop("+", x, y, None, 0) for x, y in zip(out, a)]
        it.fn_hooks["batch_multiply_inplace"] = batch_multiply_inplace
        it.fn_hooks["batch_add_inplace"] = batch_add_inplace
        it.generics["F"] = ["Field"]                # the Field trait's provided functions (plonky2_field types.rs): two_adic_subgroup, cyclic_subgroup_*

    def base_field(self):
        """the prover's generic code is instantiated with FE = F, D2 = 1 (prover.rs:669, cross_table_lookup.rs:236): `FE::from_basefield` is the
        identity there, while `F::Extension::from_basefield` stays what it says"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.it.extension = False
            try:
                yield
            finally:
                self.it.extension = True
        return cm()

    def challenger_at(self, state):
        return Struct({"__name__": "Challenger", "sponge_state": [Fe(x.v) for x in state], "input_buffer": [], "output_buffer": [], "_phantom": None})

    def prove_table(self, raw, traces, k, max_rows_for_lookups=4096, full_pow_search=True):
        """the bytes `Buffer::write_proof` gives for the reference's prove_single_table of table k, given the whole proof (for the transcript
        state before the table) and the twelve traces"""
        it = self.it
        proof = decode_all_proof(raw, self.hasher)
        ola = self.ola_stark()
        # prove_with_traces sets the two compress challenges before anything else (prover.rs:84-100)
        ola["bitwise_stark"]["compress_challenge"] = proof["compress_challenges"][2]
        ola["program_stark"]["compress_challenge"] = proof["compress_challenges"][10]
        states = it.call_assoc("AllProof", "get_challenger_states", [proof, ola, self.config], os.path.join(self.stark_dir, "get_challenges.rs"))
        # lookup Z columns: cross_table_lookup_data over the traces; tables too long to interpret are cut to 8 rows (their own Z columns are
        # then meaningless and not used: a table's Z depends on its own rows and the challenges only)
        values = []
        for t in traces:
            n = t.shape[1] if t.shape[1] <= max_rows_for_lookups else 8
            values.append([Struct({"__name__": "PolynomialValues", "values": [Fe(int(x)) for x in col[:n]]}) for col in t])
        ch = self.it.call_assoc("Challenger", "new", [], os.path.join(it.plonky2, "iop", "challenger.rs"))
        for sp in proof["stark_proofs"]:
            it.call_assoc("Challenger", "observe_cap", [ch, sp["trace_cap"]], os.path.join(it.plonky2, "iop", "challenger.rs"))
        with self.base_field():
            ctl = it.call_free(os.path.join(self.stark_dir, "cross_table_lookup.rs"), "cross_table_lookup_data", [self.config, values, ola["cross_table_lookups"], ch])
        if [x.v for x in it.call_assoc("Challenger", "compact", [ch], os.path.join(it.plonky2, "iop", "challenger.rs"))] != [x.v for x in states["states"][0]]:
            raise SystemExit("the transcript before the first table differs between prover side and verifier side")
        stark = ola[STARK_FIELDS[k]]
        twiddles = {}
        commitment = it.call_assoc("PolynomialBatch", "from_values", [clone_values(values[k]), self.config["fri_config"]["rate_bits"], False,
                                                                      self.config["fri_config"]["cap_height"], None, twiddles],
                                   os.path.join(it.plonky2, "fri", "oracle.rs"))
        challenger = self.challenger_at(states["states"][k])
        # the proof-of-work search (fri/prover.rs:126-148) from 0 costs a minute or two interpreted; `full_pow_search=False` checks the proof's
        # witness and 64 smaller candidates instead
        it.find_any_hint = None if full_pow_search else proof["stark_proofs"][k]["opening_proof"]["pow_witness"].v
        try:
            with self.base_field():
                r = it.call_free(os.path.join(self.stark_dir, "prover.rs"), "prove_single_table",
                                 [stark, self.config, values[k], commitment, ctl[k], challenger, None, twiddles])
        finally:
            it.find_any_hint = None
        if isinstance(r, Enum):
            raise R.RustError("prove_single_table returned Err " + str(r.payload))
        buf = Struct({"__name__": "Buffer", 0: R.Cursor()})
        it.call_assoc("Buffer", "write_proof", [buf, r], os.path.join(self.stark_dir, "serialization.rs"))
        after = it.call_assoc("Challenger", "compact", [challenger], os.path.join(it.plonky2, "iop", "challenger.rs"))
        return bytes(buf[0].data), [x.v for x in after] == [x.v for x in states["states"][k + 1]]


def clone_values(vs):
    return [Struct({"__name__": "PolynomialValues", "values": list(v["values"])}) for v in vs]


def table_span(raw, k):
    """byte range of table k's StarkProof inside an AllProof"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration", "pin"))
    import compare_with_dump as CD
    spans = [(n, a, b) for n, a, b in CD.parse_all_proof(raw) if n.startswith("table %d:" % k)]
    return spans[0][1] - 4, spans[-1][2]          # the cap's u32 count precedes its span
