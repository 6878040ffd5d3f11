"""Writes tests/golden/blake3_vectors.json: BLAKE3 digests computed by the official C implementation, which LLVM carries
(llvm/lib/Support/BLAKE3, exported by libclang-cpp.so with an llvm_ prefix).  The crate the reference depends on (blake3 1.5.0,
Cargo.lock:220) is not vendored under /root/reference, so this independent implementation is what pins oracle/blake3.cpp and
the device kernels.  Inputs are the byte pattern of the official test vectors (byte i = i mod 251) at the official lengths, plus
the lengths this backend hashes (8 * leaf width for every commitment of the twelve tables, 64 for an inner node, 96 and 32 for
the challenger's onion).

    python tests/golden/make_blake3_vectors.py
"""
import ctypes as C
import json
import os

LIB = "/opt/rocm/lib/llvm/lib/libclang-cpp.so"
OFFICIAL_LENGTHS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2048, 2049, 3072, 3073, 4096, 4097,
                    5120, 5121, 6144, 6145, 7168, 7169, 8192, 8193, 16384, 31744, 102400]
BACKEND_LENGTHS = [16, 24, 32, 40, 96, 8 * 16, 8 * 94, 8 * 134, 8 * 76, 8 * 53, 8 * 127, 8 * 128, 8 * 129, 8 * 256, 8 * 257, 8 * 300, 8 * 513]


def hasher():
    lib = C.CDLL(LIB)
    lib.llvm_blake3_version.restype = C.c_char_p
    for f in ("llvm_blake3_hasher_init", "llvm_blake3_hasher_update", "llvm_blake3_hasher_finalize"):
        getattr(lib, f).restype = None
    lib.llvm_blake3_hasher_init.argtypes = [C.c_void_p]
    lib.llvm_blake3_hasher_update.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.llvm_blake3_hasher_finalize.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    state = C.create_string_buffer(4096)     # sizeof(blake3_hasher) = 1912

    def h(data: bytes) -> bytes:
        lib.llvm_blake3_hasher_init(state)
        lib.llvm_blake3_hasher_update(state, data, len(data))
        out = C.create_string_buffer(32)
        lib.llvm_blake3_hasher_finalize(state, out, 32)
        return out.raw
    return h, lib.llvm_blake3_version().decode()


def main():
    h, version = hasher()
    pattern = bytes(i % 251 for i in range(max(OFFICIAL_LENGTHS) + 1))
    # two digests everybody can check by eye against the BLAKE3 README / b3sum
    assert h(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    vectors = [{"len": n, "hash": h(pattern[:n]).hex()} for n in sorted(set(OFFICIAL_LENGTHS + BACKEND_LENGTHS))]
    out = {"source": "official BLAKE3 C implementation bundled with LLVM (%s), version %s" % (os.path.basename(LIB), version),
           "input": "byte i = i mod 251", "vectors": vectors,
           "text": [{"ascii": "hello world", "hash": h(b"hello world").hex()}]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blake3_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(vectors), "vectors; BLAKE3", version)


if __name__ == "__main__":
    main()
