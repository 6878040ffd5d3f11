// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of the hash side of the reference's
// Blake3GoldilocksConfig (plonky2/plonky2/src/plonk/config.rs:153-161: Hasher = Blake3_256<32>, InnerHasher = Poseidon).
//
// The hash function itself lives in a dependency that is not vendored under /root/reference: crate `blake3` 1.5.0
// (Cargo.lock:220-223; plonky2/plonky2/Cargo.toml:36 asks for 1.3.3).  What follows restates the published BLAKE3 algorithm
// (the BLAKE3 paper, section 2: chunks of 1024 bytes, blocks of 64, 7-round compression function, binary tree of chaining
// values whose left subtree holds the largest power of two of chunks); the reference's own in-field copy of the compression
// function gives the same IV and message schedule (hash/blake3_goldilocks.rs:12-27, hash/blake3.rs:25-131).  It is pinned
// by tests/golden/blake3_vectors.json, produced with the official C implementation that LLVM bundles
// (tests/golden/make_blake3_vectors.py).
//
// Follows (relative to /root/reference/plonky2/plonky2/src):
//   hash/blake3.rs:203-213    Blake3_256::hash_no_pad = blake3::hash(bytes of the field slice)   [canonical words here: the
//                             reference hashes the words as they lie in memory, its verifier hashes deserialised -- canonical --
//                             words (hash/merkle_proofs.rs:52-80), and those are the ones a proof has to be consistent with]
//   hash/blake3.rs:215-233    two_to_one = blake3::hash(left || right)
//   hash/blake3.rs:166-201    Blake3Permutation: the "onion" h1 = H(state bytes), h2 = H(h1), ...; little-endian u64 words,
//                             words >= p rejected, the first 12 kept
//   hash/hash_types.rs:142-152  BytesHash::to_vec: chunks of 7 bytes -> 5 field elements per digest (what a challenger observes)
#include <cstring>

#include "oracle.hpp"

namespace ola_oracle {

int g_hasher = HASH_POSEIDON;
void set_hasher(int kind) { g_hasher = kind; }
int get_hasher() { return g_hasher; }

namespace {
const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
const int B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline void g(uint32_t* s, int a, int b, int c, int d, uint32_t x, uint32_t y) {
    s[a] = s[a] + s[b] + x; s[d] = rotr(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];     s[b] = rotr(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + y; s[d] = rotr(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];     s[b] = rotr(s[b] ^ s[c], 7);
}
// new chaining value = first 8 words of the output (the only part this configuration ever uses)
void compress(const uint32_t cv[8], const uint8_t block[64], uint64_t counter, uint32_t block_len, uint32_t flags, uint32_t out[8]) {
    uint32_t m[16], s[16];
    for (int i = 0; i < 16; i++) m[i] = (uint32_t)block[4 * i] | (uint32_t)block[4 * i + 1] << 8 | (uint32_t)block[4 * i + 2] << 16 | (uint32_t)block[4 * i + 3] << 24;
    for (int i = 0; i < 8; i++) s[i] = cv[i];
    for (int i = 0; i < 4; i++) s[8 + i] = B3_IV[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = block_len; s[15] = flags;
    for (int r = 0; r < 7; r++) {
        g(s, 0, 4, 8, 12, m[0], m[1]);  g(s, 1, 5, 9, 13, m[2], m[3]);  g(s, 2, 6, 10, 14, m[4], m[5]);   g(s, 3, 7, 11, 15, m[6], m[7]);
        g(s, 0, 5, 10, 15, m[8], m[9]); g(s, 1, 6, 11, 12, m[10], m[11]); g(s, 2, 7, 8, 13, m[12], m[13]); g(s, 3, 4, 9, 14, m[14], m[15]);
        uint32_t t[16];
        for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
        memcpy(m, t, sizeof m);
    }
    for (int i = 0; i < 8; i++) out[i] = s[i] ^ s[i + 8];
}
// chaining value of chunk `index` (its bytes [p, p + len), len in 0..1024); `root` when the whole input is this one chunk
void chunk_cv(const uint8_t* p, size_t len, uint64_t index, bool root, uint32_t out[8]) {
    uint32_t cv[8];
    memcpy(cv, B3_IV, sizeof cv);
    const size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < nblocks; b++) {
        uint8_t block[64] = {0};
        const size_t bl = (b + 1 == nblocks) ? len - 64 * b : 64;
        memcpy(block, p + 64 * b, bl);
        uint32_t flags = (b == 0 ? CHUNK_START : 0) | (b + 1 == nblocks ? CHUNK_END | (root ? ROOT : 0) : 0);
        compress(cv, block, index, (uint32_t)bl, flags, cv);
    }
    memcpy(out, cv, sizeof cv);
}
void parent_cv(const uint32_t l[8], const uint32_t r[8], bool root, uint32_t out[8]) {
    uint8_t block[64];
    for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) { block[4 * i + k] = (uint8_t)(l[i] >> (8 * k)); block[32 + 4 * i + k] = (uint8_t)(r[i] >> (8 * k)); }
    compress(B3_IV, block, 0, 64, PARENT | (root ? ROOT : 0), out);
}
// chaining value of the subtree over chunks [first, first + count) of the input; count >= 1
void subtree_cv(const uint8_t* in, size_t len, uint64_t first, uint64_t count, bool root, uint32_t out[8]) {
    if (count == 1) {
        const size_t off = (size_t)first * 1024;
        chunk_cv(in + off, len - off < 1024 ? len - off : 1024, first, root, out);
        return;
    }
    uint64_t left = 1;
    while (2 * left < count) left *= 2;   // largest power of two strictly below count
    uint32_t l[8], r[8];
    subtree_cv(in, len, first, left, false, l);
    subtree_cv(in, len, first + left, count - left, false, r);
    parent_cv(l, r, root, out);
}
}  // namespace

void blake3_hash(const uint8_t* in, size_t len, uint8_t out[32]) {
    const uint64_t chunks = len == 0 ? 1 : (len + 1023) / 1024;
    uint32_t cv[8];
    subtree_cv(in, len, 0, chunks, true, cv);
    for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(cv[i] >> (8 * k));
}

static HashOut digest_words(const uint8_t d[32]) {
    HashOut h;
    for (int i = 0; i < 4; i++) { u64 x = 0; for (int k = 0; k < 8; k++) x |= (u64)d[8 * i + k] << (8 * k); h[i] = x; }
    return h;
}
static void words_bytes(const u64* w, size_t n, bool canon, std::vector<uint8_t>& out) {
    out.resize(8 * n);
    for (size_t i = 0; i < n; i++) { const u64 x = canon ? gl_canon(w[i]) : w[i]; for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(x >> (8 * k)); }
}

HashOut blake3_hash_no_pad(const u64* in, size_t n) {
    std::vector<uint8_t> bytes;
    words_bytes(in, n, true, bytes);
    uint8_t d[32];
    blake3_hash(bytes.data(), bytes.size(), d);
    return digest_words(d);
}
HashOut blake3_two_to_one(const HashOut& l, const HashOut& r) {
    const u64 w[8] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3]};
    std::vector<uint8_t> bytes;
    words_bytes(w, 8, false, bytes);   // digests are bytes, not field elements: never reduced
    uint8_t d[32];
    blake3_hash(bytes.data(), 64, d);
    return digest_words(d);
}
void blake3_permutation(u64 state[12]) {
    std::vector<uint8_t> cur;
    words_bytes(state, 12, true, cur);
    int got = 0;
    while (got < 12) {
        uint8_t d[32];
        blake3_hash(cur.data(), cur.size(), d);
        cur.assign(d, d + 32);
        const HashOut h = digest_words(d);
        for (int i = 0; i < 4 && got < 12; i++)
            if (h[i] < GL_P) state[got++] = h[i];
    }
}
// BytesHash::to_vec (hash_types.rs:142-152)
void blake3_digest_elements(const HashOut& h, u64 out[5]) {
    uint8_t b[32];
    for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(h[i] >> (8 * k));
    for (int c = 0; c < 5; c++) {
        u64 x = 0;
        for (int k = 0; k < 7 && 7 * c + k < 32; k++) x |= (u64)b[7 * c + k] << (8 * k);
        out[c] = x;
    }
}

// ---- the configuration's hasher, as the Merkle trees and the challenger see it ----
HashOut merkle_hash_leaf(const u64* in, size_t n) { return g_hasher == HASH_BLAKE3 ? blake3_hash_no_pad(in, n) : hash_no_pad(in, n); }
HashOut merkle_two_to_one(const HashOut& l, const HashOut& r) { return g_hasher == HASH_BLAKE3 ? blake3_two_to_one(l, r) : two_to_one(l, r); }

}  // namespace ola_oracle
