// BLAKE3 for the Merkle trees and the challenger of the reference's Blake3GoldilocksConfig
// (plonky2/plonky2/src/plonk/config.rs:153-161: Hasher = Blake3_256<32>, InnerHasher = PoseidonHash).
//
// Replaces (reference, relative to plonky2/plonky2/src/hash):
//   blake3.rs:203-213   Blake3_256::hash_no_pad(field slice) = blake3::hash(its bytes)     -> b3_leaf_* (one thread per leaf)
//   blake3.rs:215-233   two_to_one = blake3::hash(left || right)                            -> b3_node (one 64-byte block)
//   blake3.rs:166-201   Blake3Permutation (the challenger's "onion")                        -> b3_permutation_host
//   hash_types.rs:142-152  BytesHash::to_vec: a digest is observed as 5 elements of 7 bytes -> b3_digest_elements
// The hash function is crate blake3 1.5.0 (Cargo.lock:220); this is the published algorithm: 64-byte blocks, 1024-byte chunks,
// a 7-round compression function on sixteen 32-bit words, chunk chaining values joined by a binary tree whose left subtree
// holds the largest power of two of chunks.  Field elements are hashed as canonical little-endian words -- the words the
// reference's verifier re-hashes after deserialising a proof (merkle_proofs.rs:52-80, serialization read_field) -- whereas
// the reference's prover hashes whatever representative lies in memory (blake3.rs:204-207); see DESIGN.md (f-3).
// A digest is 32 bytes = 4 little-endian u64 words in the same heap layout as the Poseidon digests; it is bytes, not field
// elements, and is never reduced.
//
// Cost on gfx950: one compression = 56 G functions x (6 add, 4 xor, 4 v_alignbit) = 784 VALU + moves; a 94-column leaf is 12
// blocks -- about 10 k instructions against 216 k for the Poseidon sponge -- so leaf hashing turns from VALU-bound into a pass
// over the LDE at HBM speed (8 B per element read, column-major, consecutive lanes on consecutive addresses).
#pragma once
#include <hip/hip_runtime.h>

#include "gl.cuh"

namespace ola {

#define B3_HD __host__ __device__ __forceinline__

enum : u32 { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };
enum : int { B3_CHUNK_WORDS = 128 };   // u64 words per 1024-byte chunk

B3_HD u32 b3_iv(int i) {
    constexpr u32 IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    return IV[i];
}
// message word used at position i of round r: the round permutation applied r times (blake3_goldilocks.rs:12-20 lists the rows)
B3_HD constexpr int b3_sched(int r, int i) {
    constexpr int P[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
    int k = i;
    for (int t = 0; t < r; t++) k = P[k];
    return k;
}
B3_HD u32 b3_rotr(u32 x, int n) { return __builtin_rotateright32(x, (u32)n); }

#define B3_G(a, b, c, d, x, y)                                   \
    s[a] = s[a] + s[b] + (x); s[d] = b3_rotr(s[d] ^ s[a], 16);   \
    s[c] = s[c] + s[d];       s[b] = b3_rotr(s[b] ^ s[c], 12);   \
    s[a] = s[a] + s[b] + (y); s[d] = b3_rotr(s[d] ^ s[a], 8);    \
    s[c] = s[c] + s[d];       s[b] = b3_rotr(s[b] ^ s[c], 7);

template <int R>
B3_HD void b3_round(u32 (&s)[16], const u32 (&m)[16]) {
    B3_G(0, 4, 8, 12, m[b3_sched(R, 0)], m[b3_sched(R, 1)])
    B3_G(1, 5, 9, 13, m[b3_sched(R, 2)], m[b3_sched(R, 3)])
    B3_G(2, 6, 10, 14, m[b3_sched(R, 4)], m[b3_sched(R, 5)])
    B3_G(3, 7, 11, 15, m[b3_sched(R, 6)], m[b3_sched(R, 7)])
    B3_G(0, 5, 10, 15, m[b3_sched(R, 8)], m[b3_sched(R, 9)])
    B3_G(1, 6, 11, 12, m[b3_sched(R, 10)], m[b3_sched(R, 11)])
    B3_G(2, 7, 8, 13, m[b3_sched(R, 12)], m[b3_sched(R, 13)])
    B3_G(3, 4, 9, 14, m[b3_sched(R, 14)], m[b3_sched(R, 15)])
}

// cv <- first half of compress(cv, m, counter, block_len, flags): the new chaining value (and, under B3_ROOT, the digest)
B3_HD void b3_compress(u32 (&cv)[8], const u32 (&m)[16], u32 counter, u32 block_len, u32 flags) {
    u32 s[16];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = cv[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s[8 + i] = b3_iv(i);
    s[12] = counter; s[13] = 0; s[14] = block_len; s[15] = flags;   // chunk counters of a leaf fit 32 bits by far
    b3_round<0>(s, m); b3_round<1>(s, m); b3_round<2>(s, m); b3_round<3>(s, m); b3_round<4>(s, m); b3_round<5>(s, m); b3_round<6>(s, m);
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = s[i] ^ s[i + 8];
}
B3_HD void b3_cv_init(u32 (&cv)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = b3_iv(i);
}
// parent node of two chaining values (also two_to_one of two digests, with extra = B3_CHUNK_START | B3_CHUNK_END | B3_ROOT and
// no B3_PARENT: a 64-byte message is a single-block chunk, not a tree node)
B3_HD void b3_pair(const u32 (&l)[8], const u32 (&r)[8], u32 flags, u32 (&out)[8]) {
    u32 m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = l[i]; m[8 + i] = r[i]; }
    b3_cv_init(out);
    b3_compress(out, m, 0, 64, flags);
}
B3_HD void b3_store_digest(u64* __restrict__ out4, const u32 (&cv)[8]) {
#pragma unroll
    for (int k = 0; k < 4; k++) out4[k] = (u64)cv[2 * k] | ((u64)cv[2 * k + 1] << 32);
}

// Chaining value of chunk `index`: words [w0, w0 + cw) of the input, 1 <= cw <= 128, delivered by `word(i)`; `root` when the
// input is this one chunk (then the result is the digest).  Control flow depends on cw only, which is uniform over a launch.
template <class WordFn>
B3_HD void b3_chunk_cv(WordFn word, u32 w0, u32 cw, u32 index, bool root, u32 (&cv)[8]) {
    const u32 nblocks = (cw + 7) / 8;
    b3_cv_init(cv);
    for (u32 b = 0; b < nblocks; b++) {
        const u32 bw = cw - 8 * b < 8u ? cw - 8 * b : 8u;
        u32 m[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 x = (u32)i < bw ? word(w0 + 8 * b + i) : 0;
            m[2 * i] = (u32)x; m[2 * i + 1] = (u32)(x >> 32);
        }
        const u32 flags = (b == 0 ? (u32)B3_CHUNK_START : 0u) | (b + 1 == nblocks ? (u32)B3_CHUNK_END | (root ? (u32)B3_ROOT : 0u) : 0u);
        b3_compress(cv, m, index, 8 * bw, flags);
    }
}

// Hash of `nwords` >= 1 canonical field words of any length up to 4096 words (32 chunks): chunk chaining values joined as the
// incremental algorithm of the BLAKE3 paper (section 5.1.2) does -- a stack of finished subtrees, one join per trailing zero of
// the count of finished chunks, the last chunk folded in from the top of the stack down, the last join flagged ROOT.
template <class WordFn>
B3_HD void b3_hash_words(WordFn word, u32 nwords, u32 (&digest)[8]) {
    const u32 nchunks = (nwords + B3_CHUNK_WORDS - 1) / B3_CHUNK_WORDS;
    if (nchunks == 1) { b3_chunk_cv(word, 0, nwords, 0, true, digest); return; }
    u32 stack[6][8];
    u32 depth = 0;
    for (u32 c = 0; c < nchunks; c++) {
        const u32 w0 = c * B3_CHUNK_WORDS;
        const u32 cw = nwords - w0 < (u32)B3_CHUNK_WORDS ? nwords - w0 : (u32)B3_CHUNK_WORDS;
        u32 cv[8], p[8];
        b3_chunk_cv(word, w0, cw, c, false, cv);
        if (c + 1 < nchunks) {
            for (u32 total = c + 1; (total & 1) == 0; total >>= 1) {
                depth--;
                b3_pair(stack[depth], cv, B3_PARENT, p);
#pragma unroll
                for (int i = 0; i < 8; i++) cv[i] = p[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) stack[depth][i] = cv[i];
            depth++;
        } else {
            while (depth > 0) {
                depth--;
                b3_pair(stack[depth], cv, B3_PARENT | (depth == 0 ? (u32)B3_ROOT : 0u), p);
#pragma unroll
                for (int i = 0; i < 8; i++) cv[i] = p[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) digest[i] = cv[i];
        }
    }
}

// digest of the 64 bytes of two digests (8 u64 words, not reduced)
B3_HD void b3_node(const u64* __restrict__ children8, u64* __restrict__ out4) {
    u32 m[16], cv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const u64 x = children8[k]; m[2 * k] = (u32)x; m[2 * k + 1] = (u32)(x >> 32); }
    b3_cv_init(cv);
    b3_compress(cv, m, 0, 64, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
    b3_store_digest(out4, cv);
}

// ---- host side of the transcript ----
static inline void b3_hash_bytes32_host(const u64* words, u32 nwords, u64 out4[4]) {
    u32 d[8];
    b3_hash_words([&](u32 i) { return words[i]; }, nwords, d);
    b3_store_digest(out4, d);
}
// Blake3Permutation::permute (blake3.rs:166-201)
static inline void b3_permutation_host(u64 state[12]) {
    u64 cur[12];
    for (int i = 0; i < 12; i++) cur[i] = gl_canon(state[i]);
    u32 nwords = 12;
    int got = 0;
    while (got < 12) {
        u64 h[4];
        b3_hash_bytes32_host(cur, nwords, h);
        for (int i = 0; i < 4; i++) {
            cur[i] = h[i];
            if (h[i] < GL_P && got < 12) state[got++] = h[i];
        }
        nwords = 4;
    }
}
// BytesHash::to_vec (hash_types.rs:142-152)
static inline void b3_digest_elements(const u64 h[4], u64 out[5]) {
    uint8_t b[32];
    for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(h[i] >> (8 * k));
    for (int c = 0; c < 5; c++) {
        u64 x = 0;
        for (int k = 0; k < 7 && 7 * c + k < 32; k++) x |= (u64)b[7 * c + k] << (8 * k);
        out[c] = x;
    }
}

}  // namespace ola
