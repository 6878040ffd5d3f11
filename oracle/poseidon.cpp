// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of Poseidon-Goldilocks.
//
// Follows (relative to /root/reference/plonky2/plonky2/src/hash):
//   poseidon.rs:24-27        HALF_N_FULL_ROUNDS = 4, N_PARTIAL_ROUNDS = 22, width 12
//   poseidon.rs:476-486      constant_layer          (round constants, table poseidon.rs:50-148)
//   poseidon.rs:516-522      sbox_monomial x -> x^7
//   poseidon.rs:170-190,236-255  mds_row_shf / mds_layer  (circulant + diagonal, poseidon_goldilocks.rs:22-23)
//   poseidon.rs:606-627      partial_rounds_naive / poseidon_naive  -- the plain round structure, which the
//                            reference proves equal to its fast form in test poseidon.rs:714-727
//   hashing.rs:84-107        hash_n_to_m_no_pad (overwrite-mode sponge, rate 8), hashing.rs:66-74 compress
//   poseidon.rs:640-652      PoseidonHash::{hash_no_pad, two_to_one}
// Pinned by the 4 known-answer vectors of poseidon_goldilocks.rs:293-314 (tests/golden/poseidon_kat.json).
#include "oracle.hpp"
#include "../include/ola_poseidon_constants.h"

namespace ola_oracle {

static inline u64 sbox7(u64 x) {
    u64 x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

// poseidon.rs:236-255 mds_layer, as the reference computes it: the state is split into 32-bit halves so that a row's twelve
// products (coefficients below 2^6) accumulate in plain u64 words -- 13 * 2^32 * 2^6 < 2^42 -- and the two sums are joined and
// reduced once per row.  Same value as the sum of 128-bit products reduced mod p (full-size commitments are checked against this
// oracle: 10^8 permutations per batch).
static void mds_layer(u64 s[12]) {
    // 32-bit halves in 32-bit arrays: the products below are then 32 x 32 -> 64-bit widening multiplications, which the
    // compiler turns into vpmuludq (four rows per instruction) under -mavx2
    uint32_t lo[24], hi[24];
    u64 al[12], ah[12];
    for (int i = 0; i < 12; i++) {
        lo[i] = lo[i + 12] = (uint32_t)s[i];
        hi[i] = hi[i + 12] = (uint32_t)(s[i] >> 32);
    }
    for (int r = 0; r < 12; r++) {
        al[r] = (u64)lo[r] * (uint32_t)OLA_POSEIDON_MDS_DIAG[r];
        ah[r] = (u64)hi[r] * (uint32_t)OLA_POSEIDON_MDS_DIAG[r];
    }
    for (int i = 0; i < 12; i++) {          // row r takes s[(i + r) % 12] * circ[i]: twelve rows at a time
        const uint32_t c = (uint32_t)OLA_POSEIDON_MDS_CIRC[i];
        for (int r = 0; r < 12; r++) {
            al[r] += (u64)lo[i + r] * c;
            ah[r] += (u64)hi[i + r] * c;
        }
    }
    for (int r = 0; r < 12; r++) s[r] = gl_reduce128((u128)al[r] + ((u128)ah[r] << 32));
}

void poseidon_naive(u64 s[12]) {
    int round = 0;
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
    for (int f = 0; f < 4; f++, round++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], OLA_POSEIDON_RC[round * 12 + i]));
        mds_layer(s);
    }
    for (int p = 0; p < 22; p++, round++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], OLA_POSEIDON_RC[round * 12 + i]);
        s[0] = sbox7(s[0]);
        mds_layer(s);
    }
    for (int f = 0; f < 4; f++, round++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], OLA_POSEIDON_RC[round * 12 + i]));
        mds_layer(s);
    }
}

State poseidon(const State& in) {
    State s = in;
    poseidon_naive(s.data());
    return s;
}

HashOut hash_no_pad(const u64* in, size_t n) {
    u64 st[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        size_t len = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < len; i++) st[i] = gl_canon(in[off + i]);
        poseidon_naive(st);
    }
    // NB (hashing.rs:84-107): with zero inputs no permutation is applied and the output is all-zero.
    return HashOut{st[0], st[1], st[2], st[3]};
}

HashOut two_to_one(const HashOut& l, const HashOut& r) {
    u64 st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    poseidon_naive(st);
    return HashOut{st[0], st[1], st[2], st[3]};
}

}  // namespace ola_oracle
