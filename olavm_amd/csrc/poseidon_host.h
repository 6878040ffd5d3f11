// Host-side Poseidon permutation for the Fiat-Shamir transcript (a few hundred permutations per proof, strictly
// sequential -- it stays on the CPU; SURVEY build plan step 7).  Same function as the device kernel and as the
// reference's Poseidon::poseidon (plonky2/plonky2/src/hash/poseidon.rs:593-603); sparse partial rounds with the
// tables derived by tools/gen_poseidon_tables.py.
#pragma once
#include "gl.cuh"
#include "../../include/ola_poseidon_constants.h"

namespace ola {

static inline u64 h_sbox7(u64 x) {
    const u64 x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}
static inline u64 h_mod(unsigned __int128 v) { return (u64)(v % GL_P); }

static inline void h_mds_full(u64 s[12]) {
    u64 o[12];
    for (int r = 0; r < 12; r++) {
        unsigned __int128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (unsigned __int128)s[(i + r) % 12] * OLA_POSEIDON_MDS_CIRC[i];
        acc += (unsigned __int128)s[r] * OLA_POSEIDON_MDS_DIAG[r];
        o[r] = h_mod(acc);
    }
    for (int r = 0; r < 12; r++) s[r] = o[r];
}

static inline void poseidon_permute_host(u64 s[12]) {
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = h_sbox7(gl_add(s[i], OLA_POSEIDON_RC[r * 12 + i]));
        h_mds_full(s);
    }
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], OLA_POSEIDON_FAST_FIRST_C[i]);
    {
        u64 t[11];
        for (int r = 0; r < 11; r++) {
            u64 acc = 0;
            for (int c = 0; c < 11; c++) acc = gl_add(acc, gl_mul(OLA_POSEIDON_FAST_INIT[r * 11 + c], s[c + 1]));
            t[r] = acc;
        }
        for (int r = 0; r < 11; r++) s[r + 1] = t[r];
    }
    for (int r = 0; r < 22; r++) {
        const u64 x0 = gl_add(h_sbox7(s[0]), OLA_POSEIDON_FAST_POST_C[r]);
        u64 d = gl_mul(x0, 25);
        for (int j = 0; j < 11; j++) d = gl_add(d, gl_mul(OLA_POSEIDON_FAST_VHAT[r * 11 + j], s[j + 1]));
        for (int j = 0; j < 11; j++) s[j + 1] = gl_add(s[j + 1], gl_mul(x0, OLA_POSEIDON_FAST_W[r * 11 + j]));
        s[0] = d;
    }
    for (int r = 26; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = h_sbox7(gl_add(s[i], OLA_POSEIDON_RC[r * 12 + i]));
        h_mds_full(s);
    }
}

}  // namespace ola
