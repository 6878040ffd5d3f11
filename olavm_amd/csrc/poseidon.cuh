// Poseidon-Goldilocks permutation (width 12, x^7, 4+22+4 rounds) for gfx950, one state per thread in registers.
//
// Computes the same function as the reference's Poseidon::poseidon
//   plonky2/plonky2/src/hash/poseidon.rs:593-603 (full_rounds :560-567, partial_rounds :570-590)
// pinned by the known-answer vectors of poseidon_goldilocks.rs:293-314.  All 30 rounds use the dense MDS layer: its
// coefficients are below 2^6, so on the 32-bit integer ALU it costs 288 single-instruction multiply-adds, less than
// the 64x64-bit products of a sparse partial-round factorisation (measured: 855 -> ~540 VALU instructions per partial
// round).  Round constants sit in __constant__ memory: every lane reads the same address, so they arrive through the
// scalar cache.
#pragma once
#include "gl.cuh"
#include "../../include/ola_poseidon_constants.h"

namespace ola {

#if defined(__HIPCC__)
__constant__ u64 c_rc[360];
__constant__ u64 c_lane0[22];
__constant__ u64 c_round26[12];

static inline void poseidon_upload_constants() {
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_rc), OLA_POSEIDON_RC, sizeof(c_rc)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lane0), OLA_POSEIDON_LANE0_C, sizeof(c_lane0)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_round26), OLA_POSEIDON_ROUND26_C, sizeof(c_round26)));
}

// ---- weakly reduced arithmetic: values are any u64 congruent to the field element; only the permutation's output is
// canonicalised.  Saves the final conditional subtraction of every reduction (the reference's CPU code does the same,
// goldilocks_field.rs:329-345 returns non-canonical u64).
__device__ __forceinline__ void mul_wide32(u64 a, u64 b, u64& lo, u64& hi) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    const u64 v = (u64)a1 * b0 + (u32)u;
    hi = (u64)a1 * b1 + (u >> 32) + (v >> 32);
    lo = (v << 32) | (u32)t;
}
__device__ __forceinline__ u64 reduce128_weak(u64 lo, u64 hi) { return gl_reduce128_weak_cc(lo, hi); }   // carry-flag form, gl.cuh
__device__ __forceinline__ u64 mul_weak(u64 a, u64 b) {
#ifdef OLA_POSEIDON_MUL_ONECHAIN
    // Round 5 experiment, measured and NOT adopted (profiles/r05_sbox_ab.txt): product and reduction as one carry chain
    // (gl_mul_weak_cs, gl.cuh) -- 15 VALU instructions per multiplication instead of 17, bit-exact, and slower: the commitment of
    // 94 x 2^22 196.5 -> 197.8 ms, the Poseidon-configuration proof 0.689 -> 0.715 s.  Every instruction of the chain waits for
    // its predecessor's carry (two wait states each); the 128-bit form below has four independent multiply-adds in flight.
    return gl_mul_weak_cs(a, b);
#else
    u64 lo, hi;
    mul_wide32(a, b, lo, hi);
    return reduce128_weak(lo, hi);
#endif
}
// x weak, c canonical -> weak
__device__ __forceinline__ u64 add_weak(u64 x, u64 c) {
    u64 s = x + c;
    s += (s < c) ? GL_EPS : 0;             // wrapped sum is < c < p, so adding EPS cannot wrap again
    return s;
}
__device__ __forceinline__ u64 sbox7_weak(u64 x) {
    const u64 x2 = mul_weak(x, x), x4 = mul_weak(x2, x2), x3 = mul_weak(x, x2);
    return mul_weak(x3, x4);
}

// MDS layer (poseidon.rs:170-190): circulant [17,15,41,16,2,28,13,13,39,18,34,20] + diag [8,0,..].  Every coefficient is
// below 2^6, so the 32-bit halves of the (weak) state are accumulated un-reduced with one v_mad_u64_u32 per term and
// each lane is folded once: value = al + ah*2^32 with al, ah < 2^42.
__device__ __forceinline__ void mds_weak(u64 (&s)[12]) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 l[12], h[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { l[i] = (u32)s[i]; h[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const u32 c = C[i] + ((r == 0 && i == 0) ? 8u : 0u);
            al += (u64)l[(i + r) % 12] * c;
            ah += (u64)h[(i + r) % 12] * c;
        }
        // ah = ah_lo + ah_hi*2^32 (ah_hi < 2^10):  value = (al + ah_hi*EPS) + ah_lo*2^32  (mod p)
        const u64 m = al + (u64)(u32)(ah >> 32) * 0xFFFFFFFFu;   // < 2^43
        const u32 m_hi = (u32)(m >> 32), a_lo = (u32)ah;
        const u32 r_hi = m_hi + a_lo;
        u64 v = ((u64)r_hi << 32) | (u32)m;
        v += (r_hi < a_lo) ? GL_EPS : 0;                          // wrapped: r_hi is small, no second wrap
        s[r] = v;
    }
}

// Fold al + ah*2^32 (al, ah < 2^63) into a weak u64: ah = ah_lo + ah_hi*2^32, 2^64 = EPS (mod p).
__device__ __forceinline__ u64 fold_halves(u64 al, u64 ah) {
    const u64 m = al + (u64)(u32)(ah >> 32) * 0xFFFFFFFFu;   // < 2^63 + 2^63
    const u32 m_hi = (u32)(m >> 32), a_lo = (u32)ah;
    const u32 r_hi = m_hi + a_lo;
    u64 v = ((u64)r_hi << 32) | (u32)m;
    v += (r_hi < a_lo) ? GL_EPS : 0;                          // wrapped once at most (see derive_fused3)
    return v;
}

// Three partial rounds as one small-integer linear layer (tools/gen_poseidon_tables.py derive_fused3): 386 multiply-adds
// instead of 3 x 288, one fold per lane instead of three.
__device__ __forceinline__ void partial3_weak(u64 (&s)[12], u64 k0, u64 k1, u64 k2) {
    constexpr u32 R1[12] = OLA_POSEIDON_FUSED3_R1_INIT;
    constexpr u32 R2[13] = OLA_POSEIDON_FUSED3_R2_INIT;
    constexpr u32 F[168] = OLA_POSEIDON_FUSED3_F_INIT;
    u32 l[14], h[14];   // halves of s[1..11], then t0, t1, t2
#pragma unroll
    for (int i = 0; i < 11; i++) { l[i] = (u32)s[i + 1]; h[i] = (u32)(s[i + 1] >> 32); }
    const u64 t0 = sbox7_weak(add_weak(s[0], k0));
    l[11] = (u32)t0; h[11] = (u32)(t0 >> 32);
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { al += (u64)l[i] * R1[i]; ah += (u64)h[i] * R1[i]; }
    const u64 t1 = sbox7_weak(add_weak(fold_halves(al, ah), k1));
    l[12] = (u32)t1; h[12] = (u32)(t1 >> 32);
    al = 0; ah = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) { al += (u64)l[i] * R2[i]; ah += (u64)h[i] * R2[i]; }
    const u64 t2 = sbox7_weak(add_weak(fold_halves(al, ah), k2));
    l[13] = (u32)t2; h[13] = (u32)(t2 >> 32);
#pragma unroll
    for (int o = 0; o < 12; o++) {
        al = 0; ah = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) { al += (u64)l[i] * F[14 * o + i]; ah += (u64)h[i] * F[14 * o + i]; }
        s[o] = fold_halves(al, ah);
    }
}

// Same function as the reference's Poseidon::poseidon (poseidon.rs:593-603).  The partial rounds are the reference's
// dense form (constant layer, x^7 on lane 0, MDS layer; partial_rounds_naive) with the constants pushed onto lane 0
// (tools/gen_poseidon_tables.py derive_lane0) and fused three at a time (derive_fused3); output is canonical.
__device__ __forceinline__ void poseidon_permute(u64 (&s)[12]) {
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox7_weak(add_weak(s[i], c_rc[r * 12 + i]));
        mds_weak(s);
    }
    // 22 partial rounds = 7 fused triples + 1 plain round
#pragma unroll 1
    for (int g = 0; g < 7; g++) partial3_weak(s, c_lane0[3 * g], c_lane0[3 * g + 1], c_lane0[3 * g + 2]);
    s[0] = sbox7_weak(add_weak(s[0], c_lane0[21]));
    mds_weak(s);
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox7_weak(add_weak(s[i], r == 26 ? c_round26[i] : c_rc[r * 12 + i]));
        mds_weak(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}
// ------------------------------------------------------------------------------------------------ quad-cooperative form
// Latency-oriented variant for small batches (the upper levels of every Merkle tree, the trees of small tables): FOUR
// lanes of a DPP quad share one state, lane q = lane & 3 holding elements 3q, 3q+1, 3q+2.  The twelve S-boxes of a full
// round run three per lane, and the MDS layer -- a circulant -- is evaluated on the inputs ROTATED by 3q (fetched from
// the other lanes with quad_perm DPP moves), which makes the coefficients the same compile-time constants on every lane.
// One permutation is then a dependent chain of ~8 k instructions instead of ~19 k: a tree level that does not fill the
// chip anyway finishes in less than half the time.  Same function, same outputs as poseidon_permute.
template <int ROT>
__device__ __forceinline__ u32 quad_rot(u32 v) {
    // lane i of each quad reads lane (i + ROT) & 3
    constexpr int ctrl = ((0 + ROT) & 3) | (((1 + ROT) & 3) << 2) | (((2 + ROT) & 3) << 4) | (((3 + ROT) & 3) << 6);
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false);
}

__device__ __forceinline__ void mds_quad(u64 (&x)[3], int q) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 l[12], h[12];   // y_m = element (m + 3q) mod 12
#pragma unroll
    for (int k = 0; k < 3; k++) { l[k] = (u32)x[k]; h[k] = (u32)(x[k] >> 32); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        l[3 + k] = quad_rot<1>(l[k]); h[3 + k] = quad_rot<1>(h[k]);
        l[6 + k] = quad_rot<2>(l[k]); h[6 + k] = quad_rot<2>(h[k]);
        l[9 + k] = quad_rot<3>(l[k]); h[9 + k] = quad_rot<3>(h[k]);
    }
    const u32 diag = (q == 0) ? 8u : 0u;   // + 8 * element 0 in output 0 only
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // out_{3q+k} = sum_i C[(i - 3q - k) mod 12] * x_i = sum_m C[(m - k) mod 12] * y_m
        u64 al = 0, ah = 0;
#pragma unroll
        for (int m = 0; m < 12; m++) {
            const u32 c = C[(m - k + 12) % 12];
            al += (u64)l[m] * c;
            ah += (u64)h[m] * c;
        }
        if (k == 0) { al += (u64)l[0] * diag; ah += (u64)h[0] * diag; }
        x[k] = fold_halves(al, ah);
    }
}

__device__ __forceinline__ void poseidon_permute_quad(u64 (&x)[3], int q) {
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < 3; k++) x[k] = sbox7_weak(add_weak(x[k], c_rc[r * 12 + 3 * q + k]));
        mds_quad(x, q);
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {   // dense partial rounds, constants on lane 0 of the state (derive_lane0)
        const u64 t = sbox7_weak(add_weak(x[0], c_lane0[r]));
        x[0] = (q == 0) ? t : x[0];
        mds_quad(x, q);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int k = 0; k < 3; k++) x[k] = sbox7_weak(add_weak(x[k], r == 26 ? c_round26[3 * q + k] : c_rc[r * 12 + 3 * q + k]));
        mds_quad(x, q);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) x[k] = gl_canon(x[k]);
}
#endif  // __HIPCC__

}  // namespace ola
