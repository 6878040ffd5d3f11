// Poseidon-Goldilocks permutation (width 12, x^7, 4+22+4 rounds) for gfx950, one state per thread in registers.
//
// Computes the same function as the reference's Poseidon::poseidon
//   plonky2/plonky2/src/hash/poseidon.rs:593-603 (full_rounds :560-567, partial_rounds :570-590)
// pinned by the known-answer vectors of poseidon_goldilocks.rs:293-314.  The partial rounds use OUR OWN sparse
// factorisation of the linear layers (tables derived by tools/gen_poseidon_tables.py), the full-round MDS layer
// (circulant [17,15,41,16,2,28,13,13,39,18,34,20] + diag [8,0..]) is accumulated un-reduced in 32-bit halves and
// reduced once per lane.  Round constants sit in __constant__ memory: every lane reads the same address, so they
// arrive through the scalar cache.
#pragma once
#include "gl.cuh"
#include "../../include/ola_poseidon_constants.h"

namespace ola {

#if defined(__HIPCC__)
__constant__ u64 c_rc[360];
__constant__ u64 c_first_c[12];
__constant__ u64 c_post_c[22];
__constant__ u64 c_vhat[22 * 11];
__constant__ u64 c_w[22 * 11];
__constant__ u64 c_init[11 * 11];

static inline void poseidon_upload_constants() {
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_rc), OLA_POSEIDON_RC, sizeof(c_rc)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_first_c), OLA_POSEIDON_FAST_FIRST_C, sizeof(c_first_c)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_post_c), OLA_POSEIDON_FAST_POST_C, sizeof(c_post_c)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_vhat), OLA_POSEIDON_FAST_VHAT, sizeof(c_vhat)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_w), OLA_POSEIDON_FAST_W, sizeof(c_w)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_init), OLA_POSEIDON_FAST_INIT, sizeof(c_init)));
}

__device__ __forceinline__ u64 sbox7(u64 x) {
    const u64 x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

// 128-bit accumulate helpers
__device__ __forceinline__ void acc128_mul(u64& lo, u64& hi, u64 a, u64 b) {
    u64 pl, ph;
    mul_wide(a, b, pl, ph);
    lo += pl;
    hi += ph + (lo < pl);
}

__device__ __forceinline__ void mds_full(u64 (&s)[12]) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 l[12], h[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { l[i] = (u32)s[i]; h[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al = 0, ah = 0;  // sums of 32-bit halves times constants < 2^6: each < 12*41*2^32 < 2^41
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (u64)l[(i + r) % 12] * C[i];
            ah += (u64)h[(i + r) % 12] * C[i];
        }
        if (r == 0) { al += (u64)l[0] * 8; ah += (u64)h[0] * 8; }
        // value = al + ah * 2^32  (< 2^74)
        const u64 lo = al + (ah << 32);
        const u64 hi = (ah >> 32) + (lo < al);
        s[r] = gl_reduce128(lo, hi);
    }
}

__device__ __forceinline__ void poseidon_permute(u64 (&s)[12]) {
    // first 4 full rounds
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], c_rc[r * 12 + i]));
        mds_full(s);
    }
    // partial rounds, sparse form
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], c_first_c[i]);
    {
        u64 t[11];
#pragma unroll
        for (int r = 0; r < 11; r++) {
            u64 lo = 0, hi = 0, top = 0;  // 11 products < 2^128 each
#pragma unroll
            for (int c = 0; c < 11; c++) {
                u64 pl, ph;
                mul_wide(c_init[r * 11 + c], s[c + 1], pl, ph);
                lo += pl;
                const u64 cy = (lo < pl);
                hi += cy; top += (hi < cy);
                hi += ph; top += (hi < ph);
            }
            // value = top*2^128 + hi*2^64 + lo ; 2^128 = -2^32 (mod p)
            u64 v = gl_reduce128(lo, hi);
            t[r] = gl_sub(v, gl_reduce128(top << 32, 0));
        }
#pragma unroll
        for (int r = 0; r < 11; r++) s[r + 1] = t[r];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        const u64 x0 = gl_add(sbox7(s[0]), c_post_c[r]);
        // d = 25*x0 + sum vhat[j]*s[j+1]
        u64 lo = x0 * 25, hi = __umul64hi(x0, 25), top = 0;
#pragma unroll
        for (int j = 0; j < 11; j++) {
            u64 pl, ph;
            mul_wide(c_vhat[r * 11 + j], s[j + 1], pl, ph);
            lo += pl;
            const u64 cy = (lo < pl);
            hi += cy; top += (hi < cy);
            hi += ph; top += (hi < ph);
        }
        const u64 d = gl_sub(gl_reduce128(lo, hi), gl_reduce128(top << 32, 0));
#pragma unroll
        for (int j = 0; j < 11; j++) {
            u64 pl, ph;
            mul_wide(x0, c_w[r * 11 + j], pl, ph);
            pl += s[j + 1];
            ph += (pl < s[j + 1]);  // < 2^128: x0*w <= (p-1)^2, + s < 2^128
            s[j + 1] = gl_reduce128(pl, ph);
        }
        s[0] = d;
    }
    // last 4 full rounds
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], c_rc[r * 12 + i]));
        mds_full(s);
    }
}
#endif  // __HIPCC__

}  // namespace ola
