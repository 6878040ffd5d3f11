// Goldilocks field arithmetic for the MI355X proving backend (host + gfx950 device).
// p = 2^64 - 2^32 + 1.  Values held in registers/LDS/HBM are CANONICAL (< p) unless a function says otherwise;
// everything written back to the caller is canonical, matching what the reference serialises
// (reference semantics: plonky2/field/src/goldilocks_field.rs:191-355 -- add/sub/mul/reduce128; the reference keeps
// non-canonical u64 in memory and canonicalises at the boundary, the exact value mod p is identical).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace ola {

typedef unsigned long long u64;
typedef unsigned int u32;

static const u64 GL_P = 0xFFFFFFFF00000001ull;
static const u64 GL_EPS = 0xFFFFFFFFull;  // 2^64 mod p = 2^32 - 1
static const u64 GL_GENERATOR = 7;        // multiplicative generator == coset shift (types.rs:430)
static const u64 GL_POWER_OF_TWO_GENERATOR = 1753635133440165772ull;

GL_HD u64 gl_canon(u64 x) { return x >= GL_P ? x - GL_P : x; }

// a, b canonical -> canonical.  a+b < 2p < 2^65: subtract p when the 64-bit add carried or the sum >= p,
// the latter detected as a carry out of (s + EPS) since p + EPS = 2^64.
GL_HD u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    u64 t = s + GL_EPS;
    bool c = (s < a) | (t < s);
    return c ? t : s;
}
// a, b canonical -> canonical.
GL_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    return (a < b) ? d - GL_EPS : d;  // + p == - EPS (mod 2^64)
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

// 64x64 -> 128 multiply
GL_HD void mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    // schoolbook on 32-bit halves: four v_mad_u64_u32 (full rate on gfx950), the partial sums cannot overflow 64 bits
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    const u64 v = (u64)a1 * b0 + (u32)u;
    hi = (u64)a1 * b1 + (u >> 32) + (v >> 32);
    lo = (v << 32) | (u32)t;
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (u64)p;
    hi = (u64)(p >> 64);
#endif
}

// (hi*2^64 + lo) mod p as any u64 of the residue class.  Uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
GL_HD u64 gl_reduce128_weak(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & GL_EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;          // wrapped: add p
    u64 t1 = (hl << 32) - hl;           // hl * (2^32 - 1) < 2^64
    u64 t2 = t0 + t1;
    if (t2 < t0) t2 += GL_EPS;          // wrapped: 2^64 = EPS (mod p); cannot wrap twice
    return t2;
}
// (hi*2^64 + lo) mod p, canonical.  The plain C++ form: the one to use when the operands have structure the optimiser can see
// (a shifted word in gl_mul_pow2: half of the steps fold away).
GL_HD u64 gl_reduce128(u64 lo, u64 hi) { return gl_canon(gl_reduce128_weak(lo, hi)); }

// ---- carry-flag forms of the same reduction for GENERAL operands (device only) ----
// hipcc cannot name the carry a VALU instruction produces: after `lo - hh` it finds the borrow again with a 64-bit compare,
// and after `t0 + hl * EPS` it multiplies a second time to compare against the product -- 17 instructions for a canonical
// reduction whose data flow needs 10 (12 against 8 for the weak one).  These helpers spell the carry chain out: v_sub_co /
// v_subbrev_co leave the borrow in an SGPR pair that the selects read directly, v_mad_u64_u32's carry-out drives the
// wrap-around fix.  One instruction per asm statement, so the compiler still schedules around them; it does not know that one
// statement's SGPR result is the next one's carry-in, hence every statement that READS a carry opens with `s_nop 1` -- the two
// wait states hipcc itself puts between a VALU write of an SGPR / VCC and a VALU read of it as carry or select mask on gfx950.
// Measured (tests/gpu_glasm_selftest.cpp, 256 CUs): 2.63 T modular multiplications per second against 1.80 T for the C++ form;
// 4 * 10^9 random and 45 k edge-case operand pairs agree.  The optimiser cannot see through them: keep the C++ form wherever it
// can simplify.  OLA_GL_NO_ASM selects the C++ forms everywhere.  The chains assume wave64 lane masks in SGPR pairs, the gfx9 VOP3
// carry encodings and gfx950's wait-state rules: they are compiled for gfx950 only (this library's one target); any other
// --offload-arch gets the C++ forms, and ola_gpu_selftest compares the two on the device.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OLA_GL_NO_ASM) && defined(__gfx950__)
#define OLA_GL_ASM 1
// t0 = lo - hh (mod p as a wrapped u64: p is added back when the subtraction borrowed), then t0 + hl * (2^32 - 1) as a
// wrapped u64 plus its carry-out in `carry` (an SGPR lane mask).
__device__ __forceinline__ u64 gl_fold128_carry(u64 lo, u64 hi, u64& carry) {
    const u32 l0 = (u32)lo, l1 = (u32)(lo >> 32), hl = (u32)hi, hh = (u32)(hi >> 32);
    u32 t0l, t0h, el, eh;
    u64 b0, b1, t2;
    asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(t0l), "=s"(b0) : "v"(l0), "v"(hh));
    asm("s_nop 1\n\tv_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(t0h), "=s"(b1) : "v"(l1), "s"(b0));
    // borrowed: the wrapped difference is 2^64 too large = EPS too large (mod p): add -EPS = 0xFFFFFFFF00000001
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %2\n\tv_cndmask_b32_e64 %1, 0, -1, %2" : "=&v"(el), "=v"(eh) : "s"(b1));
    u64 fix = ((u64)eh << 32) | el, dif = ((u64)t0h << 32) | t0l;
    asm("" : "+v"(fix), "+v"(dif));      // two 64-bit values: without this the optimiser adds their four halves one by one
    const u64 t0 = dif + fix;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t2), "=s"(carry) : "v"(hl), "v"(t0));
    return t2;
}
// weak result: any u64 in the residue class (the wrap-around fix cannot wrap again: a wrapped sum is below hl * EPS)
__device__ __forceinline__ u64 gl_reduce128_weak_cc(u64 lo, u64 hi) {
    u64 c;
    const u64 t2 = gl_fold128_carry(lo, hi, c);
    u32 e;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e) : "s"(c));
    return t2 + e;
}
// canonical result: + EPS when the sum wrapped (2^64 = EPS) or when it is >= p (then + EPS wraps to the value - p);
// both cases are "take t2 + EPS", and the second shows as the carry of that very addition
__device__ __forceinline__ u64 gl_reduce128_cc(u64 lo, u64 hi) {
    u64 c, c1, c2, m;
    const u64 t2 = gl_fold128_carry(lo, hi, c);
    const u32 tl = (u32)t2, th = (u32)(t2 >> 32);
    u32 ul, uh, r0, r1;
    asm("v_add_co_u32_e64 %0, %1, -1, %2" : "=v"(ul), "=s"(c1) : "v"(tl));
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(uh), "=s"(c2) : "v"(th), "s"(c1));
    asm("s_nop 1\n\ts_or_b64 %2, %3, %4\n\tv_cndmask_b32_e64 %0, %5, %6, %2\n\tv_cndmask_b32_e64 %1, %7, %8, %2"
        : "=&v"(r0), "=&v"(r1), "=&s"(m) : "s"(c), "s"(c2), "v"(tl), "v"(ul), "v"(th), "v"(uh) : "scc");   // s_or writes SCC
    return ((u64)r1 << 32) | r0;
}
// a * b mod p as any u64 of the residue class, for ANY u64 operands, without assembling the 128-bit product first (round 5
// experiment for the S-boxes of the Poseidon permutation -- 472 multiplications per permutation, 55 % of its instructions;
// checked by ola_gpu_selftest, measured slower than the 128-bit form and left out of the permutation, see poseidon.cuh mul_weak).  With a = a0 + a1 2^32,
// b = b0 + b1 2^32:  al = a0 b0,  ah = a1 b1,  (c, mid) = a0 b1 + a1 b0 as a 65-bit sum (the carry-out of the second
// v_mad_u64_u32), and with 2^64 = 2^32 - 1 =: EPS, 2^96 = -1 (mod p)
//     a b = al_l + (al_h + mid_l) 2^32 + (mid_h + ah_l) 2^64 + (ah_h + c) 2^96
//         = {al_l, Ah} - (ah_h + c + k2) + B EPS,     Ah + k1 2^32 = al_h + mid_l,   B + k2 2^32 = mid_h + ah_l + k1,
// i.e. two 32-bit additions chained through the carry instead of the three 64-bit additions (each with its zero-extension move)
// that build `hi` in mul_wide -- 15 VALU instructions against 17, all four multiplications independent of each other.
// ah_h <= 2^32 - 2, so ah_h + c fits a word; the subtract-with-borrow takes k2 as its borrow-in.
__device__ __forceinline__ u64 gl_mul_weak_cs(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 al = (u64)a0 * b0, ah = (u64)a1 * b1, m1 = (u64)a0 * b1;
    u64 mid, c, k1, k2, kd, bw0, bw1, carry, t2;
    u32 Ah, B, Dh, t0l, t0h, el, eh, e;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(mid), "=s"(c) : "v"(a1), "v"(b0), "v"(m1));
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(Ah), "=s"(k1) : "v"((u32)(al >> 32)), "v"((u32)mid));
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(B), "=s"(k2) : "v"((u32)(mid >> 32)), "v"((u32)ah), "s"(k1));
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(Dh), "=s"(kd) : "v"((u32)(ah >> 32)), "s"(c));
    asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(t0l), "=s"(bw0) : "v"((u32)al), "v"(Dh), "s"(k2));
    asm("s_nop 1\n\tv_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(t0h), "=s"(bw1) : "v"(Ah), "s"(bw0));
    // borrowed: the wrapped difference is 2^64 = EPS too large (mod p): add -EPS = 0xFFFFFFFF00000001
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %2\n\tv_cndmask_b32_e64 %1, 0, -1, %2" : "=&v"(el), "=v"(eh) : "s"(bw1));
    u64 fix = ((u64)eh << 32) | el, dif = ((u64)t0h << 32) | t0l;
    asm("" : "+v"(fix), "+v"(dif));
    const u64 t0 = dif + fix;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t2), "=s"(carry) : "v"(B), "v"(t0));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e) : "s"(carry));
    return t2 + e;
}
#else
GL_HD u64 gl_mul_weak_cs(u64 a, u64 b) {
    u64 lo, hi;
    mul_wide(a, b, lo, hi);
    return gl_reduce128_weak(lo, hi);
}
GL_HD u64 gl_reduce128_weak_cc(u64 lo, u64 hi) { return gl_reduce128_weak(lo, hi); }
GL_HD u64 gl_reduce128_cc(u64 lo, u64 hi) { return gl_reduce128(lo, hi); }
#endif

GL_HD u64 gl_mul(u64 a, u64 b) {
    u64 lo, hi;
    mul_wide(a, b, lo, hi);
    return gl_reduce128_cc(lo, hi);
}
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

// x * 2^S mod p for a compile-time S in [0, 96), x canonical -> canonical.  Shifts instead of a 64x64 multiply:
// 2 is a 192nd root of unity in this field (2^96 = -1), so every 2^k-th root of unity with k <= 6 is +-2^s and the
// butterflies of a radix-16 transform need no general multiplication at all.
template <int S>
GL_HD u64 gl_mul_pow2(u64 x) {
    static_assert(S >= 0 && S < 96, "shift out of range");
    if constexpr (S == 0) {
        return x;
    } else if constexpr (S < 64) {
        const u64 lo = x << S;
        const u64 hi = x >> (64 - S);
        return gl_reduce128(lo, hi);
    } else if constexpr (S == 64) {
        return gl_reduce128(0, x);
    } else {
        // x*2^S = (x*2^(S-64)) * 2^64 ; with y = ylo + yhi*2^64:  y*2^64 = ylo*2^64 + yhi*2^128 = ylo*2^64 - yhi*2^32
        constexpr int K = S - 64;
        const u64 ylo = x << K;
        const u64 yhi = x >> (64 - K);  // < 2^K <= 2^31
        const u64 r0 = gl_reduce128(0, ylo);
        return gl_sub(r0, yhi << 32);
    }
}
// run-time dispatch on a small set of shifts (multiples of 12); folds away when `s` is a compile-time constant
GL_HD u64 gl_mul_pow2_sw(u64 x, int s) {
    switch (s) {
        case 0: return x;
        case 12: return gl_mul_pow2<12>(x);
        case 24: return gl_mul_pow2<24>(x);
        case 36: return gl_mul_pow2<36>(x);
        case 48: return gl_mul_pow2<48>(x);
        case 60: return gl_mul_pow2<60>(x);
        case 72: return gl_mul_pow2<72>(x);
        default: return gl_mul_pow2<84>(x);
    }
}

GL_HD u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
GL_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }

// primitive 2^k-th root of unity the reference uses (types.rs:240-244)
GL_HD u64 gl_root_of_unity(int k) {
    u64 r = GL_POWER_OF_TWO_GENERATOR;
    for (int i = 0; i < 32 - k; i++) r = gl_mul(r, r);
    return r;
}

GL_HD u32 bitrev32(u32 x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

// ---- quadratic extension F_p[X]/(X^2 - 7)  (goldilocks_extensions.rs:14-28) ----
struct Ext2 {
    u64 a, b;
};
GL_HD Ext2 ext_make(u64 a, u64 b) { Ext2 r; r.a = a; r.b = b; return r; }
GL_HD Ext2 ext_add(Ext2 x, Ext2 y) { return ext_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GL_HD Ext2 ext_sub(Ext2 x, Ext2 y) { return ext_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GL_HD Ext2 ext_mul(Ext2 x, Ext2 y) {
    u64 bb = gl_mul(x.b, y.b);
    u64 bb7 = gl_sub(gl_mul(bb, 8), bb);
    return ext_make(gl_add(gl_mul(x.a, y.a), bb7), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a)));
}
GL_HD Ext2 ext_scalar_mul(Ext2 x, u64 s) { return ext_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
GL_HD bool ext_eq(Ext2 x, Ext2 y) { return x.a == y.a && x.b == y.b; }
GL_HD Ext2 ext_inv(Ext2 x) {
    u64 bb = gl_mul(x.b, x.b);
    u64 n = gl_sub(gl_mul(x.a, x.a), gl_sub(gl_mul(bb, 8), bb));
    u64 ni = gl_inv(n);
    return ext_make(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}
GL_HD Ext2 ext_pow(Ext2 b, u64 e) {
    Ext2 r = ext_make(1, 0);
    while (e) {
        if (e & 1) r = ext_mul(r, b);
        b = ext_mul(b, b);
        e >>= 1;
    }
    return r;
}

}  // namespace ola
