// "T-form" Goldilocks arithmetic of the transform passes (ntt2t.cuh), host + gfx950 device.
//
// p = 2^64 - 2^32 + 1 divides 2^96 + 1, so with T = 2^24 every field element can be written
//     x = v0 + v1*T + v2*T^2 + v3*T^3   (mod p),   T^4 = -1,
// with SIGNED 32-bit limbs: redundant (many limb vectors per element), 24 bits of payload, 7 of headroom.  Every function states
// the limb magnitude it needs and the magnitude it returns.  The functions are plain C++ apart from two device-only details --
// the carry chain of tf_to_u64 (inline assembly with a C++ twin) and tf_pin -- so the same code runs on the host:
// tests/host_tform_check.cpp checks every primitive and the radix-16 / radix-8 blocks against canonical arithmetic and a naive
// DFT in the CPU suite, the device self-test (selftest.hip) does the same on the GPU.
// Reference semantics: plonky2/field/src/goldilocks_field.rs:191-355 (the value mod p is what counts), cfft/serial.rs:89-152 (the
// butterflies these blocks compute).
#pragma once
#include "gl.cuh"

namespace ola {

typedef int i32;
typedef long long i64;

GL_HD u32 tf_alignbit(u32 hi, u32 lo, int s) {   // bits [s, s + 32) of hi:lo
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (u32)((((u64)hi << 32) | lo) >> s);
#endif
}

struct T4 {
    i32 v[4];
};
struct TfTw {   // a twiddle as its 24 / 24 / 16-bit pieces
    i32 w0, w1, w2, pad;
};

// Make the compiler finish the four limbs HERE.  Without it the scheduler keeps the partial terms of a multiplication alive and
// sums them where the result is consumed (three times the registers); it also hides what is known about the limbs' bits: when
// both factors of a multiplication are known to fit 24 bits, hipcc 7.2 forms 24-bit multiplies, drops the operand masks they
// make redundant and then re-combines some of them into v_mad_u64_u32 on the UNMASKED words (wrong products, found in round 2).
GL_HD void tf_pin(T4& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(y.v[0]), "+v"(y.v[1]), "+v"(y.v[2]), "+v"(y.v[3]));
#else
    (void)y;
#endif
}

GL_HD TfTw tf_split_u64(u64 w) {
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    TfTw t;
    t.w0 = (i32)(lo & 0xFFFFFFu);
    t.w1 = (i32)(((lo >> 24) | (hi << 8)) & 0xFFFFFFu);
    t.w2 = (i32)(hi >> 16);
    t.pad = 0;
    return t;
}

// any u64 -> limbs in [0, 2^24) (v2 < 2^16, v3 = 0)
GL_HD T4 tf_from_u64(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    T4 r;
    r.v[0] = (i32)(lo & 0xFFFFFFu);
    r.v[1] = (i32)(tf_alignbit(hi, lo, 24) & 0xFFFFFFu);
    r.v[2] = (i32)(hi >> 16);
    r.v[3] = 0;
    return r;
}
// a 128-bit product lo + hi * 2^64 -> limbs in (-2^24, 2^24):  2^96 = -1 takes the top word down to the bottom
GL_HD T4 tf_from_u128(u64 lo, u64 hi) {
    const u32 l0 = (u32)lo, l1 = (u32)(lo >> 32), h0 = (u32)hi, h1 = (u32)(hi >> 32);
    T4 r;
    r.v[0] = (i32)(l0 & 0xFFFFFFu) - (i32)(h1 & 0xFFFFFFu);
    r.v[1] = (i32)(tf_alignbit(l1, l0, 24) & 0xFFFFFFu) - (i32)(h1 >> 24);
    r.v[2] = (i32)(tf_alignbit(h0, l1, 16) & 0xFFFFFFu);
    r.v[3] = (i32)(h0 >> 8);
    return r;
}

// |limbs| < 2^31 - 2^8  ->  u64.  A limb vector congruent to zero with every limb near 2^31,
//   2^7 * [(2^24 - T) + (2^24 - T)*T + (2^24 - T)*T^2 + (1 + 2^24*T^3)] = (2^31+2^7, 2^31-2^7, 2^31-2^7, 2^31-2^7),
// makes all limbs non-negative 32-bit numbers u_i.  One carry step 1 -> 2 -> 3 -> (T^4 = -1) 0 brings limbs 1..3 below 2^24 while
// u0 stays a positive 32-bit number (it gives up less than 2^9).  Then
//     x = [u0 + n1 2^24 + (n2 mod 2^16) 2^48]  +  2^64 [(n2 >> 16) + n3 2^8],
// a 64-bit word plus a 32-bit multiple of 2^64 = 2^32 - 1: one multiply-add, and two carries that are each worth 2^32 - 1 and
// cannot repeat (a sum that wrapped is small).  CANON: canonical word, else any representative.
template <bool CANON>
GL_HD u64 tf_to_u64(const T4& x) {
    u32 u0 = (u32)x.v[0] + 0x80000080u, u1 = (u32)x.v[1] + 0x7FFFFF80u;
    u32 u2 = (u32)x.v[2] + 0x7FFFFF80u, u3 = (u32)x.v[3] + 0x7FFFFF80u;
    const u32 n1 = u1 & 0xFFFFFFu;
    u2 += u1 >> 24;
    const u32 n2 = u2 & 0xFFFFFFu;
    u3 += u2 >> 24;
    const u32 n3 = u3 & 0xFFFFFFu;
    u0 -= u3 >> 24;
    const u64 A = (u64)n1 * 0x1000000ull + u0;        // < 2^48 + 2^32
    const u32 chi = (n3 << 8) | (n2 >> 16);
    const u32 d16 = n2 << 16;                         // bits 48..63
#if defined(OLA_GL_ASM)
    const u32 a_lo = (u32)A, a_hi = (u32)(A >> 32);
    u32 hi1, e1, e2;
    u64 cA, cB, t2;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(hi1), "=s"(cA) : "v"(a_hi), "v"(d16));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e1) : "s"(cA));
    const u64 lo = (((u64)hi1 << 32) | a_lo) + e1;    // a wrapped sum is below 2^49: + (2^32 - 1) cannot wrap
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t2), "=s"(cB) : "v"(chi), "v"(lo));
    if (!CANON) {
        asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e2) : "s"(cB));
        return t2 + e2;                               // chi * (2^32 - 1) <= 2^64 - 2^33 + 1: a wrapped sum + (2^32 - 1) stays below 2^64
    }
    // canonical: + EPS when the sum wrapped or when it is >= p (gl_reduce128_cc's tail)
    const u32 tl = (u32)t2, th = (u32)(t2 >> 32);
    u32 ul, uh, r0, r1;
    u64 c1, c2, m;
    asm("v_add_co_u32_e64 %0, %1, -1, %2" : "=v"(ul), "=s"(c1) : "v"(tl));
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(uh), "=s"(c2) : "v"(th), "s"(c1));
    asm("s_nop 1\n\ts_or_b64 %2, %3, %4\n\tv_cndmask_b32_e64 %0, %5, %6, %2\n\tv_cndmask_b32_e64 %1, %7, %8, %2"
        : "=&v"(r0), "=&v"(r1), "=&s"(m) : "s"(cB), "s"(c2), "v"(tl), "v"(ul), "v"(th), "v"(uh) : "scc");
    return ((u64)r1 << 32) | r0;
#else
    u64 lo = A + ((u64)d16 << 32);
    if (lo < A) lo += GL_EPS;
    const u64 prod = (u64)chi * GL_EPS;
    u64 t2 = lo + prod;
    if (t2 < lo) t2 += GL_EPS;
    return CANON ? gl_canon(t2) : t2;
#endif
}

GL_HD T4 tf_add(const T4& a, const T4& b) {
    T4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}

// one carry step: same value, |limbs| < 2^31 in, limbs in (-2^7 - 1, 2^24 + 2^7) out
GL_HD T4 tf_norm(const T4& x) {
    const i32 mask = 0xFFFFFF;
    T4 y;
    y.v[0] = (x.v[0] & mask) - (x.v[3] >> 24);
    y.v[1] = (x.v[1] & mask) + (x.v[0] >> 24);
    y.v[2] = (x.v[2] & mask) + (x.v[1] >> 24);
    y.v[3] = (x.v[3] & mask) + (x.v[2] >> 24);
    return y;
}

// (a - b) * 2^S for a compile-time S in [0, 192).  S = 24q + r: the sign of T^4 is taken at the subtraction, the rotation is a
// renaming, and only r != 0 costs instructions:  y_i = ((d_i mod 2^(24-r)) << r) + floor(d_(i-1) / 2^(24-r)),  d_(-1) = -d_3.
// Needs |a_i - b_i| < 2^31; returns |y_i| < 2^24 + |d|/2^(24-r)  (r != 0)  or |d| (r = 0).
template <int S>
GL_HD T4 tf_sub_mul_pow2(const T4& a, const T4& b) {
    static_assert(S >= 0 && S < 192, "shift out of range");
    constexpr int q8 = S / 24, r = S % 24, q = q8 & 3;
    constexpr bool neg = q8 >= 4;
    // limb i of the result comes from limb (i - q) mod 4 of the difference, negated when it wrapped: take that sign at the
    // subtraction, so that the rotation is a renaming
    T4 o;
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool flip = neg != (i + q >= 4);
            o.v[(i + q) & 3] = flip ? (b.v[i] - a.v[i]) : (a.v[i] - b.v[i]);
        }
    } else {
        constexpr int k = 24 - r;
        const i32 mask = (i32)((1 << k) - 1);
        T4 d;
#pragma unroll
        for (int i = 0; i < 4; i++) d.v[i] = neg ? (b.v[i] - a.v[i]) : (a.v[i] - b.v[i]);
        T4 y;
        y.v[0] = ((d.v[0] & mask) << r) - (d.v[3] >> k);
        y.v[1] = ((d.v[1] & mask) << r) + (d.v[0] >> k);
        y.v[2] = ((d.v[2] & mask) << r) + (d.v[1] >> k);
        y.v[3] = ((d.v[3] & mask) << r) + (d.v[2] >> k);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i + q >= 4) o.v[(i + q) & 3] = -y.v[i];
            else o.v[i + q] = y.v[i];
        }
    }
    return o;
}

// x * w for a twiddle given as its pieces.  Needs |x_i| < 2^31 (the four 64-bit sums stay below 2^57); returns
// |y_i| < 2^24 + 2^24 + 2^9.
GL_HD T4 tf_mul(const T4& x_in, const TfTw& w) {
    T4 x = x_in;
    tf_pin(x);
    const i32 w0 = w.w0, w1 = w.w1, w2 = w.w2;
    const i32 n2 = -x.v[2], n3 = -x.v[3];
    // z_k = sum_{i+j=k} x_i w_j - sum_{i+j=k+4} x_i w_j
    i64 z[4];
    z[0] = (i64)x.v[0] * w0 + (i64)n3 * w1 + (i64)n2 * w2;
    z[1] = (i64)x.v[0] * w1 + (i64)x.v[1] * w0 + (i64)n3 * w2;
    z[2] = (i64)x.v[0] * w2 + (i64)x.v[1] * w1 + (i64)x.v[2] * w0;
    z[3] = (i64)x.v[1] * w2 + (i64)x.v[2] * w1 + (i64)x.v[3] * w0;
    // z = l + m*2^24 + h*2^48  (l, m in [0, 2^24), h signed)
    i32 l[4], m[4], h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 lo = (u32)(u64)z[k];
        const i32 hi = (i32)(z[k] >> 32);
        l[k] = (i32)(lo & 0xFFFFFFu);
        m[k] = (i32)(tf_alignbit((u32)hi, lo, 24) & 0xFFFFFFu);
        h[k] = hi >> 16;
    }
    T4 y;
    y.v[0] = l[0] - m[3] - h[2];
    y.v[1] = l[1] + m[0] - h[3];
    y.v[2] = l[2] + m[1] + h[0];
    y.v[3] = l[3] + m[2] + h[1];
    tf_pin(y);
    return y;
}

// exponent of two of the reference's primitive 2^k-th root of unity (types.rs:240-244): w_64 = 2^39, w_32 = 2^78, w_16 = 2^156,
// w_8 = 2^120, w_4 = 2^48, w_2 = 2^96
GL_HD constexpr int tf_root_exp(int k, bool inv) {
    int e = 39;
    for (int i = k; i < 6; i++) e = (2 * e) % 192;
    return inv ? (192 - e) % 192 : e;
}

// In-register decimation-in-frequency transform of 2^K values, all twiddles powers of two; output index j holds X[bitrev_K(j)].
// Every level doubles the limb bound except where a shift re-normalises: inputs below 2^26 and K <= 4 keep the outputs below 2^30.
template <int K, bool INV>
GL_HD void tf_dft(T4* x) {
    static_assert(K >= 0 && K <= 4, "radix");
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
#pragma unroll
        for (int j = 0; j < (1 << K); ++j) {
            if (j & (1 << i)) continue;
            // twiddle w_{2^(i+1)}^(j mod 2^i) = w_16^((j mod 2^i) << (3 - i))
            const int e = (j & ((1 << i) - 1)) << (3 - i);
            const int s = (tf_root_exp(4, INV) * e) % 192;
            const T4 a = x[j], b = x[j + (1 << i)];
            x[j] = tf_add(a, b);
            switch (s) {   // folds: s is a compile-time constant after unrolling, a multiple of 12
#define OLA_TF_CASE(S) case S: x[j + (1 << i)] = tf_sub_mul_pow2<S>(a, b); break;
                OLA_TF_CASE(0) OLA_TF_CASE(12) OLA_TF_CASE(24) OLA_TF_CASE(36) OLA_TF_CASE(48) OLA_TF_CASE(60) OLA_TF_CASE(72) OLA_TF_CASE(84)
                OLA_TF_CASE(96) OLA_TF_CASE(108) OLA_TF_CASE(120) OLA_TF_CASE(132) OLA_TF_CASE(144) OLA_TF_CASE(156) OLA_TF_CASE(168)
                OLA_TF_CASE(180)
#undef OLA_TF_CASE
                default: break;
            }
        }
    }
}

}  // namespace ola
