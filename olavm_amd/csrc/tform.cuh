// "T-form" Goldilocks arithmetic for the NTT kernels (host + gfx950 device).
//
// p = 2^64 - 2^32 + 1 divides 2^96 + 1, so with T = 2^24 every field element can be written as a polynomial
//     x = v0 + v1*T + v2*T^2 + v3*T^3   (mod p),   T^4 = -1,
// with SIGNED 32-bit limbs.  The representation is redundant (many limb vectors per field element); any vector whose value is
// congruent is as good as any other.  What this buys on the gfx950 integer pipe, where a canonical 64-bit modular add costs
// 7 half-rate instructions and a multiply by a power of two 12-20:
//   * add / sub are four plain 32-bit adds (full-rate VOP2, no carries, no reduction); limbs may grow by one bit per
//     butterfly level and the 8 bits of headroom above 24 cover a whole radix-32 round;
//   * multiplication by 2^(24k) -- in particular by w_8 = 2^120 and w_4 = 2^48 -- is a limb rotation with sign flips;
//     2^s for other s adds one shift-and-carry step that also re-normalises the limbs;
//   * a general multiplication by a canonical twiddle is twelve v_mad_i64_i32 plus a split of the four 64-bit sums into
//     24-bit pieces, and returns limbs below 2^25 whatever the (sub-2^31) input magnitudes were.
// The reference's field semantics (plonky2/field/src/goldilocks_field.rs:191-355) are unchanged: values enter and leave the
// kernels as canonical u64.  Every function states the limb magnitude it needs and the magnitude it returns; the host test
// (tests/host_ntt3_check.cpp) runs the kernels' code on a range-checked limb type.
#pragma once
#include "gl.cuh"

namespace ola {

typedef int i32;
typedef long long i64;

// Limb traits: I = limb type, wide accumulators are traits::W.  The product code instantiates TfTraits<i32>; the host
// checker instantiates a range-checked class.
template <class I> struct TfTraits;
template <> struct TfTraits<i32> {
    typedef i64 W;
    static GL_HD W mad(i32 a, i32 b, W c) { return (i64)a * (i64)b + c; }
    static GL_HD u32 lo32(W z) { return (u32)(u64)z; }
    static GL_HD i32 hi32(W z) { return (i32)(z >> 32); }
    static GL_HD i32 from_u32(u32 x) { return (i32)x; }
    static GL_HD u32 to_u32_biased(i32 v, u32 bias) { return (u32)v + bias; }
};

template <class I>
struct T4 {
    I v[4];
};

// Device only: make the compiler finish the four limbs HERE.  Without it the scheduler keeps the twelve partial terms of a
// multiplication alive (three per limb) and sums them where the result is consumed -- three times the registers.
template <class I> GL_HD void tf_pin(T4<I>&) {}
GL_HD void tf_pin(T4<i32>& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(y.v[0]), "+v"(y.v[1]), "+v"(y.v[2]), "+v"(y.v[3]));   // not volatile: only a data dependence, free to schedule
#else
    (void)y;
#endif
}

// canonical-or-not u64 -> limbs in [0, 2^24) (v2 < 2^16, v3 = 0)
template <class I>
GL_HD T4<I> tf_from_u64(u64 x) {
    typedef TfTraits<I> Tr;
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    T4<I> r;
    r.v[0] = Tr::from_u32(lo & 0xFFFFFFu);
    r.v[1] = Tr::from_u32(((lo >> 24) | (hi << 8)) & 0xFFFFFFu);
    r.v[2] = Tr::from_u32(hi >> 16);
    r.v[3] = Tr::from_u32(0u);
    return r;
}

// |limbs| < 2^31 - 2^7  ->  canonical u64.  A limb vector congruent to zero with every limb near 2^31,
//   B = 2^7 * [(2^24 - T) + (2^24 - T)*T + (2^24 - T)*T^2 + (1 + 2^24*T^3)] = (2^31+2^7, 2^31-2^7, 2^31-2^7, 2^31-2^7),
// makes all limbs non-negative 32-bit numbers; then x = (u0 + u1*2^24) + (u2 + u3*2^24)*2^48 is a 128-bit integer that the
// ordinary 128-bit reduction folds (2^64 = 2^32 - 1, 2^96 = -1).
template <class I>
GL_HD u64 tf_to_u64(const T4<I>& x) {
    typedef TfTraits<I> Tr;
    const u32 u0 = Tr::to_u32_biased(x.v[0], 0x80000080u), u1 = Tr::to_u32_biased(x.v[1], 0x7FFFFF80u);
    const u32 u2 = Tr::to_u32_biased(x.v[2], 0x7FFFFF80u), u3 = Tr::to_u32_biased(x.v[3], 0x7FFFFF80u);
    const u64 A = (u64)u1 * 0x1000000ull + u0;   // < 2^56 + 2^32
    const u64 C = (u64)u3 * 0x1000000ull + u2;
    // S = A + C * 2^48 as (lo, hi)
    const u64 clo = C << 48, chi = C >> 16;
    const u64 lo = A + clo;
    const u64 hi = chi + (lo < A ? 1u : 0u);
    return gl_reduce128(lo, hi);
}

template <class I>
GL_HD T4<I> tf_add(const T4<I>& a, const T4<I>& b) {
    T4<I> r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
template <class I>
GL_HD T4<I> tf_sub(const T4<I>& a, const T4<I>& b) {
    T4<I> r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}

// one carry step: same value, |limbs| < 2^31 in, limbs in (-2^7 - 1, 2^24 + 2^7) out
template <class I>
GL_HD T4<I> tf_norm(const T4<I>& x) {
    const I mask = (I)0xFFFFFF;
    T4<I> y;
    y.v[0] = (x.v[0] & mask) - (x.v[3] >> 24);
    y.v[1] = (x.v[1] & mask) + (x.v[0] >> 24);
    y.v[2] = (x.v[2] & mask) + (x.v[1] >> 24);
    y.v[3] = (x.v[3] & mask) + (x.v[2] >> 24);
    return y;
}

// (a - b) * 2^S for a compile-time S in [0, 192).  S = 24q + r: the sign of T^4 is taken at the subtraction, the rotation is a
// renaming, and only r != 0 costs instructions:  y_i = ((d_i mod 2^(24-r)) << r) + floor(d_(i-1) / 2^(24-r)),  d_(-1) = -d_3.
// Needs |a_i - b_i| < 2^31; returns |y_i| < 2^24 + |d|/2^(24-r)  (r != 0)  or |d| (r = 0).
template <int S, class I>
GL_HD T4<I> tf_sub_mul_pow2(const T4<I>& a, const T4<I>& b) {
    static_assert(S >= 0 && S < 192, "shift out of range");
    constexpr int q8 = S / 24, r = S % 24, q = q8 & 3;
    constexpr bool neg = q8 >= 4;
    T4<I> d;
#pragma unroll
    for (int i = 0; i < 4; i++) d.v[i] = neg ? (b.v[i] - a.v[i]) : (a.v[i] - b.v[i]);
    T4<I> y;
    if (r == 0) {
        y = d;
    } else {
        constexpr int k = 24 - r;
        const I mask = (I)((1 << k) - 1);
        y.v[0] = ((d.v[0] & mask) << r) - (d.v[3] >> k);
        y.v[1] = ((d.v[1] & mask) << r) + (d.v[0] >> k);
        y.v[2] = ((d.v[2] & mask) << r) + (d.v[1] >> k);
        y.v[3] = ((d.v[3] & mask) << r) + (d.v[2] >> k);
    }
    T4<I> o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i + q >= 4) o.v[(i + q) & 3] = -y.v[i];
        else o.v[i + q] = y.v[i];
    }
    return o;
}

// x * w for a canonical-or-not twiddle w given as its three 24/24/16-bit pieces (tf_split_u64).  Needs |x_i| < 2^31 (the four
// 64-bit sums stay below 2^57); returns |y_i| < 2^24 + 2^24 + 2^9.
struct TfTw {
    i32 w0, w1, w2;
};
GL_HD TfTw tf_split_u64(u64 w) {
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    TfTw t;
    t.w0 = (i32)(lo & 0xFFFFFFu);
    t.w1 = (i32)(((lo >> 24) | (hi << 8)) & 0xFFFFFFu);
    t.w2 = (i32)(hi >> 16);
    return t;
}
template <class I>
GL_HD T4<I> tf_mul(const T4<I>& x_in, const TfTw& w) {
    typedef TfTraits<I> Tr;
    typedef typename Tr::W W;
    // Hide what the compiler knows about the limbs' bits.  When both factors are known to fit 24 bits (an element straight
    // from tf_from_u64), hipcc 7.2 forms 24-bit multiplies, drops the operand masks they make redundant, and then re-combines
    // some of them into v_mad_u64_u32 on the UNMASKED words: wrong products (found by tests/gpu_ntt3_selftest.cpp).
    T4<I> x = x_in;
    tf_pin(x);
    const I w0 = Tr::from_u32((u32)w.w0), w1 = Tr::from_u32((u32)w.w1), w2 = Tr::from_u32((u32)w.w2);
    const I n2 = -x.v[2], n3 = -x.v[3];
    // z_k = sum_{i+j=k} x_i w_j - sum_{i+j=k+4} x_i w_j
    W z0 = Tr::mad(x.v[0], w0, W(0)); z0 = Tr::mad(n3, w1, z0); z0 = Tr::mad(n2, w2, z0);
    W z1 = Tr::mad(x.v[0], w1, W(0)); z1 = Tr::mad(x.v[1], w0, z1); z1 = Tr::mad(n3, w2, z1);
    W z2 = Tr::mad(x.v[0], w2, W(0)); z2 = Tr::mad(x.v[1], w1, z2); z2 = Tr::mad(x.v[2], w0, z2);
    W z3 = Tr::mad(x.v[1], w2, W(0)); z3 = Tr::mad(x.v[2], w1, z3); z3 = Tr::mad(x.v[3], w0, z3);
    // z = l + m*2^24 + h*2^48  (l, m in [0, 2^24), h signed)
    I l[4], m[4], h[4];
    const W zz[4] = {z0, z1, z2, z3};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 lo = Tr::lo32(zz[k]);
        const I hi = Tr::hi32(zz[k]);
        l[k] = Tr::from_u32(lo & 0xFFFFFFu);
        m[k] = Tr::from_u32(((lo >> 24) | (Tr::to_u32_biased(hi, 0u) << 8)) & 0xFFFFFFu);
        h[k] = hi >> 16;
    }
    T4<I> y;
    y.v[0] = l[0] - m[3] - h[2];
    y.v[1] = l[1] + m[0] - h[3];
    y.v[2] = l[2] + m[1] + h[0];
    y.v[3] = l[3] + m[2] + h[1];
    tf_pin(y);
    return y;
}

// exponent of two of the reference's primitive 2^k-th root of unity (types.rs:240-244): w_64 = 2^39, w_32 = 2^78, w_16 = 2^156,
// w_8 = 2^120, w_4 = 2^48, w_2 = 2^96 (checked against gl_root_of_unity by the host test)
GL_HD constexpr int tf_root_exp(int k, bool inv) {
    int e = 39;
    for (int i = k; i < 6; i++) e = (2 * e) % 192;
    return inv ? (192 - e) % 192 : e;
}

// In-register decimation-in-frequency transform of 2^K values x[0], x[STRIDE], ..., all twiddles powers of two;
// output index j holds X[bitrev_K(j)].  Needs |limbs| < 2^(31-K) on entry... precisely: every level doubles the bound
// except where a shift re-normalises; the callers keep inputs below 2^25 and K <= 5, so outputs stay below 2^30.
template <int K, bool INV, int STRIDE, class I>
GL_HD void tf_dft(T4<I>* x) {
    static_assert(K >= 0 && K <= 6, "radix");
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
#pragma unroll
        for (int j = 0; j < (1 << K); ++j) {
            if (j & (1 << i)) continue;
            // twiddle w_{2^(i+1)}^(j mod 2^i) = w_{2^K}^((j mod 2^i) << (K-1-i))
            const int e = (j & ((1 << i) - 1)) << (K - 1 - i);
            const int s = (tf_root_exp(K, INV) * e) % 192;
            const T4<I> a = x[j * STRIDE], b = x[(j + (1 << i)) * STRIDE];
            x[j * STRIDE] = tf_add(a, b);
            switch (s) {   // folds: s is a compile-time constant after unrolling
#define OLA_TF_CASE(S) case S: x[(j + (1 << i)) * STRIDE] = tf_sub_mul_pow2<S>(a, b); break;
                OLA_TF_CASE(0) OLA_TF_CASE(6) OLA_TF_CASE(12) OLA_TF_CASE(18) OLA_TF_CASE(24) OLA_TF_CASE(30) OLA_TF_CASE(36) OLA_TF_CASE(42)
                OLA_TF_CASE(48) OLA_TF_CASE(54) OLA_TF_CASE(60) OLA_TF_CASE(66) OLA_TF_CASE(72) OLA_TF_CASE(78) OLA_TF_CASE(84) OLA_TF_CASE(90)
                OLA_TF_CASE(96) OLA_TF_CASE(102) OLA_TF_CASE(108) OLA_TF_CASE(114) OLA_TF_CASE(120) OLA_TF_CASE(126) OLA_TF_CASE(132)
                OLA_TF_CASE(138) OLA_TF_CASE(144) OLA_TF_CASE(150) OLA_TF_CASE(156) OLA_TF_CASE(162) OLA_TF_CASE(168) OLA_TF_CASE(174)
                OLA_TF_CASE(180) OLA_TF_CASE(186)
#undef OLA_TF_CASE
                default: break;   // K <= 5: every exponent is a multiple of 6
            }
        }
    }
}

}  // namespace ola
