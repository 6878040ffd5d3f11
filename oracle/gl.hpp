// ORACLE -- test infrastructure only.  CPU restatement of the reference's Goldilocks arithmetic.
// Nothing under olavm_amd/ may include, link or call this file; it exists to CHECK the HIP path.
//
// Follows (reference paths relative to /root/reference):
//   plonky2/field/src/goldilocks_field.rs:14,126,191-355   (p = 2^64 - 2^32 + 1, add/sub/mul/reduce128)
//   plonky2/field/src/types.rs:199-244,430                 (inverse_2exp, primitive_root_of_unity, coset_shift)
//   plonky2/field/src/goldilocks_extensions.rs:14-28       (quadratic extension, W = 7)
//   plonky2/field/src/extension/quadratic.rs               (ext mul / inverse via conjugate)
// Unlike the reference, every value here is kept canonical (< p) at all times; since all arithmetic is
// exact mod p, canonical outputs are identical to the reference's canonicalised outputs.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace ola_oracle {

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 GL_P = 0xFFFFFFFF00000001ull;
static const u64 GL_GENERATOR = 7;                        // MULTIPLICATIVE_GROUP_GENERATOR == coset_shift()
static const u64 GL_POWER_OF_TWO_GENERATOR = 1753635133440165772ull;  // order 2^32
static const int GL_TWO_ADICITY = 32;

static inline u64 gl_canon(u64 x) { return x >= GL_P ? x - GL_P : x; }
static inline u64 gl_add(u64 a, u64 b) { u128 s = (u128)a + b; return (u64)(s >= GL_P ? s - GL_P : s); }
static inline u64 gl_sub(u64 a, u64 b) { return a >= b ? a - b : a + (GL_P - b); }
static inline u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
// goldilocks_field.rs:329-345 reduce128 (2^64 = 2^32 - 1, 2^96 = -1 mod p), followed by canonicalisation
static inline u64 gl_reduce128(u128 x) {
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFull;
    u64 t1 = hl * 0xFFFFFFFFull;
    u64 t2 = t0 + t1;
    if (t2 < t0) t2 += 0xFFFFFFFFull;
    return gl_canon(t2);
}
static inline u64 gl_mul(u64 a, u64 b) { return gl_reduce128((u128)a * b); }
static inline u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); e >>= 1; }
    return r;
}
static inline u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }  // a != 0
// types.rs:240-244
static inline u64 gl_root_of_unity(int n_log) {
    u64 r = GL_POWER_OF_TWO_GENERATOR;
    for (int i = 0; i < GL_TWO_ADICITY - n_log; i++) r = gl_mul(r, r);
    return r;
}
static inline int log2_strict(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }
static inline size_t reverse_bits(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// ---- quadratic extension F_p[X]/(X^2 - 7) ----
struct Ext2 {
    u64 a, b;  // a + b*X
    bool operator==(const Ext2& o) const { return a == o.a && b == o.b; }
    bool operator!=(const Ext2& o) const { return !(*this == o); }
};
static const Ext2 EXT_ZERO = {0, 0};
static const Ext2 EXT_ONE = {1, 0};
static inline Ext2 ext_from(u64 x) { return Ext2{x, 0}; }
static inline Ext2 ext_add(Ext2 x, Ext2 y) { return Ext2{gl_add(x.a, y.a), gl_add(x.b, y.b)}; }
static inline Ext2 ext_sub(Ext2 x, Ext2 y) { return Ext2{gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; }
static inline Ext2 ext_neg(Ext2 x) { return Ext2{gl_neg(x.a), gl_neg(x.b)}; }
static inline Ext2 ext_mul(Ext2 x, Ext2 y) {
    return Ext2{gl_add(gl_mul(x.a, y.a), gl_mul(7, gl_mul(x.b, y.b))),
                gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))};
}
static inline Ext2 ext_scalar_mul(Ext2 x, u64 s) { return Ext2{gl_mul(x.a, s), gl_mul(x.b, s)}; }
static inline Ext2 ext_inv(Ext2 x) {
    // 1/(a+bX) = (a-bX)/(a^2 - 7 b^2)
    u64 n = gl_sub(gl_mul(x.a, x.a), gl_mul(7, gl_mul(x.b, x.b)));
    u64 ni = gl_inv(n);
    return Ext2{gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni)};
}
static inline Ext2 ext_pow(Ext2 b, u64 e) {
    Ext2 r = EXT_ONE;
    while (e) { if (e & 1) r = ext_mul(r, b); b = ext_mul(b, b); e >>= 1; }
    return r;
}

}  // namespace ola_oracle
