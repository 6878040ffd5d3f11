// CPU check of the T-form limb arithmetic the transform passes run on (olavm_amd/csrc/tform.cuh is host + device code):
// every primitive against canonical Goldilocks arithmetic (gl.cuh's host forms) on limb vectors that sit on the magnitude bounds,
// and the in-register radix-16 / radix-8 blocks against a naive DFT with the REFERENCE's roots of unity
// (plonky2/field/src/types.rs:240-244: w_16 = g^(2^28), g = 1753635133440165772) in the reference's decimation-in-frequency
// output order (cfft/serial.rs:89-152: slot j holds X[bitrev(j)]).  Built and run by tests/test_host_api.py (no GPU needed).
//   g++ -std=c++17 -O1 -Wno-unknown-pragmas -Iolavm_amd/csrc tests/host_tform_check.cpp -o host_tform_check && ./host_tform_check
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tform.cuh"

using namespace ola;

static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() {
    u64 x = (rng_state += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static u64 tf_value(const T4& x) {   // sum v_i 2^(24 i) mod p, the slow way
    u64 r = 0;
    for (int i = 3; i >= 0; --i) {
        r = gl_mul(r, 1ull << 24);
        const i64 v = x.v[i];
        r = v >= 0 ? gl_add(r, (u64)v) : gl_sub(r, (u64)(-v));
    }
    return r;
}
static i32 limb(int shape, i32 bound) {   // |result| <= bound
    const u64 h = rnd();
    switch (shape % 7) {
        case 0: return (i32)(h % (2ull * bound + 1)) - bound;
        case 1: return bound - (i32)(h & 0xFF);
        case 2: return -bound + (i32)(h & 0xFF);
        case 3: return (i32)(h & 0xFFFFFF) % (bound + 1);
        case 4: return (i32)(0xFFFFFF - (h & 0xF)) % (bound + 1);
        case 5: return -((i32)(h & 0xFFFFFF) % (bound + 1));
        default: return (i32)(h & 3) - 1;
    }
}
static T4 vec(i32 bound) {
    T4 x;
    const u64 h = rnd();
    for (int i = 0; i < 4; i++) x.v[i] = limb((int)(h >> (8 * i)), bound);
    return x;
}
static int fails = 0;
#define CHECK(cond, what) do { if (!(cond)) { if (fails < 10) fprintf(stderr, "FAIL: %s (line %d)\n", what, __LINE__); fails++; } } while (0)

template <int S> static void check_shift(const T4& a, const T4& b) {
    const u64 d = gl_sub(tf_value(a), tf_value(b));
    const u64 want = S >= 96 ? gl_neg(gl_mul_pow2<S % 96>(d)) : gl_mul_pow2<S % 96>(d);
    const T4 y = tf_sub_mul_pow2<S>(a, b);
    CHECK(tf_value(y) == want, "tf_sub_mul_pow2 value");
    for (int i = 0; i < 4; i++) {
        const i64 bound = (S % 24) ? ((1ll << 24) + (1ll << 30) / (1ll << (24 - S % 24)) + 2) : (1ll << 30);
        CHECK(y.v[i] <= bound && y.v[i] >= -bound, "tf_sub_mul_pow2 magnitude");
    }
}
template <int K, bool INV> static void check_dft() {
    // inputs as a pass feeds them: |limb| < 2^25.2; outputs must stay below the fold's bound 2^31 - 2^8
    T4 x[1 << K];
    u64 in[1 << K];
    for (int j = 0; j < (1 << K); j++) { x[j] = vec((1 << 25) + (1 << 22)); in[j] = tf_value(x[j]); }
    tf_dft<K, INV>(x);
    u64 w = gl_root_of_unity(K);
    if (INV) w = gl_inv(w);
    for (int j = 0; j < (1 << K); j++) {
        const u32 q = bitrev32((u32)j, K);
        u64 acc = 0;
        for (int m = 0; m < (1 << K); m++) acc = gl_add(acc, gl_mul(in[m], gl_pow(w, (u64)m * q)));
        CHECK(tf_value(x[j]) == acc, "tf_dft slot j != X[bitrev(j)]");
        for (int i = 0; i < 4; i++) CHECK(x[j].v[i] < (1ll << 31) - 256 && x[j].v[i] > -((1ll << 31) - 256), "tf_dft output magnitude");
    }
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 200000;
    CHECK(gl_pow(2, 96) == GL_P - 1, "2^96 = -1");
    for (int k = 1; k <= 6; k++) {   // the exponents of two the blocks use ARE the reference's roots
        CHECK(gl_pow(2, tf_root_exp(k, false)) == gl_root_of_unity(k), "tf_root_exp forward");
        CHECK(gl_mul(gl_pow(2, tf_root_exp(k, true)), gl_root_of_unity(k)) == 1, "tf_root_exp inverse");
    }
    for (long it = 0; it < n; it++) {
        const T4 x = vec((1 << 30) - 1);
        const u64 want = tf_value(x);
        CHECK(tf_to_u64<true>(x) == want, "tf_to_u64<canonical>");
        CHECK(gl_canon(tf_to_u64<false>(x)) == want, "tf_to_u64<weak>");
        const T4 nx = tf_norm(x);
        CHECK(tf_value(nx) == want, "tf_norm value");
        for (int i = 0; i < 4; i++) CHECK(nx.v[i] > -130 && nx.v[i] < (1 << 24) + 130, "tf_norm magnitude");
        u64 a = rnd(), b = rnd();
        if (it % 5 == 0) a |= 0xFFFFFFFF00000000ull;
        if (it % 7 == 0) b = ~0ull - (b & 0xFF);
        CHECK(tf_value(tf_from_u64(a)) == gl_canon(a), "tf_from_u64");
        u64 lo, hi;
        mul_wide(a, b, lo, hi);
        const T4 pr = tf_from_u128(lo, hi);
        CHECK(tf_value(pr) == gl_mul(gl_canon(a), gl_canon(b)), "tf_from_u128");
        for (int i = 0; i < 4; i++) CHECK(pr.v[i] > -(1 << 24) && pr.v[i] < (1 << 24), "tf_from_u128 magnitude");
        const T4 y = vec((1 << 29) - 1);
        const u64 w = gl_canon(b);
        const T4 m = tf_mul(y, tf_split_u64(w));
        CHECK(tf_value(m) == gl_mul(tf_value(y), w), "tf_mul value");
        for (int i = 0; i < 4; i++) CHECK(m.v[i] > -((1 << 25) + 600) && m.v[i] < (1 << 25) + 600, "tf_mul magnitude");
        const T4 p = vec((1 << 28) - 1), q = vec((1 << 28) - 1);
        CHECK(tf_value(tf_add(p, q)) == gl_add(tf_value(p), tf_value(q)), "tf_add");
        switch (it % 16) {
#define C(S) case (S) / 12: check_shift<S>(p, q); break;
            C(0) C(12) C(24) C(36) C(48) C(60) C(72) C(84) C(96) C(108) C(120) C(132) C(144) C(156) C(168) C(180)
#undef C
        }
        if (it % 64 == 0) { check_dft<4, false>(); check_dft<4, true>(); check_dft<3, false>(); check_dft<3, true>(); check_dft<2, false>(); check_dft<1, true>(); }
    }
    if (fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
    printf("tform ok: %ld samples\n", n);
    return 0;
}
