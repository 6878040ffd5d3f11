// Host-side check of product header functions (olavm_amd/csrc/gl.cuh is host+device code) against plain 128-bit
// arithmetic.  Built and run by tests/test_host_code.py; prints "ok" on success.
#include <cstdio>
#include <cstdlib>

#include "../olavm_amd/csrc/gl.cuh"
#include "../olavm_amd/csrc/poseidon_host.h"

using namespace ola;
typedef unsigned __int128 u128;

static u64 ref_mul(u64 a, u64 b) { return (u64)(((u128)a * b) % GL_P); }
static u64 rnd() { static u64 s = 88172645463325252ull; s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

template <int S>
static int check_shift(const u64* xs, int n) {
    u64 c = 1;
    for (int i = 0; i < S; i++) c = ref_mul(c, 2);
    for (int i = 0; i < n; i++)
        if (gl_mul_pow2<S>(xs[i]) != ref_mul(xs[i], c)) { printf("mul_pow2<%d> mismatch at %llx\n", S, xs[i]); return 1; }
    return 0;
}

int main() {
    const int N = 20000;
    u64* xs = (u64*)malloc(N * 8);
    const u64 edge[] = {0, 1, GL_P - 1, GL_P - 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 1ull << 63, 7};
    int n = 0;
    for (u64 e : edge) xs[n++] = e;
    while (n < N) xs[n++] = rnd() % GL_P;
    int bad = 0;
    bad |= check_shift<12>(xs, N); bad |= check_shift<24>(xs, N); bad |= check_shift<36>(xs, N); bad |= check_shift<48>(xs, N);
    bad |= check_shift<60>(xs, N); bad |= check_shift<72>(xs, N); bad |= check_shift<84>(xs, N);
    bad |= check_shift<1>(xs, N); bad |= check_shift<32>(xs, N); bad |= check_shift<63>(xs, N); bad |= check_shift<64>(xs, N);
    bad |= check_shift<95>(xs, N);
    for (int i = 0; i + 1 < N; i += 2) {
        u64 a = xs[i], b = xs[i + 1];
        if (gl_mul(a, b) != ref_mul(a, b)) { printf("gl_mul mismatch\n"); bad = 1; break; }
        if (gl_add(a, b) != (u64)(((u128)a + b) % GL_P)) { printf("gl_add mismatch\n"); bad = 1; break; }
        if (gl_sub(a, b) != (u64)(((u128)a + GL_P - b) % GL_P)) { printf("gl_sub mismatch\n"); bad = 1; break; }
    }
    // roots of unity are the powers of two the radix-16 butterflies assume
    if (gl_root_of_unity(4) != ref_mul(1, (u64)0) + gl_pow(2, 156)) { printf("w16 != 2^156\n"); bad = 1; }
    if (gl_inv(gl_root_of_unity(4)) != gl_pow(2, 36)) { printf("w16^-1 != 2^36\n"); bad = 1; }
    // host Poseidon (transcript) KAT: all-zero input (poseidon_goldilocks.rs:297-300)
    u64 st[12] = {0};
    poseidon_permute_host(st);
    if (st[0] != 0x3c18a9786cb0b359ull || st[11] != 0x1792b1c4342109d7ull) { printf("host poseidon KAT mismatch\n"); bad = 1; }
    if (!bad) printf("ok\n");
    return bad;
}
