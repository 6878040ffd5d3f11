// Device self-test of the field arithmetic (ola_gpu_selftest): the T-form primitives of the transform passes (ntt2t.cuh: fold with
// its two carry fixes, carry step, product-to-limbs cut, table multiplication, every shift of a radix-16 block) against canonical
// arithmetic on limb vectors that sit on the bounds, and the carry-flag forms of the 128 -> 64 bit reduction (gl.cuh, inline
// assembly with hand-placed wait states) against the plain C++ forms of the same header, on every pair of a table of edge values
// and on `pairs` pseudo-random operand pairs biased towards words whose halves are all ones or all zeros -- where the carries
// live; and (round 6) the lazy sums of 22-bit limb products of the quotient and opening-evaluation kernels against canonical
// multiply-and-add, up to the 512 products a sum may hold.  A product check an integrator can run once at start-up on a new driver /
// compiler; the GPU test-suite runs it too.
#include <hip/hip_runtime.h>

#include <vector>

#include "device_ctx.h"
#include "gl.cuh"
#include "ntt2t.cuh"
#include "airq.cuh"

namespace ola {

__device__ __forceinline__ u64 st_mix(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned st_check(u64 x, u64 y) {
    const u64 want = gl_reduce128(x, y);           // C++ form
    unsigned bad = (gl_canon(gl_reduce128_weak_cc(x, y)) != want) + (gl_reduce128_cc(x, y) != want);
    u64 lo, hi;
    mul_wide(x, y, lo, hi);
    bad += gl_mul(x, y) != gl_reduce128(lo, hi);
    bad += gl_canon(gl_mul_weak_cs(x, y)) != gl_reduce128(lo, hi);      // the one-chain product of the Poseidon S-boxes
    return bad;
}
__global__ __launch_bounds__(256) void field_selftest_kernel(const u64* __restrict__ edges, int nedges, u64 per_thread, unsigned long long* __restrict__ bad) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (u64)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (u64 k = gid; k < (u64)nedges * nedges; k += nthreads) local += st_check(edges[k / nedges], edges[k % nedges]);
    u64 s = st_mix(gid * 0xD1342543DE82EF95ull + 0xC0FFEEull);
    for (u64 i = 0; i < per_thread; i++) {
        const u64 x = st_mix(s), y = st_mix(s + 1);
        s = y;
        const u64 xx = (i & 7) == 3 ? (x | 0xFFFFFFFF00000000ull) : (i & 7) == 5 ? (x & 0xFFFFFFFFull) : x;
        const u64 yy = (i & 15) == 9 ? (y | 0xFFFFFFFFull) : (i & 15) == 11 ? (y << 32) : y;
        local += st_check(xx, yy);
    }
    if (local) atomicAdd(bad, local);
}

// ---- the T-form arithmetic of the transform passes (ntt2t.cuh) against canonical arithmetic on the same values --------------
// value of a limb vector, the slow way: sum v_i 2^(24 i) mod p with canonical operations only
__device__ __forceinline__ u64 st_tf_value(const T4& x) {
    u64 r = 0;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        r = gl_mul(r, 1ull << 24);
        const i64 v = x.v[i];
        r = v >= 0 ? gl_add(r, (u64)v) : gl_sub(r, (u64)(-v));
    }
    return r;
}
__device__ __forceinline__ i32 st_limb(u64 h, int shape) {
    // magnitudes up to the bound the passes allow (|v| < 2^30), with the corners where carries and borrows live
    const i32 big = (1 << 30) - 1;
    switch (shape & 7) {
        case 0: return (i32)(h % (2ull * big + 1)) - big;
        case 1: return big - (i32)(h & 0xFF);
        case 2: return -big + (i32)(h & 0xFF);
        case 3: return (i32)(h & 0xFFFFFF);
        case 4: return (i32)(0xFFFFFF - (h & 0xF));
        case 5: return -(i32)(h & 0xFFFFFF);
        case 6: return (i32)((h & 1) ? 0 : ((h >> 1) & 3) - 1);
        default: return (i32)((h & 0xFFFF) << 8) - (i32)((h >> 16) & 0xFFFFFF);
    }
}
__device__ __forceinline__ unsigned st_tf_check(u64 seed) {
    unsigned bad = 0;
    T4 x;
    const u64 h = st_mix(seed);
#pragma unroll
    for (int i = 0; i < 4; i++) x.v[i] = st_limb(st_mix(h + i), (int)(h >> (3 * i + 40)));
    const u64 want = st_tf_value(x);
    // the fold, both forms
    bad += tf_to_u64<true>(x) != want;
    bad += gl_canon(tf_to_u64<false>(x)) != want;
    // one carry step keeps the value
    bad += st_tf_value(tf_norm(x)) != want;
    // u64 -> limbs and 128-bit product -> limbs
    const u64 a = st_mix(h ^ 0xA5A5A5A5ull) | ((h & 1) ? 0xFFFFFFFF00000000ull : 0), b = st_mix(h ^ 0x5A5A5A5Aull);
    bad += st_tf_value(tf_from_u64(a)) != gl_canon(a);
    u64 lo, hi;
    mul_wide(a, b, lo, hi);
    bad += st_tf_value(tf_from_u128(lo, hi)) != gl_mul(gl_canon(a), gl_canon(b));
    // general multiplication by a table twiddle (inputs within the bound the passes feed it: |limb| < 2^29)
    T4 y = x;
#pragma unroll
    for (int i = 0; i < 4; i++) y.v[i] >>= 1;
    const u64 w = gl_canon(b);
    bad += st_tf_value(tf_mul(y, tf_split_u64(w))) != gl_mul(st_tf_value(y), w);
    // (a - b) 2^S for the shifts a radix-16 block uses
    T4 z;
#pragma unroll
    for (int i = 0; i < 4; i++) { y.v[i] >>= 1; z.v[i] = st_limb(st_mix(h + 17 + i), (int)(h >> (3 * i + 20))) >> 2; }
    const u64 d = gl_sub(st_tf_value(y), st_tf_value(z));
#define OLA_ST_SHIFT(S) bad += st_tf_value(tf_sub_mul_pow2<S>(y, z)) != ((S) >= 96 ? gl_neg(gl_mul_pow2<(S) % 96>(d)) : gl_mul_pow2<(S) % 96>(d));
    OLA_ST_SHIFT(0) OLA_ST_SHIFT(12) OLA_ST_SHIFT(24) OLA_ST_SHIFT(36) OLA_ST_SHIFT(48) OLA_ST_SHIFT(60) OLA_ST_SHIFT(72) OLA_ST_SHIFT(84)
    OLA_ST_SHIFT(96) OLA_ST_SHIFT(108) OLA_ST_SHIFT(120) OLA_ST_SHIFT(132) OLA_ST_SHIFT(144) OLA_ST_SHIFT(156) OLA_ST_SHIFT(168) OLA_ST_SHIFT(180)
#undef OLA_ST_SHIFT
    return bad;
}
__global__ __launch_bounds__(256) void tform_selftest_kernel(u64 per_thread, unsigned long long* __restrict__ bad) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long local = 0;
    for (u64 i = 0; i < per_thread; i++) local += st_tf_check(gid * 0x9E3779B97F4A7C15ull + i * 0xD1342543DE82EF95ull);
    if (local) atomicAdd(bad, local);
}

// ---- the lazy sums of 22-bit limb products (round 6: airq.cuh Acc3 = the arithmetic of eval_points_wide_kernel, fri.hip) against
// canonical multiply-and-add: sums of 1 .. 512 products (512 = the most a sum may hold), operands biased towards all-ones words and
// p - 1, whose limbs are the largest -- the sample with 512 maximal products sits exactly on the headroom the design claims.
__device__ __forceinline__ unsigned st_limb_sum_check(u64 h) {
    const int count = (h & 7) == 0 ? 512 : 1 + (int)((h >> 8) % 512);
    const int shape = (int)((h >> 3) & 3);
    Acc3 a = {0, 0, 0};
    u64 want = 0;
    u64 s = st_mix(h);
    for (int i = 0; i < count; i++) {
        u64 x = st_mix(s), w = st_mix(s + 1);
        s = w;
        if (shape == 1) { x = ~0ull; w = GL_P - 1; }                       // every product maximal
        else if (shape == 2) { x |= 0xFFFFFFFF00000000ull; w |= 0xFFFFFFFFull; }
        const u64 wc = gl_canon(w), v = gl_mul(wc, 1ull << 32), M = 0x3FFFFF;   // the host's push_limbs (stark.hip) / make_ext_pows (fri.hip)
        acc3_mad(a, x, (wc & M) | (((wc >> 22) & M) << 32), (wc >> 44) | ((v & M) << 32), ((v >> 22) & M) | ((v >> 44) << 32));
        want = gl_add(want, gl_mul(x, wc));
    }
    unsigned bad = acc3_reduce(a) != want;
    const Acc3 again = acc3_init(acc3_reduce(a));                              // AIRQ_REFOLD_*: fold and restart
    bad += acc3_reduce(again) != want;
    if ((h & 0xFF) == 0) {
        // the corner itself: 512 multiply-accumulates of an all-ones word with six all-ones limbs (no multiplier has them; the sums
        // then hold 1024 x (2^32 - 1)(2^22 - 1) < 2^64 each)
        const u64 M = 0x3FFFFF, all = M | (M << 32);
        Acc3 c = {0, 0, 0};
        for (int i = 0; i < 512; i++) acc3_mad(c, ~0ull, all, all, all);
        const u64 L = gl_add(gl_add(M, gl_mul(M, 1ull << 22)), gl_mul(M, 1ull << 44));     // the value the limbs stand for
        const u64 one = gl_mul(gl_add(0xFFFFFFFFull, 0xFFFFFFFFull), L);                     // x_lo L + x_hi L
        bad += acc3_reduce(c) != gl_mul(one, 512);
    }
    return bad;
}
__global__ __launch_bounds__(256) void limb_sum_selftest_kernel(u64 per_thread, unsigned long long* __restrict__ bad) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long local = 0;
    for (u64 i = 0; i < per_thread; i++) local += st_limb_sum_check(st_mix(gid * 0xD1342543DE82EF95ull + i * 0x9E3779B97F4A7C15ull + 0x5EED));
    if (local) atomicAdd(bad, local);
}

u64 field_selftest(DeviceCtx* ctx, u64 pairs) {
    std::vector<u64> e = {0, 1, 2, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, 0xFFFFFFFEull, GL_P - 1, GL_P, GL_P + 1, GL_P - 2, ~0ull, ~0ull - 1,
                          0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFF00000002ull,
                          0x00000001FFFFFFFFull, 0xFFFFFFFE00000001ull, 0xFFFFFFFE00000000ull};
    for (int s = 1; s < 64; s++) { e.push_back(1ull << s); e.push_back((1ull << s) - 1); e.push_back(~0ull << s); }
    DevBuf mem(ctx);                              // released on every path, errors included
    u64* d_e = mem.alloc(e.size());
    unsigned long long* d_bad = (unsigned long long*)mem.alloc_bytes(8);
    unsigned long long bad = 0;
    HIP_CHECK(hipMemcpyAsync(d_e, e.data(), e.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    const unsigned blocks = 2048;
    hipLaunchKernelGGL(field_selftest_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_e, (int)e.size(), (pairs + blocks * 256 - 1) / (blocks * 256), d_bad);
    // the T-form primitives on a sixteenth as many samples (each sample checks 22 identities)
    hipLaunchKernelGGL(tform_selftest_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (pairs / 16 + blocks * 256 - 1) / (blocks * 256), d_bad);
    // the limb-product sums: one sum of up to 512 products per 4096 pairs asked for (at least one per thread of a small grid)
    hipLaunchKernelGGL(limb_sum_selftest_kernel, dim3(256), dim3(256), 0, ctx->stream, std::max<u64>(1, pairs / 4096 / (256 * 256)), d_bad);
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return bad;
}

}  // namespace ola
