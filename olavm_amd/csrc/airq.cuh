// Shared by stark.hip and the generated per-table quotient kernels (gen/airq_*.hip, printed by
// olavm_amd/air/codegen.py): kernel parameters, the lazy constraint accumulator and the macros the generated
// straight-line code is written in.
#pragma once
#include <hip/hip_runtime.h>

#include "gl.cuh"

namespace ola {

struct QuotParams {
    const u64* trace_lde; const u64* zs_lde; const u64* lag_lde;  // leaf order, column stride N
    size_t N, n;
    int log_n, log_N, qdb;
    const u64* gN_lo; const u64* gN_hi; int gN_h;   // two-level powers of the order-N root
    const u64* desc;                                // per-proof descriptor (layout in build_quot_desc)
    u64 g_inv;                                      // last = g^-1 (prover.rs:617)
    u64* out;                                       // [num_challenges][size]
    int n_regs;
};

// Straight-line kernels printed by olavm_amd/air/codegen.py, one per table signature.  Descriptor D (u64 words):
//   [0,8) 1/Z_H per coset | [8, 8+2K) alpha weights alpha_c^(K-1-i), challenge-major | params | permutation (beta,gamma)
//   per batch slot | CTL (beta,gamma) per Z column.
struct Acc160 { u64 lo, hi; u32 top; };   // sum of < 2^32 products of two u64
__device__ __forceinline__ void acc_mad(Acc160& a, u64 x, u64 w) {
    u64 plo, phi;
    mul_wide(x, w, plo, phi);
    a.lo += plo;
    phi += (a.lo < plo) ? 1ull : 0ull;     // phi <= 2^64 - 2, cannot wrap
    a.hi += phi;
    a.top += (a.hi < phi) ? 1u : 0u;
}
// lo + hi*2^64 + top*2^128 mod p, with 2^128 = -2^32 (mod p) and top*2^32 <= p - 1
__device__ __forceinline__ u64 acc_reduce(const Acc160& a) { return gl_sub(gl_reduce128(a.lo, a.hi), (u64)a.top << 32); }

#define AIRQ_THREADS 256
#ifdef AIRQ_GENERATED_TU
#define AIRQ_PROLOGUE(K_)                                                                                                  \
    const size_t size = P.n << P.qdb;                                                                                      \
    const size_t j = (size_t)blockIdx.x * AIRQ_THREADS + threadIdx.x;                                                      \
    const bool active = j < size;                                                                                          \
    const size_t jj = active ? j : 0;                                                                                      \
    const size_t c_ = jj >> P.log_n, r_ = jj & (P.n - 1);                                                                  \
    const u32 rr_ = bitrev32((u32)r_, P.log_n);                                                                            \
    const size_t jn = (c_ << P.log_n) + bitrev32((rr_ + 1) & (u32)(P.n - 1), P.log_n);                                     \
    const u64 m_ = ((u64)rr_ << (P.log_N - P.log_n)) + bitrev32((u32)c_, P.log_N - P.log_n);                               \
    const u64 x_ = gl_mul(GL_GENERATOR, gl_mul(P.gN_lo[m_ & (((u64)1 << P.gN_h) - 1)], P.gN_hi[m_ >> P.gN_h]));            \
    const u64 z_last = gl_sub(x_, P.g_inv);                                                                                \
    const size_t N = P.N;                                                                                                  \
    const u64 lag_first = P.lag_lde[jj], lag_last = P.lag_lde[N + jj];                                                     \
    const u64* __restrict__ D = P.desc;                                                                                    \
    const u64* __restrict__ T_ = P.trace_lde;                                                                              \
    const u64* __restrict__ Z_ = P.zs_lde;                                                                                 \
    constexpr int AIRQ_K = (K_);                                                                                           \
    Acc160 accA0 = {0, 0, 0}, accA1 = {0, 0, 0}, accT0 = {0, 0, 0}, accT1 = {0, 0, 0};
#define LC(c) gl_canon(T_[(size_t)(c) * N + jj])
#define NC(c) gl_canon(T_[(size_t)(c) * N + jn])
#define ZL(c) Z_[(size_t)(c) * N + jj]
#define ZN(c) Z_[(size_t)(c) * N + jn]
#define AIRQ_EMIT_ALL(i, v) { const u64 v_ = (v); acc_mad(accA0, v_, D[8 + (i)]); acc_mad(accA1, v_, D[8 + AIRQ_K + (i)]); }
#define AIRQ_EMIT_TRANS(i, v) { const u64 v_ = (v); acc_mad(accT0, v_, D[8 + (i)]); acc_mad(accT1, v_, D[8 + AIRQ_K + (i)]); }
#define AIRQ_EPILOGUE                                                                                                      \
    if (active) {                                                                                                          \
        const u64 zh_inv = D[c_];                                                                                          \
        P.out[j] = gl_mul(gl_add(acc_reduce(accA0), gl_mul(z_last, acc_reduce(accT0))), zh_inv);                           \
        P.out[size + j] = gl_mul(gl_add(acc_reduce(accA1), gl_mul(z_last, acc_reduce(accT1))), zh_inv);                    \
    }

#endif  // AIRQ_GENERATED_TU

struct AirKernelEntry { u64 signature; void (*kernel)(QuotParams); int n_emits, n_params, n_perm; const char* name; };


}  // namespace ola
