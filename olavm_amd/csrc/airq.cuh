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
    // a rank's share (coset partition): the buffers hold cosets [coset_first, ...), N is their column stride, `out` holds
    // `out_plane` values per challenge, and leaf j of the buffers is global leaf j + coset_first * n
    u32 coset_first;
    size_t out_plane;
    // points this launch evaluates (0: out_plane of them).  A memory-lean proof evaluates one coset per launch into its slice
    // of the full planes: npoints = n, out = planes + coset * n, out_plane = the full plane.
    size_t npoints;
};

// Straight-line kernels printed by olavm_amd/air/codegen.py, one per table signature.  Descriptor D (u64 words):
//   [0,8) 1/Z_H per coset | [8, 8+2K) alpha weights alpha_c^(K-1-i), challenge-major | params | permutation (beta,gamma)
//   per batch slot | per CTL Z column: gamma, beta^0 .. beta^(ncol-1) | limb forms (three words each) of the alpha weights and
//   beta powers in the order the kernel uses them (AirKernelEntry::limb_src).
struct Acc160 { u64 lo, hi; u32 top; };   // sum of < 2^32 products of two u64
__device__ __forceinline__ void acc_mad(Acc160& a, u64 x, u64 w) {
    u64 plo, phi;
    mul_wide(x, w, plo, phi);
#if defined(OLA_GL_ASM) && !defined(OLA_ACC_NO_ASM)
    // the 160-bit addition as ONE carry chain through VCC (hipcc re-derives each carry with a 64-bit compare and a select:
    // 8 instructions for these 5); consecutive VCC producers / consumers need no wait states
    u32 l0 = (u32)a.lo, l1 = (u32)(a.lo >> 32), h0 = (u32)a.hi, h1 = (u32)(a.hi >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %5\n\t"
        "v_addc_co_u32 %1, vcc, %1, %6, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %7, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc"
        : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1), "+v"(a.top)
        : "v"((u32)plo), "v"((u32)(plo >> 32)), "v"((u32)phi), "v"((u32)(phi >> 32))
        : "vcc");
    a.lo = ((u64)l1 << 32) | l0;
    a.hi = ((u64)h1 << 32) | h0;
#else
    a.lo += plo;
    phi += (a.lo < plo) ? 1ull : 0ull;     // phi <= 2^64 - 2, cannot wrap
    a.hi += phi;
    a.top += (a.hi < phi) ? 1u : 0u;
#endif
}
// lo + hi*2^64 + top*2^128 mod p, with 2^128 = -2^32 (mod p) and top*2^32 <= p - 1
__device__ __forceinline__ u64 acc_reduce(const Acc160& a) { return gl_sub(gl_reduce128_cc(a.lo, a.hi), (u64)a.top << 32); }

// Round 6: the same lazy sums when the multiplier w is UNIFORM (the alpha weights of the constraint combination, the beta powers
// of a lookup: descriptor words, read with scalar loads).  The host cuts w and w 2^32 mod p into three 22-bit limbs each
// (six 32-bit words = three descriptor words per multiplier, stark.hip push_limbs); then x w = x_lo w + x_hi (w 2^32) is six
// products below 2^54 that land on three limb positions -- three plain 64-bit sums, six v_mad_u64_u32 per multiply-accumulate with
// the limb in an SGPR, no 128-bit product, no carry chain, no VCC (the 160-bit form: four multiply-adds, five moves and adds to
// assemble the product, a five-instruction carry chain).  A sum holds 2 x 512 products; the generator re-folds before that
// (AIRQ_REFOLD_*).  Same arithmetic as eval_points_wide_kernel (fri.hip).
struct Acc3 { u64 c0, c1, c2; };
__device__ __forceinline__ u64 acc3_step(u32 a, u32 w, u64 c) {
    u64 d = c + (u64)a * w;
    asm("" : "+v"(d));      // pins the order: the optimiser would add the two products of a position first, a third instruction
    return d;
}
__device__ __forceinline__ void acc3_mad(Acc3& a, u64 x, u64 w01, u64 w23, u64 w45) {      // [w0 w1] [w2 v0] [v1 v2], v = limbs of w 2^32 mod p
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    a.c0 = acc3_step(x0, (u32)w01, a.c0);
    a.c1 = acc3_step(x0, (u32)(w01 >> 32), a.c1);
#ifdef OLA_SELFTEST_FAULT_ACC3     // fault injection for ola_gpu_selftest (tools/gpu.sh selftest_fault): one limb off by one for 1 operand in 4096
    a.c2 = acc3_step(x0, (u32)w23 ^ ((x0 & 0xFFFu) == 0 ? 1u : 0u), a.c2);
#else
    a.c2 = acc3_step(x0, (u32)w23, a.c2);
#endif
    a.c0 = acc3_step(x1, (u32)(w23 >> 32), a.c0);
    a.c1 = acc3_step(x1, (u32)w45, a.c1);
    a.c2 = acc3_step(x1, (u32)(w45 >> 32), a.c2);
}
__device__ __forceinline__ Acc3 acc3_init(u64 v) { Acc3 a = {v & 0x3FFFFFull, (v >> 22) & 0x3FFFFFull, v >> 44}; return a; }
// c0 + c1 2^22 + c2 2^44 (below 2^109) mod p, canonical
__device__ __forceinline__ u64 acc3_reduce(const Acc3& a) {
    const unsigned __int128 v = (unsigned __int128)a.c0 + ((unsigned __int128)a.c1 << 22) + ((unsigned __int128)a.c2 << 44);
    return gl_reduce128_cc((u64)v, (u64)(v >> 64));
}

#define AIRQ_THREADS 256
#ifdef AIRQ_GENERATED_TU
typedef const u64 __attribute__((address_space(1)))* airq_gptr;
typedef const u64 __attribute__((address_space(4)))* airq_cptr;
// A workgroup of bs_ = blockDim.x threads (a power of two, min(256, n): the launcher's choice) covers bs_ consecutive leaves of
// one coset, and both the local rows (leaf j) and the next rows (leaf jn, see stark.hip quotient_kernel) of the workgroup are
// bs_-aligned runs -- so every load is "uniform base + small lane offset" and its address lives in SGPRs (global_load ... v_off,
// s[base]) instead of one VGPR pair per column and row.  Tables of fewer than 256 rows run one workgroup per coset (round 6:
// they went to the interpreter kernel before -- 37 to 320 us per 8-row table, 1.5 ms for the Poseidon table's program).
#define AIRQ_PROLOGUE(K_)                                                                                                  \
    const u32 lane_ = threadIdx.x;                                                                                         \
    const u32 bs_ = blockDim.x;                                                                                            \
    const size_t jb_ = (size_t)blockIdx.x * bs_;                                                                           \
    const size_t j = jb_ + lane_;                                                                                          \
    const size_t c_ = (jb_ >> P.log_n) + P.coset_first, r_ = j & (P.n - 1);                                                                  \
    const u32 rr_ = bitrev32((u32)r_, P.log_n);                                                                            \
    const size_t jn_ = ((jb_ >> P.log_n) << P.log_n) + bitrev32((rr_ + 1) & (u32)(P.n - 1), P.log_n);                                    \
    const u32 noff_ = (u32)jn_ & (bs_ - 1);                                                                                \
    const size_t nb_ = ((size_t)__builtin_amdgcn_readfirstlane((u32)(jn_ >> 32)) << 32) |                                  \
                       (__builtin_amdgcn_readfirstlane((u32)jn_) & ~(u32)(bs_ - 1));                                       \
    const u64 m_ = ((u64)rr_ << (P.log_N - P.log_n)) + bitrev32((u32)c_, P.log_N - P.log_n);                               \
    const u64 x_ = gl_mul(GL_GENERATOR, gl_mul(P.gN_lo[m_ & (((u64)1 << P.gN_h) - 1)], P.gN_hi[m_ >> P.gN_h]));            \
    const u64 z_last = gl_sub(x_, P.g_inv);                                                                                \
    const size_t N = P.N;                                                                                                  \
    const u64 lag_first = P.lag_lde[j], lag_last = P.lag_lde[N + j];                                                       \
    airq_cptr D = (airq_cptr)P.desc;                                                                                       \
    airq_gptr T_ = (airq_gptr)P.trace_lde;                                                                                 \
    airq_gptr Z_ = (airq_gptr)P.zs_lde;                                                                                    \
    constexpr int AIRQ_K = (K_);                                                                                           \
    AIRQ_ACC accA0 = {0, 0, 0}, accA1 = {0, 0, 0}, accT0 = {0, 0, 0}, accT1 = {0, 0, 0};
// LDE values are canonical (the NTT writes canonical words)
#ifdef AIRQ_TIMING_ONLY_L2_LOADS
// Timing experiment (wrong results): every workgroup reads the first 2048 points of each column, so all cell loads hit L2 --
// the instruction stream is unchanged, the HBM traffic is gone.  What the kernel would cost if its loads were free.
#define LC(c) (T_ + ((size_t)(c) * N + (jb_ & 0x700)))[lane_]
#define NC(c) (T_ + ((size_t)(c) * N + (nb_ & 0x700)))[noff_]
#define ZL(c) (Z_ + ((size_t)(c) * N + (jb_ & 0x700)))[lane_]
#define ZN(c) (Z_ + ((size_t)(c) * N + (nb_ & 0x700)))[noff_]
#else
#define LC(c) (T_ + ((size_t)(c) * N + jb_))[lane_]
#define NC(c) (T_ + ((size_t)(c) * N + nb_))[noff_]
#define ZL(c) (Z_ + ((size_t)(c) * N + jb_))[lane_]
#define ZN(c) (Z_ + ((size_t)(c) * N + nb_))[noff_]
#endif
// Accumulator interface of the generated code.  AIRQ_LIMBS_AT(o): descriptor word where the limb forms start (alpha weights
// challenge-major, then the beta powers of the lookups in descriptor order); AIRQ_MAD(a, x, wi, li): a += x * (the multiplier whose
// u64 form is D[wi] and whose limb form starts li words into the limb area).  -DAIRQ_ACC160 selects the 160-bit sums of rounds 2 - 5.
#define AIRQ_LIMBS_AT(o) constexpr int AIRQ_L0 = (o); (void)AIRQ_L0
// AIRQ_W(s) loads limb slot s (printed one use ahead of AIRQ_MAD(a, x, wi, s) / AIRQ_EMIT_*(i, s0, s1, v), codegen.py hoist_limb_loads)
#ifdef AIRQ_ACC160
#define AIRQ_ACC Acc160
#define AIRQ_ACC_INIT(v) {(v), 0, 0}
#define AIRQ_W(s)
#define AIRQ_MAD(a, x, wi, s) acc_mad(a, x, D[wi])
#define AIRQ_ACC_REDUCE(a) acc_reduce(a)
#define AIRQ_REFOLD_ALL
#define AIRQ_REFOLD_TRANS
#define AIRQ_ACC_FIRST(a) (a).lo
#else
#define AIRQ_ACC Acc3
#define AIRQ_ACC_INIT(v) acc3_init(v)
#define AIRQ_W(s) const u64 lw##s##_0 = D[AIRQ_L0 + 3 * (s)], lw##s##_1 = D[AIRQ_L0 + 3 * (s) + 1], lw##s##_2 = D[AIRQ_L0 + 3 * (s) + 2]
#define AIRQ_MAD(a, x, wi, s) acc3_mad(a, x, lw##s##_0, lw##s##_1, lw##s##_2)
#define AIRQ_ACC_REDUCE(a) acc3_reduce(a)
#define AIRQ_REFOLD_ALL { accA0 = acc3_init(acc3_reduce(accA0)); accA1 = acc3_init(acc3_reduce(accA1)); }
#define AIRQ_REFOLD_TRANS { accT0 = acc3_init(acc3_reduce(accT0)); accT1 = acc3_init(acc3_reduce(accT1)); }
#define AIRQ_ACC_FIRST(a) (a).c0
#endif
// constraint i (its alpha power), limb slots s0 and s1 (the two challenges)
#define AIRQ_EMIT_ALL(i, s0, s1, v) { const u64 v_ = (v); AIRQ_MAD(accA0, v_, 8 + (i), s0); AIRQ_MAD(accA1, v_, 8 + AIRQ_K + (i), s1); }
#define AIRQ_EMIT_TRANS(i, s0, s1, v) { const u64 v_ = (v); AIRQ_MAD(accT0, v_, 8 + (i), s0); AIRQ_MAD(accT1, v_, 8 + AIRQ_K + (i), s1); }
// Segment boundary: the base pointers and the running accumulators pass through one opaque (empty) volatile asm.  The
// compiler can then neither merge a re-load of a trace cell with the load of an earlier segment, nor start the loads of
// this segment before the emits of the previous one are done -- the live ranges of re-loaded cells stay inside their
// segment.  (A plain memory clobber does not do it: loads through the kernel's read-only arguments are treated as
// invariant and all ~300 of them were hoisted to the top of the kernel.)  The pointers keep their address spaces
// (global for the trace / Z tables, constant for the descriptor, which is read with scalar loads).
#define AIRQ_SEGMENT_BARRIER asm volatile("" : "+s"(T_), "+s"(Z_), "+s"(D), "+v"(AIRQ_ACC_FIRST(accA0)), "+v"(AIRQ_ACC_FIRST(accT0)))
// Cells that many segments read (round 4: the CPU table's kernel fetched 3.2 x its algorithmic bytes, every re-load of a segment
// going out to HBM -- the points in flight touch far more than the L2 holds -- and with its loads served from L2 it ran in
// 44 ms instead of 70).  The most re-read cells are therefore loaded once per point and parked in LDS: AIRQ_CACHE_DECL(S)
// reserves S lane-private 8-byte slots per thread ([slot][lane]: conflict-free, no synchronisation -- a lane only reads what it
// wrote), AIRQ_CACHE_PUT fills one, CL(s) reads it back.  The lane's byte offset passes through the segment barrier like the
// table pointers, so that the LDS reads, too, stay inside their segment.
#define AIRQ_CACHE_DECL(S_) __shared__ u64 cache_[(S_) * AIRQ_THREADS]; u32 coff_ = lane_ * 8u
#define AIRQ_CACHE_PUT(s, v) (*(u64*)((char*)cache_ + coff_ + (s) * (AIRQ_THREADS * 8)) = (v))
#define CL(s) (*(const u64*)((const char*)cache_ + coff_ + (s) * (AIRQ_THREADS * 8)))
#define AIRQ_SEGMENT_BARRIER_C asm volatile("" : "+s"(T_), "+s"(Z_), "+s"(D), "+v"(AIRQ_ACC_FIRST(accA0)), "+v"(AIRQ_ACC_FIRST(accT0)), "+v"(coff_))
#define AIRQ_EPILOGUE                                                                                                      \
    {                                                                                                                      \
        const u64 zh_inv = D[c_];                                                                                          \
        P.out[j] = gl_mul(gl_add(AIRQ_ACC_REDUCE(accA0), gl_mul(z_last, AIRQ_ACC_REDUCE(accT0))), zh_inv);                 \
        P.out[P.out_plane + j] = gl_mul(gl_add(AIRQ_ACC_REDUCE(accA1), gl_mul(z_last, AIRQ_ACC_REDUCE(accT1))), zh_inv);   \
    }

#endif  // AIRQ_GENERATED_TU

// limb_src[s]: the descriptor word whose limb form sits in limb slot s (n_limbs slots of three words, in the order the kernel reads them)
struct AirKernelEntry { u64 signature; void (*kernel)(QuotParams); int n_emits, n_params, n_perm; const char* name; const int* limb_src; int n_limbs; };


}  // namespace ola
