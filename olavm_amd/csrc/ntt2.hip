// Second-generation NTT passes for large transforms (2^14 .. 2^32), gfx950.
//
// Same contract as ntt.hip (reference: plonky2/field/src/cfft/mod.rs:22-231, serial.rs:9-78) with a leaner kernel:
//   * every pass carries 4..8 bits; a 256-thread workgroup owns 4096 elements (2^(8-R) tiles of 2^R x 16), each thread
//     holds 16 values in registers;
//   * a pass is two register rounds: a radix-16 transform straight from the global loads, one LDS exchange, then a
//     radix-2^(R-4) transform whose results go straight to the global stores (the closing bit-reversed pass makes one
//     more LDS trip so that its stores are contiguous);
//   * the butterflies use NO general multiplication: w_16 = 2^156 = -2^60 in this field, so all twiddles inside a
//     radix-16 block are +-2^(12k) and are applied with shifts (gl_mul_pow2);
//   * the only 64x64 multiplications are one "round twiddle" per element between the two rounds (table of 2^R entries in
//     LDS) and one "pass twiddle" per element between passes, generated per thread as base * step^k from two lookups;
//   * every global access is a 128-byte segment (16 consecutive elements).
#include <hip/hip_runtime.h>

#include "device_ctx.h"
#include "gl.cuh"
#include "ntt2t.cuh"   // Ntt2Params, the tile geometry helpers and the T-form pass kernels

namespace ola {

#ifndef NTT2_DEFAULT_GROUP_MB
#define NTT2_DEFAULT_GROUP_MB 0   // Infinity-Cache blocking of ntt2_run (see there); the environment overrides it
#endif
#ifndef NTT2_STRIDED_COLS
#define NTT2_STRIDED_COLS 8   // columns per workgroup of a strided pass (see ntt2_pass_kernel): 8 against 4 is +1.4 % on the NTT, +3 % on the LDE
#endif

// In-register decimation-in-frequency transform of 2^K values; output x[j] = X[bitrev_K(j)].
// Twiddle of stage i for pair (j, j + 2^i): w_{2^(i+1)}^(j mod 2^i) = w_16^((j mod 2^i) << (3 - i)) = +-2^s.
template <int K, bool INV>
__device__ __forceinline__ void dft_pow2(u64* x) {
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
#pragma unroll
        for (int j = 0; j < (1 << K); ++j) {
            if (j & (1 << i)) continue;
            const int e16 = (j & ((1 << i) - 1)) << (3 - i);
            const int t = ((INV ? 36 : 156) * e16) % 192;  // w_16 = 2^156, w_16^-1 = 2^36 ; 2^96 = -1
            const u64 a = x[j], c = x[j + (1 << i)];
            x[j] = gl_add(a, c);
            const u64 d = (t >= 96) ? gl_sub(c, a) : gl_sub(a, c);
            x[j + (1 << i)] = gl_mul_pow2_sw(d, t % 96);
        }
    }
}

// CB = columns per workgroup.  Everything that depends only on the position inside the transform -- the pass twiddles and
// the coset pre-scale weights, 2-3 of the ~5 general multiplications per element of a strided pass -- is computed once
// and reused for the workgroup's CB columns.
template <int R, int MODE, bool INV, int CB, bool PRE>
__device__ __forceinline__ void ntt2_pass_body(const Ntt2Params& p) {
    static_assert(!PRE || MODE == N2_STRIDED, "only the first (strided) pass of a coset transform pre-scales");
    static_assert(R >= 4 && R <= 8, "pass width");
    constexpr int K2 = R - 4;     // bits of the second round
    constexpr int D = 8 - R;      // log2(tiles per workgroup)
    constexpr int G2 = 1 << K2;   // values per second-round group
    constexpr int ROW = (1 << R) + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* lds = reinterpret_cast<u64*>(smem_raw);
    u64* tw1 = lds + ntt2_lds_elems<R, MODE>();

    const int tid = threadIdx.x;
    const u32 blk = blockIdx.x;
    const size_t col0 = (size_t)blockIdx.y * CB, coset = blockIdx.z;
    const int L = p.log_n, lo = p.lo;

    if (R > 4) {
        if (tid < (1 << R)) tw1[tid] = p.tw_r[tid];   // [q1][m_low] for the strided pass
    }

    auto pad1 = [](int e) -> int { return K2 >= 1 ? e + ((e >> (K2 + 4)) << 4) : e; };

    // ---- phase A thread -> element map
    int uA, m_low, tA;
    if (MODE == N2_STRIDED) {
        uA = tid & 15;
        const int rest = tid >> 4;
        m_low = rest & ((1 << K2) - 1);
        tA = rest >> K2;
    } else {
        m_low = tid & ((1 << K2) - 1);
        uA = (tid >> K2) & 15;
        tA = tid >> R;
    }
    size_t a0;       // address of register element j = 0
    size_t jstride;  // address distance between consecutive register elements
    // STRIDED: a0 = a0_uni (same for the whole workgroup, lives in SGPRs) + a0_lane (< 2^(lo+K2+1)); with the column base and
    // j*jstride also uniform, every load/store is "SGPR base + 32-bit lane offset" and needs no per-element address registers
    size_t a0_uni = 0;
    u32 a0_lane = 0;
    {
        const u32 ntile = (blk << D) + tA;
        if (MODE == N2_STRIDED) {
            const u32 lowblks = 1u << (lo - 4);
            const u32 lb = ntile & (lowblks - 1), hi = ntile >> (lo - 4);
            a0 = ((size_t)hi << (lo + R)) + ((size_t)lb << 4) + ((size_t)m_low << lo) + uA;
            jstride = (size_t)1 << (lo + K2);
            // from blk alone (provably uniform): with lo - 4 >= D the workgroup's 2^D tiles share `hi` and are consecutive
            const u32 ntile0 = blk << D;
            a0_uni = ((size_t)(ntile0 >> (lo - 4)) << (lo + R)) + ((size_t)(ntile0 & (lowblks - 1)) << 4);
            a0_lane = ((u32)tA << 4) + ((u32)m_low << lo) + (u32)uA;
        } else if (MODE == N2_BITREV_LAST) {
            a0 = ((((size_t)ntile << 4) + uA) << R) + m_low;
            jstride = (size_t)1 << K2;
        } else {
            const int ub = L - R - 4;
            const size_t row = ((size_t)rev_bits_c(uA, 4) << ub) + (ub ? bitrev32(ntile, ub) : 0);
            a0 = (row << R) + m_low;
            jstride = (size_t)1 << K2;
        }
    }
    // ---- phase B thread -> element map
    const int uB = tid & 15, m_hi = tid >> 4;

    // ---- column-independent multipliers
    // Coset pre-scale (first pass of a coset transform, lo + R = L): the full weight s^k of each of the thread's 16 elements in
    // registers, built once per workgroup from a two-level lookup and the per-coset ratio, reused for its CB columns (194 VGPRs;
    // a factored form at 108 VGPRs was measured 4 % slower in round 3 and is gone: docs/EXPERIMENTS.md)
    constexpr bool prescale = PRE;
    u64 up[PRE ? 16 : 1];
    if (prescale) {
        const u64 sb = two_level(p.sc_lo + coset * p.sc_coset_stride, p.sc_hi + coset * p.sc_coset_stride, p.sc_h, a0);
        const u64 st = p.sc_step[coset];
        u64 w = sb;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            up[j] = w;
            if (j < 15) w = gl_mul(w, st);
        }
    }
    u64 wt[16];   // pass twiddle w_B^(low*q) (times the final scale) of the 16 phase-B elements, in store order
    if (MODE == N2_STRIDED) {
        // q = rev_R(m) = rev_K2(j2)*16 + rev4(m_hi):  base * step^rev_K2(j2)
        const u32 lowblks = 1u << (lo - 4);
#pragma unroll
        for (int t = 0; t < (1 << D); t++) {
            const u32 ntile = (blk << D) + t;
            const u64 low = ((u64)(ntile & (lowblks - 1)) << 4) + uB;
            u64 base = gl_mul(two_level(p.tw_lo, p.tw_hi, p.tw_h, low * (u64)rev_bits_c(m_hi & 15, 4)), p.post_scale);
            const u64 step = (K2 >= 1) ? two_level(p.tw_lo, p.tw_hi, p.tw_h, low << 4) : 1;
            u64 w = base;
#pragma unroll
            for (int k = 0; k < G2; k++) {
                wt[t * G2 + k] = w;
                if (k + 1 < G2) w = gl_mul(w, step);
            }
        }
    }

#pragma unroll 1
    for (int cc = 0; cc < CB; cc++) {
        const size_t col = col0 + cc;
        if (col >= p.ncols) break;
        const u64* __restrict__ in = p.in + col * p.in_col_stride + coset * p.in_coset_stride;
        u64* __restrict__ out = p.out + col * p.out_col_stride + coset * p.out_coset_stride;
        // An opaque per-iteration zero in the lane offsets: the accesses then stay "uniform base (SGPRs) + 32-bit lane
        // offset"; without it LICM hoists 32 per-lane 64-bit addresses out of the column loop (64 VGPRs).
        u32 zlane = 0;
        asm volatile("" : "+v"(zlane));

        // ------------------------------------------------------------ phase A: load + radix-16 round
        {
            u64 x[16];
            if (MODE == N2_STRIDED) {   // the host guarantees lo - 4 >= D (ntt2_run)
                const char* __restrict__ base = reinterpret_cast<const char*>(in + a0_uni);
                const u32 off = (a0_lane + zlane) * 8u;   // a0_lane < 2^(lo+K2+1) <= 2^(L-3): the byte offset fits 32 bits
#pragma unroll
                for (int j = 0; j < 16; j++) x[j] = gl_canon(*reinterpret_cast<const u64*>(base + j * jstride * 8 + off));
            } else {
#pragma unroll
                for (int j = 0; j < 16; j++) x[j] = gl_canon(in[a0 + j * jstride]);
            }
            if (prescale) {
#pragma unroll
                for (int j = 0; j < 16; j++) x[j] = gl_mul(x[j], up[j]);
            }
            dft_pow2<4, INV>(x);
            if (R > 4 && cc == 0) __syncthreads();  // tw1 visible
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int q1 = rev_bits_c(j, 4);
                u64 v = x[j];
                // round twiddle w_{2^R}^(m_low*q1) -- times the pre-scale's t^m_low in a coset transform's first pass.  The strided
                // pass reads it from a [q1][m_low] table: one LDS read at a compile-time row, no data-dependent skipping (a wave
                // mixes all m_low, so skipping the unit entries saves nothing and costs registers: 134 -> 108 VGPRs without it)
                if (MODE == N2_STRIDED) {
                    if (R > 4 && q1 != 0) v = gl_mul(v, tw1[(q1 << K2) + m_low]);
                } else if (R > 4 && q1 != 0) {
                    const int idx = m_low * q1;  // < 2^R
                    if (idx) v = gl_mul(v, tw1[idx]);
                }
                const int m = (j << K2) + m_low;
                if (MODE == N2_STRIDED) lds[pad1((((tA << R) + m) << 4) + uA)] = v;
                else lds[((tA << 4) + uA) * ROW + m] = v;
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ phase B: second round + store
        {
            u64 y[16];
#pragma unroll
            for (int t = 0; t < (1 << D); t++) {
#pragma unroll
                for (int j2 = 0; j2 < G2; j2++) {
                    const int m = (m_hi << K2) + j2;
                    y[t * G2 + j2] = (MODE == N2_STRIDED) ? lds[pad1((((t << R) + m) << 4) + uB)] : lds[((t << 4) + uB) * ROW + m];
                }
            }
#pragma unroll
            for (int t = 0; t < (1 << D); t++)
                if (K2 >= 1) dft_pow2<(K2 >= 1 ? K2 : 1), INV>(y + t * G2);
            if (MODE == N2_STRIDED) {
                const u32 lowblks = 1u << (lo - 4);
#pragma unroll
                for (int t = 0; t < (1 << D); t++) {
                    const u32 ntile = (blk << D) + t;   // t is an unrolled constant: uniform
                    const u32 lb = ntile & (lowblks - 1), hi = ntile >> (lo - 4);
                    char* __restrict__ obase = reinterpret_cast<char*>(out + ((size_t)hi << (lo + R)) + ((size_t)lb << 4));   // uniform
                    const u32 ooff = (((u32)m_hi << (K2 + lo)) + (u32)uB + zlane) * 8u;
#pragma unroll
                    for (int k = 0; k < G2; k++) {
                        const int j2 = rev_bits_c(k, K2);
                        *reinterpret_cast<u64*>(obase + ((size_t)j2 << (lo + 3)) + ooff) = gl_mul(y[t * G2 + j2], wt[t * G2 + k]);
                    }
                }
            } else if (MODE == N2_NATURAL_LAST) {
#pragma unroll
                for (int t = 0; t < (1 << D); t++) {
                    const u32 A = (blk << D) + t;
#pragma unroll
                    for (int j2 = 0; j2 < G2; j2++) {
                        const u32 q = ((u32)rev_bits_c(j2, K2) << 4) + (u32)rev_bits_c(m_hi & 15, 4);
                        out[((size_t)A << 4) + uB + ((size_t)q << (L - R))] = y[t * G2 + j2];
                    }
                }
            } else {
                // bit-reversed (in-place) order: exchange through LDS once more so that the stores are contiguous
                __syncthreads();
#pragma unroll
                for (int t = 0; t < (1 << D); t++)
#pragma unroll
                    for (int j2 = 0; j2 < G2; j2++) lds[((t << 4) + uB) * ROW + (m_hi << K2) + j2] = y[t * G2 + j2];
                __syncthreads();
                const size_t b0 = (size_t)blk << 12;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int c = (k << 8) + tid;  // element within the workgroup's 4096
                    out[b0 + c] = lds[(c >> R) * ROW + (c & ((1 << R) - 1))];
                }
            }
        }
        if (cc + 1 < CB) __syncthreads();  // the exchange buffer is reused by the next column
    }
}

template <int R, int MODE, bool INV, int CB, bool PRE>
__global__ __launch_bounds__(256) void ntt2_pass_kernel(Ntt2Params p) { ntt2_pass_body<R, MODE, INV, CB, PRE>(p); }
// ------------------------------------------------------------------------------------------------ host side
template <int R, int MODE, bool INV, int CB>
static void ntt2_launch_cb(const Ntt2Params& p0, size_t cols, size_t cosets, hipStream_t stream) {
    Ntt2Params p = p0;
    p.ncols = cols;
    const size_t lds_bytes = (size_t)(ntt2_lds_elems<R, MODE>() + (R > 4 ? (1 << R) : 0)) * 8;
    dim3 grid((unsigned)(((size_t)1 << p.log_n) >> 12), (unsigned)((cols + CB - 1) / CB), (unsigned)cosets);
    if (MODE == N2_STRIDED && p.sc_lo) hipLaunchKernelGGL((ntt2_pass_kernel<R, MODE, INV, CB, MODE == N2_STRIDED>), grid, dim3(256), lds_bytes, stream, p);
    else hipLaunchKernelGGL((ntt2_pass_kernel<R, MODE, INV, CB, false>), grid, dim3(256), lds_bytes, stream, p);
}
template <int R, int MODE, bool INV>
static void ntt2_launch(const Ntt2Params& p, size_t cols, size_t cosets, hipStream_t stream) {
    // a closing pass has no pass twiddles to share among columns: one column per workgroup
    ntt2_launch_cb<R, MODE, INV, (MODE == N2_STRIDED ? NTT2_STRIDED_COLS : 1)>(p, cols, cosets, stream);
}
template <int MODE, bool INV>
static void ntt2_dispatch_r(int R, const Ntt2Params& p, size_t cols, size_t cosets, hipStream_t s) {
    switch (R) {
        case 4: ntt2_launch<4, MODE, INV>(p, cols, cosets, s); break;
        case 5: ntt2_launch<5, MODE, INV>(p, cols, cosets, s); break;
        case 6: ntt2_launch<6, MODE, INV>(p, cols, cosets, s); break;
        case 7: ntt2_launch<7, MODE, INV>(p, cols, cosets, s); break;
        default: ntt2_launch<8, MODE, INV>(p, cols, cosets, s); break;
    }
}
static void ntt2_dispatch(int R, int mode, bool inv, const Ntt2Params& p, size_t cols, size_t cosets, hipStream_t s) {
    if (mode == N2_STRIDED) { if (inv) ntt2_dispatch_r<N2_STRIDED, true>(R, p, cols, cosets, s); else ntt2_dispatch_r<N2_STRIDED, false>(R, p, cols, cosets, s); }
    else if (mode == N2_BITREV_LAST) { if (inv) ntt2_dispatch_r<N2_BITREV_LAST, true>(R, p, cols, cosets, s); else ntt2_dispatch_r<N2_BITREV_LAST, false>(R, p, cols, cosets, s); }
    else { if (inv) ntt2_dispatch_r<N2_NATURAL_LAST, true>(R, p, cols, cosets, s); else ntt2_dispatch_r<N2_NATURAL_LAST, false>(R, p, cols, cosets, s); }
}

// ---- T-form passes (ntt2t.cuh): same plan, same tiles, the load multiplier instead of the store-side twiddle
// OLA_NTT2_TFORM=0 runs the canonical-arithmetic passes above instead (the A/B switch of the round-4 experiment).
static bool ntt2_tform() {
    static const bool v = [] { const char* e = getenv("OLA_NTT2_TFORM"); return !(e && atoi(e) == 0); }();
    return v;
}
template <int R, int MODE, bool INV, int LM>
static void ntt2t_launch(const Ntt2Params& p0, size_t cols, size_t cosets, hipStream_t stream) {
    constexpr int CB = NTT2_STRIDED_COLS;   // every pass has sixteen load multipliers per thread to amortise
    Ntt2Params p = p0;
    p.ncols = cols;
    const size_t lds_bytes = (size_t)ntt2_lds_elems<R, MODE>() * 8 + (R > 4 ? ((size_t)sizeof(TfTw) << R) : 0) + (LM == 1 ? ((size_t)8 << R) : 0);
    dim3 grid((unsigned)(((size_t)1 << p.log_n) >> 12), (unsigned)((cols + CB - 1) / CB), (unsigned)cosets);
    hipLaunchKernelGGL((ntt2t_pass_kernel<R, MODE, INV, CB, LM>), grid, dim3(256), lds_bytes, stream, p);
}
template <int MODE, bool INV, int LM>
static void ntt2t_dispatch_r(int R, const Ntt2Params& p, size_t cols, size_t cosets, hipStream_t s) {
    switch (R) {
        case 4: ntt2t_launch<4, MODE, INV, LM>(p, cols, cosets, s); break;
        case 5: ntt2t_launch<5, MODE, INV, LM>(p, cols, cosets, s); break;
        case 6: ntt2t_launch<6, MODE, INV, LM>(p, cols, cosets, s); break;
        case 7: ntt2t_launch<7, MODE, INV, LM>(p, cols, cosets, s); break;
        default: ntt2t_launch<8, MODE, INV, LM>(p, cols, cosets, s); break;
    }
}
template <bool INV>
static void ntt2t_dispatch_inv(int R, int mode, int lm, const Ntt2Params& p, size_t cols, size_t cosets, hipStream_t s) {
    if (mode == N2_STRIDED) {
        if (lm == 0) ntt2t_dispatch_r<N2_STRIDED, INV, 0>(R, p, cols, cosets, s);
        else ntt2t_dispatch_r<N2_STRIDED, INV, 1>(R, p, cols, cosets, s);
    } else if (mode == N2_BITREV_LAST) ntt2t_dispatch_r<N2_BITREV_LAST, INV, 2>(R, p, cols, cosets, s);
    else ntt2t_dispatch_r<N2_NATURAL_LAST, INV, 2>(R, p, cols, cosets, s);
}
static void ntt2t_dispatch(int R, int mode, bool inv, int lm, const Ntt2Params& p, size_t cols, size_t cosets, hipStream_t s) {
    if (inv) ntt2t_dispatch_inv<true>(R, mode, lm, p, cols, cosets, s);
    else ntt2t_dispatch_inv<false>(R, mode, lm, p, cols, cosets, s);
}
// the same launch between two events (NttTables::pass_timing)
static void ntt2t_dispatch_timed(NttTables& t, int R, int mode, bool inv, int lm, const Ntt2Params& p, size_t cols, size_t cosets, hipStream_t s) {
    if (!t.pass_timing || t.pass_recs.size() >= 4096) { ntt2t_dispatch(R, mode, inv, lm, p, cols, cosets, s); return; }
    NttTables::PassRec r{R, mode, lm, inv, ((size_t)1 << p.log_n) * cols * cosets, nullptr, nullptr};
    HIP_CHECK(hipEventCreate(&r.a));
    HIP_CHECK(hipEventCreate(&r.b));
    HIP_CHECK(hipEventRecord(r.a, s));
    ntt2t_dispatch(R, mode, inv, lm, p, cols, cosets, s);
    HIP_CHECK(hipEventRecord(r.b, s));
    t.pass_recs.push_back(r);
}

// round twiddles of a strided pass, [q1][m]: w_{2^R}^(m*q1), q1 < 16, m < 2^(R-4)
static const u64* get_round_table(NttTables& t, int R, int inverse) {
    auto key = std::make_pair(200 + R, inverse);
    auto it = t.small.find(key);
    if (it != t.small.end()) return it->second;
    const std::vector<u64> full = powers(root_for(R, inverse), (size_t)1 << R);
    std::vector<u64> v;
    for (int q1 = 0; q1 < 16; q1++)
        for (int m = 0; m < (1 << (R - 4)); m++) v.push_back(full[(size_t)m * q1]);
    return t.small[key] = upload(t.ctx, v);
}

// full-circle table w_{2^R}^e, e < 2^R
static const u64* get_full_small(NttTables& t, int R, int inverse) {
    auto key = std::make_pair(100 + R, inverse);
    auto it = t.small.find(key);
    if (it != t.small.end()) return it->second;
    std::vector<u64> v = powers(root_for(R, inverse), (size_t)1 << R);
    return t.small[key] = upload(t.ctx, v);
}

// per-coset ratio s_c^(2^e) for the pre-scale recurrence
static const u64* get_coset_steps(NttTables& t, int log_n, int rate_bits, int e, u64 single_shift) {
    std::vector<u64> v;
    if (rate_bits < 0) {
        v.push_back(gl_pow(single_shift, (u64)1 << e));
    } else {
        const u64 g = gl_root_of_unity(log_n + rate_bits);
        for (int c = 0; c < (1 << rate_bits); c++) {
            const u64 s = gl_mul(gl_pow(g, bitrev32((u32)c, rate_bits)), GL_GENERATOR);
            v.push_back(gl_pow(s, (u64)1 << e));
        }
    }
    return upload(t.ctx, v);  // small; lives in the persistent pool
}

// T-form passes of a coset transform: per coset c, (s_c^(2^lo))^m for m < 2^R -- the part of the pre-scale s^a that belongs to
// the pass over index bits [lo, lo + R)
static const u64* get_coset_pows(NttTables& t, int log_n, int rate_bits, int lo, int R, u64 single_shift) {
    std::vector<u64> shifts;
    if (rate_bits < 0) shifts.push_back(single_shift);
    else {
        const u64 g = gl_root_of_unity(log_n + rate_bits);
        for (int c = 0; c < (1 << rate_bits); c++) shifts.push_back(gl_mul(gl_pow(g, bitrev32((u32)c, rate_bits)), GL_GENERATOR));
    }
    std::vector<u64> v;
    for (u64 sc : shifts) {
        const std::vector<u64> pw = powers(gl_pow(sc, (u64)1 << lo), (size_t)1 << R);
        v.insert(v.end(), pw.begin(), pw.end());
    }
    return upload(t.ctx, v);  // small; lives in the persistent pool
}

// Same contract as ntt_run (ntt.hip) for L >= 14.  prescale: rate_bits >= 0 selects the LDE coset family
// (7*g^bitrev(c)) of which cosets [coset_first, coset_first + cosets) are produced, rate_bits = -1 with `shift` a single
// coset, -2 none.
static void ntt2_run_group(NttTables& t, const u64* in, size_t in_col_stride, u64* out, size_t out_col_stride, u64* scratch,
                           size_t scratch_col_stride, int L, size_t cols, bool inverse, bool natural_out, int sc_rate_bits, u64 sc_shift,
                           size_t cosets, size_t out_coset_stride, size_t coset_first) {
    hipStream_t stream = t.ctx->stream;
    const int P = (L + 7) / 8;
    std::vector<int> Rs = split_even(L, P);          // largest first ...
    std::reverse(Rs.begin(), Rs.end());              // ... make the closing pass the widest (best coalescing)
    const u64 final_scale = inverse ? gl_inv(((u64)1 << L) % GL_P) : 1;
    u64* work = natural_out ? scratch : out;
    const size_t work_col_stride = natural_out ? scratch_col_stride : out_col_stride;
    const size_t work_coset_stride = natural_out ? 0 : out_coset_stride;

    const u64* cur_in = in;
    size_t cur_in_stride = in_col_stride, cur_in_coset = 0;
    int lo = L;
    if (ntt2_tform() && L <= 28) {   // (their lane offsets are 32-bit byte offsets)
        // T-form passes (ntt2t.cuh): pass i multiplies what it LOADS by w_{2^(L-lo)}^(a_i * K) -- the inter-pass twiddles in the
        // factorisation that depends on the pass's own index a_i only -- and, in a coset transform, by (s^(2^lo))^a_i; the closing
        // pass also takes the final scale and writes canonical words
        for (int i = 0; i < P; i++) {
            const int R = Rs[i];
            lo -= R;
            const bool last = (i == P - 1);
            Ntt2Params p = {};
            p.log_n = L; p.lo = lo;
            p.tw_r = get_round_table(t, R, inverse);
            p.post_scale = last ? final_scale : 1;
            p.in = cur_in; p.in_col_stride = cur_in_stride; p.in_coset_stride = cur_in_coset;
            if (i > 0) {
                TwoLevel tw = get_two(t, L - lo, inverse);
                p.tw_lo = tw.lo; p.tw_hi = tw.hi; p.tw_h = tw.h;
            }
            if (sc_rate_bits != -2) {
                const auto skey = std::make_tuple(L, sc_rate_bits, 1000 + lo * 16 + R, sc_rate_bits >= 0 ? (u64)0 : sc_shift);
                auto st = t.coset_steps.find(skey);
                if (st == t.coset_steps.end()) st = t.coset_steps.emplace(skey, get_coset_pows(t, L, sc_rate_bits, lo, R, sc_shift)).first;
                p.sc_pow = st->second + (coset_first << R);
            }
            const int lm = last ? 2 : ((i > 0 || p.sc_pow) ? 1 : 0);
            if (!last) {
                p.out = work; p.out_col_stride = work_col_stride; p.out_coset_stride = work_coset_stride;
                if (lo - 4 < 8 - R) throw OlaError(-7, "ntt2: pass split leaves a strided pass with lo + R < 12");
                ntt2t_dispatch_timed(t, R, N2_STRIDED, inverse, lm, p, cols, cosets, stream);
                cur_in = work; cur_in_stride = work_col_stride; cur_in_coset = work_coset_stride;
            } else {
                p.out = out; p.out_col_stride = out_col_stride; p.out_coset_stride = out_coset_stride;
                ntt2t_dispatch_timed(t, R, natural_out ? N2_NATURAL_LAST : N2_BITREV_LAST, inverse, 2, p, cols, cosets, stream);
            }
        }
        return;
    }
    for (int i = 0; i < P; i++) {
        const int R = Rs[i];
        lo -= R;
        const bool last = (i == P - 1);
        Ntt2Params p = {};
        p.log_n = L; p.lo = lo;
        p.tw_r = last ? get_full_small(t, R, inverse) : get_round_table(t, R, inverse);
        p.post_scale = 1;
        p.in = cur_in; p.in_col_stride = cur_in_stride; p.in_coset_stride = cur_in_coset;
        if (!last) {
            TwoLevel tw = get_two(t, lo + R, inverse);
            p.tw_lo = tw.lo; p.tw_hi = tw.hi; p.tw_h = tw.h;
            if (i == P - 2) p.post_scale = final_scale;
            p.out = work; p.out_col_stride = work_col_stride; p.out_coset_stride = work_coset_stride;
            if (i == 0 && sc_rate_bits != -2) {
                TwoLevel sc;
                size_t stride = 0;
                if (sc_rate_bits >= 0) sc = get_coset(t, L, sc_rate_bits, &stride);
                else sc = get_shift(t, L, sc_shift);
                // a coset sub-range (one GPU's share of the LDE) starts `coset_first` entries into the per-coset tables
                p.sc_lo = sc.lo + coset_first * stride; p.sc_hi = sc.hi + coset_first * stride; p.sc_h = sc.h; p.sc_coset_stride = stride;
                {
                    const auto skey = std::make_tuple(L, sc_rate_bits, lo + R - 4, sc_rate_bits >= 0 ? (u64)0 : sc_shift);
                    auto st = t.coset_steps.find(skey);
                    if (st == t.coset_steps.end()) st = t.coset_steps.emplace(skey, get_coset_steps(t, L, sc_rate_bits, lo + R - 4, sc_shift)).first;
                    p.sc_step = st->second + coset_first;
                }
            }
            // the strided kernel addresses a workgroup's 2^(8-R) tiles as one uniform base + lane offsets
            if (lo - 4 < 8 - R) throw OlaError(-7, "ntt2: pass split leaves a strided pass with lo + R < 12");
            ntt2_dispatch(R, N2_STRIDED, inverse, p, cols, cosets, stream);
            cur_in = work; cur_in_stride = work_col_stride; cur_in_coset = work_coset_stride;
        } else {
            p.out = out; p.out_col_stride = out_col_stride; p.out_coset_stride = out_coset_stride;
            ntt2_dispatch(R, natural_out ? N2_NATURAL_LAST : N2_BITREV_LAST, inverse, p, cols, cosets, stream);
        }
    }
}

// Column-group blocking for the Infinity Cache (256 MiB, memory side): a transform of P passes launched over the whole batch
// streams the batch through HBM P times (3.15 GB per pass at 94 x 2^22 -- nothing survives on-die between launches).  Run
// group by group instead -- all P passes over `gc` columns (and `gk` cosets of an LDE) back to back -- and the intermediate of
// pass i is still in the cache when pass i + 1 reads it, as long as a group's working set (gc * gk * 8n bytes, twice that for
// an out-of-place natural-order transform) stays well under the cache size.  OLA_NTT2_GROUP_MB sets the working-set target
// (0 = one group: the whole batch per launch); columns per group are a multiple of the strided kernel's columns per workgroup.
static size_t ntt2_group_bytes() {
    const char* e = getenv("OLA_NTT2_GROUP_MB");
    return e ? (size_t)atol(e) << 20 : (size_t)NTT2_DEFAULT_GROUP_MB << 20;
}

void ntt2_run(NttTables& t, const u64* in, size_t in_col_stride, u64* out, size_t out_col_stride, u64* scratch,
              size_t scratch_col_stride, int L, size_t cols, bool inverse, bool natural_out, int sc_rate_bits, u64 sc_shift,
              size_t cosets, size_t out_coset_stride, size_t coset_first) {
    if (cols == 0) return;
    const size_t budget = ntt2_group_bytes(), col_bytes = ((size_t)8 << L) * (natural_out ? 2 : 1);
    size_t gc = cols, gk = cosets;
    if (budget && L >= 18 && col_bytes * cols * cosets > budget) {
        // whole cosets first (a coset of a column group is an independent transform), then columns in units of the
        // workgroup's column block
        const size_t unit = NTT2_STRIDED_COLS;
        gk = 1;
        gc = std::max<size_t>(unit, budget / col_bytes / unit * unit);
        if (gc >= cols) { gc = cols; gk = std::max<size_t>(1, std::min(cosets, budget / (col_bytes * cols))); }
    }
    for (size_t c0 = 0; c0 < cols; c0 += gc) {
        const size_t nc = std::min(gc, cols - c0);
        for (size_t k0 = 0; k0 < cosets; k0 += gk) {
            const size_t nk = std::min(gk, cosets - k0);
            ntt2_run_group(t, in + c0 * in_col_stride, in_col_stride, out + c0 * out_col_stride + k0 * out_coset_stride, out_col_stride,
                           scratch ? scratch + c0 * scratch_col_stride : nullptr, scratch_col_stride, L, nc, inverse, natural_out,
                           sc_rate_bits, sc_shift, nk, out_coset_stride, coset_first + k0);
        }
    }
}

}  // namespace ola

