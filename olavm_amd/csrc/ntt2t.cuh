// "T-form" passes of the large transforms (ntt2.hip), gfx950: the same pass plan, tiles, LDS layout and results as
// ntt2_pass_kernel, with the arithmetic of a pass done in a redundant representation in which the butterflies need no
// carries, no reductions and no 64-bit instructions.
//
// p = 2^64 - 2^32 + 1 divides 2^96 + 1, so with T = 2^24 a field element can be written
//     x = v0 + v1*T + v2*T^2 + v3*T^3   (mod p),   T^4 = -1,
// with SIGNED 32-bit limbs (24 bits of payload, 7 bits of headroom, many limb vectors per element).  On this ALU
// (profiles/r02_valu_rates_waves.txt: plain 32-bit add / sub / and / shift-right issue in 2.6 cycles per wave64 instruction,
// 64-bit adds, carries, selects, compares and multiply-adds in 4.3 - 4.8):
//   * a butterfly is eight plain 32-bit adds (canonical u64: 6 + 5 instructions of the slow class, with carries and selects);
//   * the twiddles inside a radix-16 / radix-8 block are powers of two (w_16 = 2^156): a multiple of 24 bits is a renaming of
//     limbs with the sign taken at the subtraction (free), the four odd multiples of 12 in a radix-16 block cost one
//     shift-and-carry step (12 instructions, which also re-normalises);
//   * a general multiplication by a table twiddle is twelve v_mad_i64_i32 and a split of the four sums into 24-bit pieces;
//   * elements enter a pass as u64 (any representative) and are multiplied as they are loaded -- a 64 x 64 -> 128-bit product
//     that is cut straight into limbs, no modular reduction -- and leave it through one 128 -> 64-bit fold; passes other than
//     the closing one store any representative;
//   * the inter-pass twiddles are taken on the INPUT side of a pass, in the factorisation where they depend on the pass's own
//     index only:  with a = a_1 M_1 + a_2 M_2 + ... (a_i = the index bits pass i transforms, M_i = 2^lo_i) and k_1, k_2, ... the
//     frequencies the earlier passes left at the element's position, the element is multiplied by
//         w_{N_1...N_i}^(a_i * K),   K = k_1 + N_1 k_2 + ... = bitrev(address bits above the pass),
//     times (s^M_i)^a_i in a coset transform (the pre-scale s^a splits the same way: the part that does not depend on a pass's
//     index commutes with that pass).  For a strided pass K is the same for the whole workgroup, so the multipliers are a
//     2^R-entry table in LDS instead of sixteen 64-bit registers per thread (what the store-side twiddles of ntt2_pass_body
//     cost); only the closing pass, whose tile holds sixteen rows, keeps them in registers.
// Reference semantics unchanged (plonky2/field/src/cfft/mod.rs:22-231, serial.rs:9-78, goldilocks_field.rs:191-355): values
// enter and leave a TRANSFORM as in ntt2.hip, the closing pass writes canonical words.
#pragma once
#include <hip/hip_runtime.h>

#include "gl.cuh"
#include "tform.cuh"   // the limb arithmetic (host + device: tests/host_tform_check.cpp runs it on the CPU)

namespace ola {

enum { N2_STRIDED = 0, N2_BITREV_LAST = 1, N2_NATURAL_LAST = 2 };

struct Ntt2Params {
    const u64* in;
    u64* out;
    size_t in_col_stride, out_col_stride, in_coset_stride, out_coset_stride;
    int log_n;  // L
    size_t ncols;  // columns of the batch (a workgroup handles CB of them)
    int lo;     // pass handles index bits [lo, lo+R)
    const u64* tw_r;   // w_{2^R}^e, e < 2^R
    const u64* tw_lo;  // two-level powers of w_{2^(lo+R)}  (T-form passes: of w_{2^(L-lo)}, NULL in the first pass)
    const u64* tw_hi;
    int tw_h;
    u64 post_scale;    // folded into the pass twiddle (STRIDED only; T-form passes: into the load multiplier)
    const u64* sc_lo;  // optional pre-scale s^k (coset transforms), two-level per coset
    const u64* sc_hi;
    int sc_h;
    size_t sc_coset_stride;
    const u64* sc_step;  // canonical passes: per coset s^(2^(lo+R-4)) = ratio between consecutive register elements
    const u64* sc_pow;   // T-form passes of a coset transform: per coset, 2^R entries (s^(2^lo))^m
};

__device__ __forceinline__ constexpr int rev_bits_c(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

__device__ __forceinline__ u64 two_level(const u64* __restrict__ lo, const u64* __restrict__ hi, int h, u64 e) {
    return gl_mul(lo[e & (((u64)1 << h) - 1)], hi[e >> h]);
}

template <int R, int MODE>
constexpr int ntt2_lds_elems() {
    constexpr int K2 = R - 4;
    if (MODE == N2_STRIDED) return 4096 + (K2 >= 1 ? 16 * (4096 >> (K2 + 4)) : 0);
    return (1 << (8 - R)) * 16 * ((1 << R) + 1);
}

// ------------------------------------------------------------------------------------------------ the pass
// LM: how an element is multiplied as it is loaded (see the header): 0 not at all (first pass of a plain transform), 1 by a
// 2^R-entry table in LDS (strided passes), 2 by sixteen per-thread registers (closing passes; times the transform's final scale).
// Either way the multipliers are computed once per workgroup and reused for its CB columns.
template <int R, int MODE, bool INV, int CB, int LM>
__device__ __forceinline__ void ntt2t_pass_body(const Ntt2Params& p) {
    static_assert(R >= 4 && R <= 8, "pass width");
    static_assert((LM == 2) == (MODE != N2_STRIDED), "strided passes keep their multipliers in LDS, closing passes in registers");
    constexpr int K2 = R - 4;     // bits of the second round
    constexpr int D = 8 - R;      // log2(tiles per workgroup)
    constexpr int G2 = 1 << K2;   // values per second-round group
    constexpr int ROW = (1 << R) + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* lds = reinterpret_cast<u64*>(smem_raw);
    TfTw* tw1 = reinterpret_cast<TfTw*>(lds + ntt2_lds_elems<R, MODE>());
    u64* lmtab = reinterpret_cast<u64*>(tw1 + (R > 4 ? (1 << R) : 0));   // LM == 1: 2^R load multipliers

    const int tid = threadIdx.x;
    const u32 blk = blockIdx.x;
    const size_t col0 = (size_t)blockIdx.y * CB, coset = blockIdx.z;
    const int L = p.log_n, lo = p.lo;

    if (R > 4) {
        if (tid < (1 << R)) tw1[tid] = tf_split_u64(p.tw_r[tid]);   // [q1][m_low]: w_{2^R}^(m_low*q1)
    }

    auto pad1 = [](int e) -> int { return K2 >= 1 ? e + ((e >> (K2 + 4)) << 4) : e; };

    // ---- phase A thread -> element map (as ntt2_pass_body)
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
    size_t a0, jstride;
    size_t a0_uni = 0;
    u32 a0_lane = 0;
    {
        const u32 ntile = (blk << D) + tA;
        if (MODE == N2_STRIDED) {
            const u32 lowblks = 1u << (lo - 4);
            const u32 lb = ntile & (lowblks - 1), hi = ntile >> (lo - 4);
            a0 = ((size_t)hi << (lo + R)) + ((size_t)lb << 4) + ((size_t)m_low << lo) + uA;
            jstride = (size_t)1 << (lo + K2);
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
    const int uB = tid & 15, m_hi = tid >> 4;

    // ---- load multipliers (column-independent)
    const int HB = L - lo - R;   // address bits above the pass
    u64 wl[LM == 2 ? 16 : 1];
    if (LM == 1) {
        if (tid < (1 << R)) {
            const u32 hi = (blk << D) >> (lo - 4);                      // the same for the workgroup's 2^D tiles (lo - 4 >= D)
            const u64 K = HB ? bitrev32(hi, HB) : 0;
            u64 v = 1;
            if (p.tw_lo) v = two_level(p.tw_lo, p.tw_hi, p.tw_h, K * (u64)tid);
            if (p.sc_pow) v = p.tw_lo ? gl_mul(v, p.sc_pow[(coset << R) + tid]) : p.sc_pow[(coset << R) + tid];
            lmtab[tid] = v;
        }
    } else if (LM == 2) {
        // closing pass (lo = 0): the thread's sixteen elements are m = m_low + j * 2^K2 of ONE row; c = w_n^K * s, wl[j] = c^m
        const u64 K = bitrev32((u32)(a0 >> R), HB);
        u64 base = gl_mul(two_level(p.tw_lo, p.tw_hi, p.tw_h, K * (u64)m_low), p.post_scale);
        u64 step = two_level(p.tw_lo, p.tw_hi, p.tw_h, K << K2);
        if (p.sc_pow) {
            base = gl_mul(base, p.sc_pow[(coset << R) + m_low]);
            step = gl_mul(step, p.sc_pow[(coset << R) + (1 << K2)]);
        }
        u64 w = base;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            wl[j] = w;
            if (j < 15) w = gl_mul(w, step);
        }
    }

    if (LM == 1 || R > 4) __syncthreads();   // tw1 / lmtab visible

#pragma unroll 1
    for (int cc = 0; cc < CB; cc++) {
        const size_t col = col0 + cc;
        if (col >= p.ncols) break;
        const u64* __restrict__ in = p.in + col * p.in_col_stride + coset * p.in_coset_stride;
        u64* __restrict__ out = p.out + col * p.out_col_stride + coset * p.out_coset_stride;
        u32 zlane = 0;
        asm volatile("" : "+v"(zlane));   // keeps the accesses "uniform base + 32-bit lane offset" (see ntt2_pass_body)

        T4 y[16];
        {
            // ------------------------------------------------------------ phase A: load (x multiplier) + radix-16 round
            T4 x[16];
            {
                u64 raw[16];
                if (MODE == N2_STRIDED) {
                    const char* __restrict__ base = reinterpret_cast<const char*>(in + a0_uni);
                    const u32 off = (a0_lane + zlane) * 8u;
#pragma unroll
                    for (int j = 0; j < 16; j++) raw[j] = *reinterpret_cast<const u64*>(base + j * jstride * 8 + off);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) raw[j] = in[a0 + j * jstride];
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (LM) {
                        u64 plo, phi;
                        mul_wide(raw[j], LM == 1 ? lmtab[(j << K2) + m_low] : wl[j], plo, phi);
                        x[j] = tf_from_u128(plo, phi);
                    } else {
                        x[j] = tf_from_u64(raw[j]);
                    }
                }
            }
            tf_dft<4, INV>(x);
            if (R > 4) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int q1 = rev_bits_c(j, 4);
                    if (q1 != 0) x[j] = tf_mul(x[j], tw1[(q1 << K2) + m_low]);
                    else x[j] = tf_norm(x[j]);
                }
            }
            // ------------------------------------------------------------ exchange, two limbs at a time (the buffer of ntt2_pass_body)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (h == 1) __syncthreads();
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int m = (j << K2) + m_low;
                    const u64 v = (u64)(u32)x[j].v[2 * h] | ((u64)(u32)x[j].v[2 * h + 1] << 32);
                    if (MODE == N2_STRIDED) lds[pad1((((tA << R) + m) << 4) + uA)] = v;
                    else lds[((tA << 4) + uA) * ROW + m] = v;
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < (1 << D); t++) {
#pragma unroll
                    for (int j2 = 0; j2 < G2; j2++) {
                        const int m = (m_hi << K2) + j2;
                        const u64 v = (MODE == N2_STRIDED) ? lds[pad1((((t << R) + m) << 4) + uB)] : lds[((t << 4) + uB) * ROW + m];
                        y[t * G2 + j2].v[2 * h] = (i32)(u32)v;
                        y[t * G2 + j2].v[2 * h + 1] = (i32)(u32)(v >> 32);
                    }
                }
            }
        }

        // ------------------------------------------------------------ phase B: second round + store
#pragma unroll
        for (int t = 0; t < (1 << D); t++)
            if (K2 >= 1) tf_dft<(K2 >= 1 ? K2 : 1), INV>(y + t * G2);
        if (MODE == N2_STRIDED) {
            const u32 lowblks = 1u << (lo - 4);
#pragma unroll
            for (int t = 0; t < (1 << D); t++) {
                const u32 ntile = (blk << D) + t;   // t is an unrolled constant: uniform
                const u32 lb = ntile & (lowblks - 1), hi = ntile >> (lo - 4);
                char* __restrict__ obase = reinterpret_cast<char*>(out + ((size_t)hi << (lo + R)) + ((size_t)lb << 4));   // uniform
                const u32 ooff = (((u32)m_hi << (K2 + lo)) + (u32)uB + zlane) * 8u;
#pragma unroll
                for (int j2 = 0; j2 < G2; j2++)
                    *reinterpret_cast<u64*>(obase + ((size_t)j2 << (lo + 3)) + ooff) = tf_to_u64<false>(y[t * G2 + j2]);
            }
        } else if (MODE == N2_NATURAL_LAST) {
#pragma unroll
            for (int t = 0; t < (1 << D); t++) {
                // frequency q = rev(j2) * 16 + rev4(m_hi) of rows A*16 + uB goes to q * 2^(L-R) + A*16 + uB: the part that depends on
                // (t, j2) is uniform, the lane's part is a 32-bit byte offset (the host keeps L below 29 for these passes)
                const u32 A = (blk << D) + t;
                const u32 ooff = ((((u32)rev_bits_c(m_hi & 15, 4)) << (L - R)) + (u32)uB + zlane) * 8u;
#pragma unroll
                for (int j2 = 0; j2 < G2; j2++) {
                    char* __restrict__ obase = reinterpret_cast<char*>(out + ((size_t)A << 4) + ((size_t)rev_bits_c(j2, K2) << (L - R + 4)));
                    *reinterpret_cast<u64*>(obase + ooff) = tf_to_u64<true>(y[t * G2 + j2]);
                }
            }
        } else {
            // bit-reversed (in-place) order: exchange through LDS once more so that the stores are contiguous
            __syncthreads();
#pragma unroll
            for (int t = 0; t < (1 << D); t++)
#pragma unroll
                for (int j2 = 0; j2 < G2; j2++) lds[((t << 4) + uB) * ROW + (m_hi << K2) + j2] = tf_to_u64<true>(y[t * G2 + j2]);
            __syncthreads();
            const size_t b0 = (size_t)blk << 12;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int c = (k << 8) + tid;  // element within the workgroup's 4096
                out[b0 + c] = lds[(c >> R) * ROW + (c & ((1 << R) - 1))];
            }
        }
        if (cc + 1 < CB) __syncthreads();  // the exchange buffer is reused by the next column
    }
}

// Strided passes are held to 128 VGPRs = four waves per SIMD (they sit at 124 - 130 by themselves; at most two spilled words):
// with their multipliers in LDS the pass is a balance of load latency and issue time, and the fourth wave is what hides the loads.
// Closing passes carry sixteen 64-bit multipliers per thread (142 - 156 VGPRs, three waves); held to 128 they spill 16 - 32 words
// and the x8 LDE of 94 x 2^22 goes from 32.7 to 35.2 ms (same-box alternation, round 4): left alone.
template <int R, int MODE, bool INV, int CB, int LM>
__global__ __launch_bounds__(256, (MODE == N2_STRIDED ? 4 : 1)) void ntt2t_pass_kernel(Ntt2Params p) { ntt2t_pass_body<R, MODE, INV, CB, LM>(p); }

}  // namespace ola
