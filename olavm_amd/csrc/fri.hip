// Host-side Fiat-Shamir challenger + device-side opening / FRI pipeline.
//
// Replaces, for commitments resident in HBM (reference paths relative to /root/reference):
//   plonky2/plonky2/src/iop/challenger.rs:36-162          Challenger (host; a few hundred permutations per proof)
//   circuits/src/stark/proof.rs:198-233                    StarkOpeningSet::new      -> eval_points_kernel
//   circuits/src/stark/stark.rs:87-146                     fri_instance (batches zeta, g*zeta, g^-1)
//   plonky2/plonky2/src/fri/oracle.rs:167-241              prove_openings: composition, divide_by_linear, *X, LDE
//   plonky2/field/src/polynomial/division.rs:74-87         divide_by_linear          -> weighted suffix scan
//   plonky2/plonky2/src/fri/prover.rs:72-121               fri_committed_trees       -> leaf_hash_ext + fold + coset NTT
//   plonky2/plonky2/src/fri/prover.rs:126-148              fri_proof_of_work         -> pow_kernel (minimal nonce)
//   plonky2/plonky2/src/fri/prover.rs:150-204              query rounds              -> gathers
//   circuits/src/stark/serialization.rs:163-176,305-317    wire format
//
// Data layout: extension-field arrays are two planes (a, b) of base elements so that an extension NTT is two base
// NTTs (twiddles are base-field, plonky2/field/src/extension/quadratic.rs:61-65).  (p(X)-p(z))/(X-z) is not computed
// by the reference's sequential Horner scan but as  q[k-1] = z^-k * sum_{j>=k} c_j z^j  (a parallel suffix sum);
// multiplying by X afterwards (oracle.rs:218) cancels the index shift, so final[k] = sum_batches w_b * B_b[k], k >= 1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/ola_gpu.h"
#include "device_ctx.h"
#include "gl.cuh"
#include "poseidon_host.h"
#include "blake3.cuh"

namespace ola {

// ------------------------------------------------------------------------------------------------ challenger
static void challenger_duplex(OlaChallenger& ch) {
    for (uint32_t i = 0; i < ch.input_len; i++) ch.sponge_state[i] = ch.input_buffer[i];
    ch.input_len = 0;
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = ch.sponge_state[i];
    if (ch.hasher == OLA_HASH_BLAKE3) b3_permutation_host(s);   // H::Permutation of Challenger<F, H> (challenger.rs:134-153)
    else poseidon_permute_host(s);
    for (int i = 0; i < 12; i++) ch.sponge_state[i] = s[i];
    for (int i = 0; i < 8; i++) ch.output_buffer[i] = s[i];
    ch.output_len = 8;
}
void challenger_observe(OlaChallenger& ch, const u64* e, size_t n) {
    for (size_t i = 0; i < n; i++) {
        ch.output_len = 0;
        ch.input_buffer[ch.input_len++] = gl_canon(e[i]);
        if (ch.input_len == 8) challenger_duplex(ch);
    }
}
// observe_cap (challenger.rs:75-84): GenericHashOut::to_vec of every digest
void challenger_observe_cap(OlaChallenger& ch, const u64* digests, size_t n) {
    if (ch.hasher != OLA_HASH_BLAKE3) { challenger_observe(ch, digests, 4 * n); return; }
    for (size_t i = 0; i < n; i++) {
        u64 e[5];
        b3_digest_elements(digests + 4 * i, e);
        challenger_observe(ch, e, 5);
    }
}
void challenger_init(OlaChallenger& ch, uint32_t hasher) {
    memset(&ch, 0, sizeof(ch));
    ch.hasher = hasher;
}
u64 challenger_get(OlaChallenger& ch) {
    if (ch.input_len != 0 || ch.output_len == 0) challenger_duplex(ch);
    return ch.output_buffer[--ch.output_len];
}
void challenger_compact(OlaChallenger& ch) {
    if (ch.input_len != 0) challenger_duplex(ch);
    ch.output_len = 0;
}
static Ext2 challenger_get_ext(OlaChallenger& ch) {
    const u64 a = challenger_get(ch);
    const u64 b = challenger_get(ch);
    return ext_make(a, b);
}
static void challenger_observe_ext(OlaChallenger& ch, Ext2 e) {
    u64 v[2] = {e.a, e.b};
    challenger_observe(ch, v, 2);
}

// ------------------------------------------------------------------------------------------------ kernels
// two-level extension power table: z^k = lo[k & mask] * hi[k >> h]; stored as 4 planes [lo.a | lo.b | hi.a | hi.b]
struct ExtPow {
    const u64* lo_a; const u64* lo_b; const u64* hi_a; const u64* hi_b;
    int h;
    // hi[kh] cut into 22-bit limbs for eval_points_wide_kernel: per kh twelve words [a: w0 w1 w2 | a 2^32: w0 w1 w2 | b: ... | b 2^32: ...]
    // (null when the table was built without them)
    const u32* hi_limbs;
};
__device__ __forceinline__ Ext2 ext_pow_lookup(const ExtPow& t, size_t k) {
    const size_t il = k & (((size_t)1 << t.h) - 1), ih = k >> t.h;
    return ext_mul(ext_make(t.lo_a[il], t.lo_b[il]), ext_make(t.hi_a[ih], t.hi_b[ih]));
}

// Evaluate `ncols` base-field polynomials (column-major coefficients, n each) at up to two extension points.
// Block (x = coefficient chunk, y = group of CG columns); partial sums out[pt][chunk][col] as (a, b).
#define EVAL_CG 8
__global__ __launch_bounds__(256) void eval_points_kernel(const u64* __restrict__ coeffs, size_t n, int ncols, ExtPow p0, ExtPow p1,
                                                          int npoints, size_t chunk_len, u64* __restrict__ partial) {
    __shared__ u64 red[256 * 2];
    const int c0 = blockIdx.y * EVAL_CG;
    const size_t k_begin = (size_t)blockIdx.x * chunk_len;
    const size_t k_end = min(n, k_begin + chunk_len);
    Ext2 acc[2][EVAL_CG];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int c = 0; c < EVAL_CG; c++) acc[p][c] = ext_make(0, 0);
    for (size_t k = k_begin + threadIdx.x; k < k_end; k += 256) {
        const Ext2 w0 = ext_pow_lookup(p0, k);
        Ext2 w1 = ext_make(0, 0);
        if (npoints > 1) w1 = ext_pow_lookup(p1, k);
#pragma unroll
        for (int c = 0; c < EVAL_CG; c++) {
            if (c0 + c < ncols) {
                const u64 f = coeffs[(size_t)(c0 + c) * n + k];
                acc[0][c] = ext_add(acc[0][c], ext_scalar_mul(w0, f));
                if (npoints > 1) acc[1][c] = ext_add(acc[1][c], ext_scalar_mul(w1, f));
            }
        }
    }
    // the accumulators are addressed with compile-time indices only (fully unrolled, the run-time conditions are uniform): a
    // run-time index would put the whole array into scratch memory -- it did until round 3 (272 bytes per lane), and every
    // accumulation of the loop above went through it
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int c = 0; c < EVAL_CG; c++) {
            if (p < npoints && c0 + c < ncols) {
                __syncthreads();
                red[threadIdx.x] = acc[p][c].a;
                red[256 + threadIdx.x] = acc[p][c].b;
                __syncthreads();
                for (int s = 128; s > 0; s >>= 1) {
                    if ((int)threadIdx.x < s) {
                        red[threadIdx.x] = gl_add(red[threadIdx.x], red[threadIdx.x + s]);
                        red[256 + threadIdx.x] = gl_add(red[256 + threadIdx.x], red[256 + threadIdx.x + s]);
                    }
                    __syncthreads();
                }
                if (threadIdx.x == 0) {
                    u64* o = partial + (((size_t)p * gridDim.x + blockIdx.x) * ncols + (c0 + c)) * 2;
                    o[0] = red[0];
                    o[1] = red[256];
                }
            }
        }
    }
}

// The same evaluations for n >= 2^15 (round 6).  eval_points_kernel spends 124 VALU instructions per coefficient (two points):
// four modular multiplications with their reductions, four modular additions, and a share of the two extension multiplications
// that assemble z^k from the table -- 11 GB of coefficients of a 2^22-row proof at 1.0 - 1.2 TB/s, issue-bound
// (profiles/r05_proof_pmc_blake3.txt: 2.72 G wave-instructions, traffic 1.0 x algorithmic).  Here
//   f(z) = sum_kl lo[kl] * ( sum_kh hi[kh] f[kh 2^h + kl] ),      z^k = lo[k mod 2^h] hi[k >> h]:
// a thread owns one kl and C columns and walks kh, so that
//   * the multiplier hi[kh] is the same for the whole workgroup: it arrives through the scalar cache in SGPRs, cut by the host
//     into three 22-bit limbs of w and of w 2^32 mod p (f w = f_lo w + f_hi (w 2^32));
//   * a product f_half x limb is below 2^54, so the three limb positions are plain 64-bit sums of up to 1024 products: six
//     v_mad_u64_u32 per coefficient, point and component, no carry, no reduction, no dependency between them;
//   * the three sums are folded once per thread (one 128-bit reduction), multiplied by lo[kl] and added up over the workgroup.
// 24 instructions per coefficient for two points instead of 124; the kernel then waits for HBM.
// partial[pt][chunk][col] as for eval_points_kernel, chunk = blockIdx.x.
#define EVALW_MAX_KH 512      // 2 products per step and sum: 1024 x 2^54 <= 2^64
__device__ __forceinline__ u64 evalw_fold(const u64 (&c)[3]) {
    // c0 + c1 2^22 + c2 2^44 as a 128-bit number (< 2^109), then one reduction
    const unsigned __int128 v = (unsigned __int128)c[0] + ((unsigned __int128)c[1] << 22) + ((unsigned __int128)c[2] << 44);
    return gl_reduce128((u64)v, (u64)(v >> 64));
}
// a load at (uniform base) + (the lane's 32-bit byte offset): the base stays in an SGPR pair (the empty statement hides the
// offset's value, which keeps the optimiser from folding it into per-column VGPR pointers outside the loop: eight registers
// that cost the kernel its fourth wave)
__device__ __forceinline__ u64 evalw_load(const char* base, u32 off) {
    asm("" : "+v"(off));
    return *reinterpret_cast<const u64*>(base + off);
}
// a w + c = one v_mad_u64_u32 with w from an SGPR.  The empty statement pins the order of the additions: without it the optimiser
// adds the two products of a limb position first and the running sum last -- a third instruction per pair.
__device__ __forceinline__ u64 evalw_mad(u32 a, u32 w, u64 c) {
    u64 d = c + (u64)a * w;
    asm("" : "+v"(d));
    return d;
}
__device__ __forceinline__ u64 wave_sum_gl(u64 v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const u32 lo = (u32)__shfl_xor((int)(u32)v, s, 64), hi = (u32)__shfl_xor((int)(u32)(v >> 32), s, 64);
        v = gl_add(v, ((u64)hi << 32) | lo);
    }
    return v;
}
template <int C, int NP>
__global__ __launch_bounds__(256) void eval_points_wide_kernel(const u64* __restrict__ coeffs, size_t n, int ncols, ExtPow p0, ExtPow p1,
                                                               unsigned nlo_blocks, unsigned kh_per_block, u64* __restrict__ partial) {
    __shared__ u64 red[4][C * NP * 2];
    const int h = p0.h;
    const unsigned lb = blockIdx.x % nlo_blocks, split = blockIdx.x / nlo_blocks;
    const size_t kl = (size_t)lb * 256 + threadIdx.x;
    const int c0 = blockIdx.y * C;
    const unsigned kh0 = split * kh_per_block, kh1 = kh0 + kh_per_block;
    u64 acc[C][NP][2][3];
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int i = 0; i < 3; i++) acc[c][p][e][i] = 0;
    // addresses: a uniform base per column and step (SGPRs) plus the thread's byte offset (one VGPR); the coefficients of step
    // kh + 1 are asked for before step kh is multiplied
    const u32 koff = (u32)(kl * 8);
    const char* cbase[C];
#pragma unroll
    for (int c = 0; c < C; c++) cbase[c] = reinterpret_cast<const char*>(coeffs + (size_t)min(c0 + c, ncols - 1) * n);     // surplus columns redo the last one
    const u32* __restrict__ L0 = p0.hi_limbs;
    const u32* __restrict__ L1 = p1.hi_limbs;
    auto load = [&](u64 (&f)[C], unsigned kh) {
#pragma unroll
        for (int c = 0; c < C; c++) f[c] = evalw_load(cbase[c] + (((size_t)kh << h) << 3), koff);
    };
    auto step = [&](const u64 (&f)[C], unsigned kh) {
        u32 w[NP][12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            w[0][i] = L0[(size_t)kh * 12 + i];
            if (NP > 1) w[NP - 1][i] = L1[(size_t)kh * 12 + i];
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            const u32 f0 = (u32)f[c], f1 = (u32)(f[c] >> 32);
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int i = 0; i < 3; i++) {       // two multiply-adds INTO the sum (written as one expression the optimiser adds the products first: a third instruction)
                        acc[c][p][e][i] = evalw_mad(f0, w[p][6 * e + i], acc[c][p][e][i]);
                        acc[c][p][e][i] = evalw_mad(f1, w[p][6 * e + 3 + i], acc[c][p][e][i]);
                    }
        }
    };
    // two steps per turn (kh_per_block is even), each multiplying one buffer while the other one's loads are in flight
    u64 fa[C], fb[C];
    load(fa, kh0);
    for (unsigned kh = kh0; kh < kh1; kh += 2) {
        load(fb, kh + 1);
        step(fa, kh);
        load(fa, kh + 2 < kh1 ? kh + 2 : kh);
        step(fb, kh + 1);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const ExtPow& tp = p == 0 ? p0 : p1;
        const Ext2 wl = ext_make(tp.lo_a[kl], tp.lo_b[kl]);
#pragma unroll
        for (int c = 0; c < C; c++) {
            const Ext2 v = ext_mul(ext_make(evalw_fold(acc[c][p][0]), evalw_fold(acc[c][p][1])), wl);
            const u64 sa = wave_sum_gl(v.a), sb = wave_sum_gl(v.b);
            if (lane == 0) { red[wave][(c * NP + p) * 2] = sa; red[wave][(c * NP + p) * 2 + 1] = sb; }
        }
    }
    __syncthreads();
    if (threadIdx.x < C * NP * 2) {
        const int idx = threadIdx.x, e = idx & 1, p = (idx >> 1) % NP, c = (idx >> 1) / NP;
        if (c0 + c < ncols) {
            const u64 sum = gl_add(gl_add(red[0][idx], red[1][idx]), gl_add(red[2][idx], red[3][idx]));
            partial[(((size_t)p * gridDim.x + blockIdx.x) * ncols + (c0 + c)) * 2 + e] = sum;
        }
    }
}

// Composition polynomials of the three FRI batches (fri/oracle.rs:193-211, circuits stark.rs:117-141), thread per k:
//   C1 = sum_{i<W+Z} alpha^i f_i     (batch g*zeta: trace, zs)
//   C0 = C1 + sum_j alpha^(W+Z+j) q_j (batch zeta: trace, zs, quotient)
//   C2 = sum_j alpha^j zs[nperm + j]  (batch g^-1: ctl zs)
// out: 6 planes of n: C0.a C0.b C1.a C1.b C2.a C2.b ; alpha powers in apow (a-plane then b-plane, napow entries each)
__global__ __launch_bounds__(256) void compose_kernel(const u64* __restrict__ trace, int W, const u64* __restrict__ zs, int Z,
                                                      const u64* __restrict__ quot, int Q, int nperm, size_t n,
                                                      const u64* __restrict__ apow, int napow, u64* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Ext2 c1 = ext_make(0, 0), cq = ext_make(0, 0), c2 = ext_make(0, 0);
    for (int i = 0; i < W; i++) c1 = ext_add(c1, ext_scalar_mul(ext_make(apow[i], apow[napow + i]), trace[(size_t)i * n + k]));
    for (int i = 0; i < Z; i++) {
        const u64 f = zs[(size_t)i * n + k];
        c1 = ext_add(c1, ext_scalar_mul(ext_make(apow[W + i], apow[napow + W + i]), f));
        if (i >= nperm) c2 = ext_add(c2, ext_scalar_mul(ext_make(apow[i - nperm], apow[napow + i - nperm]), f));
    }
    for (int i = 0; i < Q; i++)
        cq = ext_add(cq, ext_scalar_mul(ext_make(apow[W + Z + i], apow[napow + W + Z + i]), quot[(size_t)i * n + k]));
    const Ext2 c0 = ext_add(c1, cq);
    out[k] = c0.a; out[n + k] = c0.b;
    out[2 * n + k] = c1.a; out[3 * n + k] = c1.b;
    out[4 * n + k] = c2.a; out[5 * n + k] = c2.b;
}

// D[k] = C[k] * z^k, in place on a pair of planes
__global__ __launch_bounds__(256) void weight_kernel(u64* __restrict__ pa, u64* __restrict__ pb, size_t n, ExtPow zp) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Ext2 d = ext_mul(ext_make(pa[k], pb[k]), ext_pow_lookup(zp, k));
    pa[k] = d.a;
    pb[k] = d.b;
}

// Additive suffix scan over one plane of base elements, 3 phases; blocks of SCAN_B elements.
#define SCAN_B 2048
__global__ __launch_bounds__(256) void scan_local_kernel(u64* __restrict__ d, size_t n, u64* __restrict__ block_tot) {
    __shared__ u64 sh[256];
    const size_t b0 = (size_t)blockIdx.x * SCAN_B;
    const int t = threadIdx.x;
    // thread t owns elements [b0 + 8t, b0 + 8t + 8); suffix order: later indices first
    u64 v[8];
    u64 run = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        const size_t k = b0 + (size_t)t * 8 + i;
        run = gl_add(run, k < n ? d[k] : 0);
        v[i] = run;
    }
    sh[t] = run;
    __syncthreads();
    // exclusive suffix scan of the thread totals (Hillis-Steele over 256 entries)
    u64 incl = run;
    for (int s = 1; s < 256; s <<= 1) {
        const u64 other = (t + s < 256) ? sh[t + s] : 0;
        __syncthreads();
        incl = gl_add(incl, other);
        sh[t] = incl;
        __syncthreads();
    }
    const u64 excl = gl_sub(incl, run);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t k = b0 + (size_t)t * 8 + i;
        if (k < n) d[k] = gl_add(v[i], excl);
    }
    if (t == 0) block_tot[blockIdx.x] = incl;
}
// single block: exclusive suffix scan of nblocks totals in place
__global__ __launch_bounds__(256) void scan_totals_kernel(u64* __restrict__ tot, size_t nblocks) {
    // exclusive suffix sum of the block totals, one workgroup: thread t owns a contiguous chunk (chunks in reverse order)
    __shared__ u64 sh[256];
    const int t = threadIdx.x;
    const size_t per = (nblocks + 255) / 256;
    // chunk of thread t counted from the END: indices [nblocks - hi_off, nblocks - lo_off)
    const size_t lo_off = (size_t)t * per < nblocks ? (size_t)t * per : nblocks;
    const size_t hi_off = lo_off + per < nblocks ? lo_off + per : nblocks;
    u64 run = 0;
    for (size_t o = lo_off; o < hi_off; o++) run = gl_add(run, tot[nblocks - 1 - o]);
    sh[t] = run;
    __syncthreads();
    u64 incl = run;
    for (int s = 1; s < 256; s <<= 1) {
        const u64 other = (t >= s) ? sh[t - s] : 0;
        __syncthreads();
        incl = gl_add(incl, other);
        sh[t] = incl;
        __syncthreads();
    }
    u64 acc = (t > 0) ? sh[t - 1] : 0;
    for (size_t o = lo_off; o < hi_off; o++) {
        const size_t i = nblocks - 1 - o;
        const u64 v = tot[i];
        tot[i] = acc;
        acc = gl_add(acc, v);
    }
}
__global__ __launch_bounds__(256) void scan_add_kernel(u64* __restrict__ d, size_t n, const u64* __restrict__ tot) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    d[k] = gl_add(d[k], tot[k / SCAN_B]);
}

// final[k] = sum_b w_b * S_b[k] * z_b^-k  (k >= 1), final[0] = 0.  S planes as produced by compose/weight/scan.
// fa/fb have length N >= n and are zero beyond n.
__global__ __launch_bounds__(256) void finalize_kernel(const u64* __restrict__ S, size_t n, ExtPow zi0, ExtPow zi1, ExtPow zi2,
                                                       Ext2 w0, Ext2 w1, Ext2 w2, int use2, u64* __restrict__ fa,
                                                       u64* __restrict__ fb, size_t N) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    Ext2 r = ext_make(0, 0);
    if (k >= 1 && k < n) {
        r = ext_mul(w0, ext_mul(ext_make(S[k], S[n + k]), ext_pow_lookup(zi0, k)));
        r = ext_add(r, ext_mul(w1, ext_mul(ext_make(S[2 * n + k], S[3 * n + k]), ext_pow_lookup(zi1, k))));
        if (use2) r = ext_add(r, ext_mul(w2, ext_mul(ext_make(S[4 * n + k], S[5 * n + k]), ext_pow_lookup(zi2, k))));
    }
    fa[k] = r.a;
    fb[k] = r.b;
}

// FRI fold in the coefficient domain (fri/prover.rs:102-109): out[j] = sum_{k<arity} beta^k c[arity*j + k]
__global__ __launch_bounds__(256) void fold_kernel(const u64* __restrict__ ca, const u64* __restrict__ cb, size_t out_len, int arity,
                                                   Ext2 beta, u64* __restrict__ oa, u64* __restrict__ ob) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= out_len) return;
    Ext2 s = ext_make(0, 0);
    for (int k = arity - 1; k >= 0; k--) s = ext_add(ext_mul(s, beta), ext_make(ca[j * arity + k], cb[j * arity + k]));
    oa[j] = s.a;
    ob[j] = s.b;
}

// The same fold for arity 16 with loads on consecutive addresses (round 6): thread j of fold_kernel reads sixteen words 128 bytes
// apart from thread j + 1's -- 16.4 GB fetched per 2^22-row proof for 1.1 GB of coefficients (profiles/r05_proof_pmc_blake3.txt).
// Here a thread loads TWO neighbouring coefficients of both planes with 16-byte loads, eight neighbouring lanes hold one output's
// sixteen, and three exchanges join them: s = c_even + beta c_odd, then pairs by beta^2, beta^4, beta^8.  `nz_out` outputs are
// computed (the caller zero-fills the rest: the first layer's planes are zero beyond the n coefficients, 7/8 of their length).
__device__ __forceinline__ Ext2 ext_shfl_xor(Ext2 v, int m) {
    Ext2 r;
    r.a = ((u64)(u32)__shfl_xor((int)(u32)(v.a >> 32), m, 64) << 32) | (u32)__shfl_xor((int)(u32)v.a, m, 64);
    r.b = ((u64)(u32)__shfl_xor((int)(u32)(v.b >> 32), m, 64) << 32) | (u32)__shfl_xor((int)(u32)v.b, m, 64);
    return r;
}
__global__ __launch_bounds__(256) void fold16_kernel(const u64* __restrict__ ca, const u64* __restrict__ cb, size_t nz_out, Ext2 b1, Ext2 b2, Ext2 b4,
                                                     Ext2 b8, u64* __restrict__ oa, u64* __restrict__ ob) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // pair index: coefficients 2t, 2t + 1
    const size_t last = nz_out * 8 - 1;
    const size_t tt = t < last ? t : last;                               // surplus lanes redo the last pair: the wave stays whole
    const ulonglong2 va = *reinterpret_cast<const ulonglong2*>(ca + 2 * tt);
    const ulonglong2 vb = *reinterpret_cast<const ulonglong2*>(cb + 2 * tt);
    Ext2 s = ext_add(ext_make(va.x, vb.x), ext_mul(ext_make(va.y, vb.y), b1));
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int m = 1 << r;
        const Ext2 pw = r == 0 ? b2 : (r == 1 ? b4 : b8);
        const Ext2 o = ext_shfl_xor(s, m);
        const bool up = (lane & m) != 0;
        const Ext2 lo = up ? o : s, hi = up ? s : o;
        s = ext_add(lo, ext_mul(hi, pw));
    }
    if ((t & 7) == 0 && t <= last) {
        oa[t >> 3] = s.a;
        ob[t >> 3] = s.b;
    }
}

// out[q][2k], out[q][2k+1] = (a, b)[idx[q]*arity + k]
__global__ void gather_ext_leaves_kernel(const u64* __restrict__ pa, const u64* __restrict__ pb, int arity,
                                         const unsigned long long* __restrict__ idx, u64* __restrict__ out) {
    const int q = blockIdx.x, k = threadIdx.x;
    if (k >= arity) return;
    out[((size_t)q * arity + k) * 2] = pa[idx[q] * arity + k];
    out[((size_t)q * arity + k) * 2 + 1] = pb[idx[q] * arity + k];
}

// ------------------------------------------------------------------------------------------------ host helpers
// host memory that outlives the asynchronous copies issued from / into it: pinned when the context's arena has room
struct HostSpan {
    u64* p = nullptr; size_t n = 0;
    u64& operator[](size_t i) const { return p[i]; }
    u64* data() const { return p; }
    size_t size() const { return n; }
    u64* begin() const { return p; }
    u64* end() const { return p + n; }
};
struct DevBuf {
    DeviceCtx* ctx;
    std::vector<void*> ptrs;
    // Host staging of asynchronous copies (descriptors, power tables, index lists, read-backs): lives as long as the scope, whose
    // destructor drains the stream first -- so an upload needs no synchronisation of its own (each one stalled the host's
    // run-ahead: 14 of the 22 synchronisations per table were of this kind; the small tables are bound by exactly that latency).
    // Pinned (the context's arena, DeviceCtx::pinned_alloc) while it has room, pageable otherwise.
    std::deque<std::vector<u64>> hosts;
    size_t pinned_mark;
    struct Pending { void* dst; const u64* src; size_t bytes; };
    std::vector<Pending> pending;      // read-backs that landed in pinned memory and still have to reach the caller's buffer
    bool use_pinned = true;            // false: a scope that outlives its call (OlaFri) -- the arena is a stack and is left to the calls
    HostSpan host(size_t elems) {
        if (use_pinned)
            if (void* p = ctx->pinned_alloc(std::max<size_t>(1, elems) * 8)) return {(u64*)p, elems};
        hosts.emplace_back(std::max<size_t>(1, elems));
        return {hosts.back().data(), elems};
    }
    u64* upload(const std::vector<u64>& v) {
        HostSpan h = host(v.size());
        std::copy(v.begin(), v.end(), h.begin());
        u64* d = alloc(std::max<size_t>(1, v.size()));
        if (!v.empty()) HIP_CHECK(hipMemcpyAsync(d, h.data(), v.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        return d;
    }
    // device -> the caller's (pageable) buffer, complete after the next synchronisation + collect(): through pinned memory, so
    // that the copy is really asynchronous
    void readback(void* dst, const void* src_dev, size_t bytes) {
        if (!bytes) return;
        if (void* p = use_pinned ? ctx->pinned_alloc(bytes) : nullptr) {
            HIP_CHECK(hipMemcpyAsync(p, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
            pending.push_back({dst, (const u64*)p, bytes});
        } else {
            HIP_CHECK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    // after a synchronisation of the context's stream
    void collect() {
        for (const Pending& r : pending) memcpy(r.dst, r.src, r.bytes);
        pending.clear();
    }
    void sync_collect() {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        collect();
    }
    explicit DevBuf(DeviceCtx* c) : ctx(c), pinned_mark(c->pinned_top) {}
    u64* alloc(size_t elems) { return (u64*)alloc_bytes(elems * 8); }
    void* alloc_bytes(size_t bytes) {
        void* p = ctx->alloc(bytes);
        ptrs.push_back(p);
        return p;
    }
    ~DevBuf() {
        (void)hipStreamSynchronize(ctx->stream);
        for (void* p : ptrs) ctx->free(p);
        if (use_pinned) ctx->pinned_top = pinned_mark;      // (a scope that never took from the arena does not rewind it either)
    }
};

// two-level power tables of `count` extension points, one host buffer, one upload (a table's opening phase needs six: zeta,
// g zeta, g^-1 and their inverses -- six copies of a few hundred bytes each before round 6).  The first `nlimb` of them also get
// the limb form of their hi table (eval_points_wide_kernel).
static void make_ext_pows(DevBuf& mem, const Ext2* z, int count, int log_n, ExtPow* out, int nlimb = 0) {
    const int h = (log_n + 1) / 2;
    const size_t nlo = (size_t)1 << h, nhi = (size_t)1 << (log_n - h), per = 2 * (nlo + nhi), lper = 6 * nhi;   // 12 u32 per kh
    const size_t total = per * (size_t)count + lper * (size_t)nlimb;
    HostSpan host = mem.host(total);
    u64* d = mem.alloc(total);
    for (int i = 0; i < count; i++) {
        u64* hp = host.data() + per * (size_t)i;
        Ext2 acc = ext_make(1, 0);
        for (size_t k = 0; k < nlo; k++) { hp[k] = acc.a; hp[nlo + k] = acc.b; acc = ext_mul(acc, z[i]); }
        const Ext2 zh = acc;  // z^(2^h)
        acc = ext_make(1, 0);
        for (size_t k = 0; k < nhi; k++) { hp[2 * nlo + k] = acc.a; hp[2 * nlo + nhi + k] = acc.b; acc = ext_mul(acc, zh); }
        u64* dp = d + per * (size_t)i;
        out[i].lo_a = dp; out[i].lo_b = dp + nlo; out[i].hi_a = dp + 2 * nlo; out[i].hi_b = dp + 2 * nlo + nhi; out[i].h = h;
        out[i].hi_limbs = nullptr;
        if (i < nlimb) {
            u32* lp = reinterpret_cast<u32*>(host.data() + per * (size_t)count + lper * (size_t)i);
            for (size_t k = 0; k < nhi; k++)
                for (int e = 0; e < 2; e++) {
                    const u64 w = gl_canon(hp[2 * nlo + (e ? nhi : 0) + k]), ws = gl_mul(w, (u64)1 << 32);
                    u32* q = lp + 12 * k + 6 * e;
                    q[0] = (u32)(w & 0x3FFFFF); q[1] = (u32)((w >> 22) & 0x3FFFFF); q[2] = (u32)(w >> 44);
                    q[3] = (u32)(ws & 0x3FFFFF); q[4] = (u32)((ws >> 22) & 0x3FFFFF); q[5] = (u32)(ws >> 44);
                }
            out[i].hi_limbs = reinterpret_cast<const u32*>(d + per * (size_t)count + lper * (size_t)i);
        }
    }
    HIP_CHECK(hipMemcpyAsync(d, host.data(), total * 8, hipMemcpyHostToDevice, mem.ctx->stream));
}

// Opened rows and Merkle paths of the query indices.  Under the coset partition a leaf lives on the rank that owns its
// coset: every rank fills the records of its own queries, the records are all-gathered and each query is read from its
// owner's copy (paths end at the owner's cap slice, which is a slice of the full cap).
static void query_leaves(DevBuf& mem, NttTables& tables, const OlaBatch& b, const size_t* xs, int nq, int depth, u64* rows_out, u64* paths_out) {
    DeviceCtx* ctx = mem.ctx;
    if (!b.is_shard()) {
        // what a partitioned run gathers here (records of nq queries from each of 8 ranks), counted for the one-GPU projection
        if (ctx->acct.shardable && ctx->shard.world <= 1) acct_exchange(ctx, (size_t)nq * (b.ncols + (size_t)std::max(depth, 0) * 4) * 8 * 8);
        if (b.lean) { batch_get_leaves(ctx, b, xs, nq, rows_out, paths_out, &tables); return; }
        // resident batch: enqueued only -- rows_out / paths_out are complete after the caller's next synchronisation (the three
        // oracles and the FRI layers of a proof share ONE; batch_get_leaves, the C-ABI accessor's path, waits per call)
        HostSpan h_idx = mem.host((size_t)nq);
        for (int r = 0; r < nq; r++) h_idx[(size_t)r] = (u64)xs[r];
        unsigned long long* d_idx = (unsigned long long*)mem.alloc((size_t)nq);
        HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), (size_t)nq * 8, hipMemcpyHostToDevice, ctx->stream));
        u64* d_rows = mem.alloc((size_t)nq * b.ncols);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nq), dim3(64), 0, ctx->stream, b.lde, b.num_leaves(), (int)b.ncols, d_idx, d_rows);
        mem.readback(rows_out, d_rows, (size_t)nq * b.ncols * 8);
        if (depth > 0) {
            u64* d_paths = mem.alloc((size_t)nq * (size_t)depth * 4);
            hipLaunchKernelGGL(gather_paths_kernel, dim3((unsigned)nq), dim3(((depth * 4 + 63) / 64) * 64), 0, ctx->stream, b.heap, b.num_leaves(), depth, d_idx, d_paths);
            mem.readback(paths_out, d_paths, (size_t)nq * (size_t)depth * 32);
        }
        return;
    }
    const size_t n_loc = b.num_leaves(), first = (size_t)b.coset_first << b.log_n;
    const size_t dwords = (size_t)std::max(depth, 0) * 4, rec = b.ncols + dwords;
    const uint32_t world = ctx->shard.world;
    std::vector<size_t> mine_idx;
    std::vector<int> mine_pos;
    for (int r = 0; r < nq; r++)
        if (xs[r] >= first && xs[r] < first + n_loc) { mine_idx.push_back(xs[r] - first); mine_pos.push_back(r); }
    std::vector<u64> lrows(mine_idx.size() * b.ncols), lpaths(mine_idx.size() * std::max<size_t>(dwords, 1));
    if (!mine_idx.empty()) batch_get_leaves(ctx, b, mine_idx.data(), mine_idx.size(), lrows.data(), lpaths.data(), &tables);
    std::vector<u64> send((size_t)nq * rec, 0), recv((size_t)world * nq * rec);
    for (size_t k = 0; k < mine_idx.size(); k++) {
        u64* d = send.data() + (size_t)mine_pos[k] * rec;
        std::copy(lrows.begin() + k * b.ncols, lrows.begin() + (k + 1) * b.ncols, d);
        std::copy(lpaths.begin() + k * dwords, lpaths.begin() + (k + 1) * dwords, d + b.ncols);
    }
    u64* d_send = (u64*)ctx->alloc(send.size() * 8);
    u64* d_recv = (u64*)ctx->alloc(recv.size() * 8);
    try {
        HIP_CHECK(hipMemcpyAsync(d_send, send.data(), send.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        shard_all_gather(ctx, d_send, d_recv, send.size() * 8);
        HIP_CHECK(hipMemcpyAsync(recv.data(), d_recv, recv.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } catch (...) { ctx->free(d_send); ctx->free(d_recv); throw; }
    ctx->free(d_send);
    ctx->free(d_recv);
    for (int r = 0; r < nq; r++) {
        const size_t owner = xs[r] / n_loc;
        const u64* d = recv.data() + (owner * nq + r) * rec;
        std::copy(d, d + b.ncols, rows_out + (size_t)r * b.ncols);
        std::copy(d + b.ncols, d + rec, paths_out + (size_t)r * dwords);
    }
}

// all polynomials of a batch at one or two extension points -> host vectors
// (two phases so that the evaluations of all oracles share one synchronisation: launch + async read-back, then collect)
// Under the coset partition the COLUMNS of a batch are divided among the ranks (every rank holds all coefficients: the
// interpolation is replicated): rank r evaluates columns [r * cpr, (r+1) * cpr), reduces its chunk partials on the device, and
// the evaluations of all four jobs of a table travel in ONE all-gather of a few KB (proof.rs:198-233).
struct EvalJob {
    std::vector<u64> part;
    unsigned nchunks = 0;
    uint32_t ncols = 0;
    int npoints = 0;
    std::vector<Ext2>* out0 = nullptr;
    std::vector<Ext2>* out1 = nullptr;
    // sharded form
    uint32_t cpr = 0;           // columns per rank
    size_t send_off = 0;        // this job's slot in a rank's record: [npoints][cpr][2] words
};
// partial[pt][chunk][col][2] -> out[pt][cpr][2] (columns beyond ncols_loc stay zero); one wave per (point, column)
__global__ __launch_bounds__(64) void eval_reduce_kernel(const u64* __restrict__ partial, unsigned nchunks, int ncols_loc, int npoints,
                                                         int cpr, u64* __restrict__ out) {
    const int p = blockIdx.x / cpr, c = blockIdx.x % cpr;
    (void)npoints;
    u64 a = 0, b = 0;
    if (c < ncols_loc)
        for (unsigned ch = threadIdx.x; ch < nchunks; ch += 64) {
            const u64* v = partial + (((size_t)p * nchunks + ch) * ncols_loc + c) * 2;
            a = gl_add(a, v[0]);
            b = gl_add(b, v[1]);
        }
    a = wave_sum_gl(a);
    b = wave_sum_gl(b);
    if (threadIdx.x == 0) {
        out[((size_t)p * cpr + c) * 2] = a;
        out[((size_t)p * cpr + c) * 2 + 1] = b;
    }
}
// one batch of columns at one or two points -> chunk partials; returns the number of chunks
#define EVALW_C 4
static unsigned eval_points_plan(size_t n, int log_n, uint32_t ncols, const ExtPow& p0, const ExtPow& p1, size_t* chunk_len, unsigned* khpb) {
    static const bool wide_off = getenv("OLA_EVAL_WIDE") && !strcmp(getenv("OLA_EVAL_WIDE"), "0");
    *khpb = 0;
    if (!wide_off && log_n >= 15 && p0.hi_limbs && p1.hi_limbs && ncols) {
        const int h = p0.h;
        const size_t nlo_blocks = ((size_t)1 << h) / 256, nhi = (size_t)1 << (log_n - h), groups = (ncols + EVALW_C - 1) / EVALW_C;
        // enough workgroups to fill the chip several times over, as many steps per workgroup as that leaves (the fold, the
        // multiplication by lo[kl] and the sum over the workgroup cost about twenty steps)
        size_t per = EVALW_MAX_KH;
        while (per > 32 && nlo_blocks * groups * (nhi / per) < 2048) per >>= 1;
        per = std::min(per, nhi);
        *khpb = (unsigned)per;
        return (unsigned)(nlo_blocks * (nhi / per));
    }
    *chunk_len = std::max<size_t>(4096, (n + 255) / 256);
    return (unsigned)((n + *chunk_len - 1) / *chunk_len);
}
static void eval_points_launch(DeviceCtx* ctx, const u64* coeffs, size_t n, uint32_t ncols, const ExtPow& p0, const ExtPow& p1, int npoints,
                               unsigned nchunks, size_t chunk_len, unsigned khpb, u64* d_part) {
    if (khpb) {
        const dim3 grid(nchunks, (ncols + EVALW_C - 1) / EVALW_C);
        const unsigned nlo_blocks = (unsigned)(((size_t)1 << p0.h) / 256);
        if (npoints > 1)
            hipLaunchKernelGGL((eval_points_wide_kernel<EVALW_C, 2>), grid, dim3(256), 0, ctx->stream, coeffs, n, (int)ncols, p0, p1, nlo_blocks, khpb, d_part);
        else
            hipLaunchKernelGGL((eval_points_wide_kernel<EVALW_C, 1>), grid, dim3(256), 0, ctx->stream, coeffs, n, (int)ncols, p0, p0, nlo_blocks, khpb, d_part);
        return;
    }
    hipLaunchKernelGGL(eval_points_kernel, dim3(nchunks, (ncols + EVAL_CG - 1) / EVAL_CG), dim3(256), 0, ctx->stream, coeffs, n, (int)ncols, p0, p1,
                       npoints, chunk_len, d_part);
}
static void eval_batch_launch(DevBuf& mem, const OlaBatch& b, int npoints, ExtPow p0, ExtPow p1, std::vector<Ext2>* out0,
                              std::vector<Ext2>* out1, EvalJob& job, u64* d_send = nullptr) {
    DeviceCtx* ctx = mem.ctx;
    const size_t n = b.n();
    size_t chunk_len = 0;
    unsigned khpb = 0;
    const uint32_t cols_here = d_send ? std::min(b.ncols, std::min(b.ncols, ctx->shard.rank * job.cpr) + job.cpr) - std::min(b.ncols, ctx->shard.rank * job.cpr) : b.ncols;
    const unsigned nchunks = eval_points_plan(n, (int)b.log_n, cols_here, p0, p1, &chunk_len, &khpb);
    job.nchunks = nchunks; job.ncols = b.ncols; job.npoints = npoints; job.out0 = out0; job.out1 = out1;
    if (d_send) {           // this rank's columns only
        const uint32_t cpr = job.cpr, c0 = std::min(b.ncols, ctx->shard.rank * cpr), c1 = std::min(b.ncols, c0 + cpr);
        const uint32_t mine = c1 - c0;
        u64* d_part = mem.alloc(std::max<size_t>(1, (size_t)npoints * nchunks * mine * 2));
        if (mine) eval_points_launch(ctx, b.coeffs + (size_t)c0 * n, n, mine, p0, p1, npoints, nchunks, chunk_len, khpb, d_part);
        hipLaunchKernelGGL(eval_reduce_kernel, dim3((unsigned)(npoints * cpr)), dim3(64), 0, ctx->stream, d_part, nchunks, (int)mine, npoints, (int)cpr,
                           d_send + job.send_off);
        return;
    }
    const size_t pelems = (size_t)npoints * nchunks * b.ncols * 2;
    u64* d_part = mem.alloc(pelems);
    eval_points_launch(ctx, b.coeffs, n, b.ncols, p0, p1, npoints, nchunks, chunk_len, khpb, d_part);
    if (nchunks > 16 && b.ncols) {      // many chunks: add them up on the device, the host reads one record per column
        u64* d_sum = mem.alloc((size_t)npoints * b.ncols * 2);
        hipLaunchKernelGGL(eval_reduce_kernel, dim3((unsigned)(npoints * b.ncols)), dim3(64), 0, ctx->stream, d_part, nchunks, (int)b.ncols, npoints,
                           (int)b.ncols, d_sum);
        job.nchunks = 1;
        job.part.resize((size_t)npoints * b.ncols * 2);
        mem.readback(job.part.data(), d_sum, job.part.size() * 8);
        return;
    }
    job.part.resize(pelems);
    mem.readback(job.part.data(), d_part, pelems * 8);
}
static void eval_batch_collect(const EvalJob& job) {
    for (int p = 0; p < job.npoints; p++) {
        std::vector<Ext2>& o = p == 0 ? *job.out0 : *job.out1;
        o.assign(job.ncols, ext_make(0, 0));
        for (unsigned ch = 0; ch < job.nchunks; ch++)
            for (uint32_t c = 0; c < job.ncols; c++) {
                const u64* v = &job.part[(((size_t)p * job.nchunks + ch) * job.ncols + c) * 2];
                o[c] = ext_add(o[c], ext_make(v[0], v[1]));
            }
    }
}
// sharded: recv = [world][record], a record = the jobs' slots one after the other
static void eval_batch_collect_sharded(const EvalJob& job, const u64* recv, size_t record_words) {
    for (int p = 0; p < job.npoints; p++) {
        std::vector<Ext2>& o = p == 0 ? *job.out0 : *job.out1;
        o.assign(job.ncols, ext_make(0, 0));
        for (uint32_t c = 0; c < job.ncols; c++) {
            const u64* v = recv + (size_t)(c / job.cpr) * record_words + job.send_off + ((size_t)p * job.cpr + c % job.cpr) * 2;
            o[c] = ext_make(v[0], v[1]);
        }
    }
}

static void scan_plane(DevBuf& mem, u64* plane, size_t n, u64* tot) {
    DeviceCtx* ctx = mem.ctx;
    const size_t nblocks = (n + SCAN_B - 1) / SCAN_B;
    hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, plane, n, tot);
    if (nblocks > 1) {
        hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(256), 0, ctx->stream, tot, nblocks);
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, plane, n, tot);
    }
}

struct ByteWriter {
    std::vector<uint8_t>& b;
    bool bytes_hash = false;   // Blake3 digests are 32 bytes (BytesHash::to_bytes), written as they are; a HashOut is 4 canonical words
    void u8(uint8_t x) { b.push_back(x); }
    void u32(uint32_t x) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void field(u64 x) { x = gl_canon(x); for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void ext(Ext2 e) { field(e.a); field(e.b); }
    void ext_vec(const std::vector<Ext2>& v) { u32((uint32_t)v.size()); for (auto& e : v) ext(e); }
    void field_vec(const u64* v, size_t n) { u32((uint32_t)n); for (size_t i = 0; i < n; i++) field(v[i]); }
    void hash(const u64* h) {
        for (int i = 0; i < 4; i++) {
            if (bytes_hash) { for (int k = 0; k < 8; k++) b.push_back((uint8_t)(h[i] >> (8 * k))); }
            else field(h[i]);
        }
    }
    void cap(const u64* c, size_t len) { u32((uint32_t)len); for (size_t i = 0; i < len; i++) hash(c + 4 * i); }
    void merkle_proof(const u64* sib, int depth) { u8((uint8_t)depth); for (int i = 0; i < depth; i++) hash(sib + 4 * i); }
};

static std::vector<int> fri_arities(const OlaGpuConfig& cfg, int degree_bits) {
    // fri/reduction_strategies.rs:40-52  ConstantArityBits(arity_bits, final_poly_bits)
    std::vector<int> v;
    int d = degree_bits;
    while (d > (int)cfg.fri_final_poly_bits && d + (int)cfg.rate_bits - (int)cfg.fri_arity_bits >= (int)cfg.cap_height) {
        v.push_back((int)cfg.fri_arity_bits);
        d -= (int)cfg.fri_arity_bits;
    }
    return v;
}

struct FriLayer {
    u64* va; u64* vb;  // bit-reversed evaluations (planes), length len
    u64* heap;         // 2 * (len/arity) digests
    size_t len;
    int arity_bits;
    std::vector<u64> cap;
    size_t shard_leaves = 0;   // > 0: this rank holds leaves [rank * shard_leaves, (rank+1) * shard_leaves) only (first layer on the partition)
};

// The proof-of-work witness of a table (fri/prover.rs:126-148) enters nothing but the proof bytes: the query indices are drawn
// from the transcript as it stands after the final polynomial (fri/prover.rs:53-62: current_hash, then fri_prover_query_rounds),
// and the next table's transcript does not see it either.  Inside a whole proof the search therefore runs on the context's side
// stream -- 120 us of Poseidon permutations per table that the launch-bound small tables no longer wait for, and one host round
// trip less per table -- and the witnesses are patched into the bytes when the proof is complete (pow_finish).
struct PowDefer {
    struct Job { size_t at; u64 h[4]; u32 bits; unsigned long long* d_best; unsigned long long* h_best; };
    std::vector<Job> jobs;
    unsigned long long* slots = nullptr;   // pinned, one per table, allocated by the scope that owns the whole proof
    size_t nslots = 0;
};
static u64 pow_batch(u32 bits) { return std::max<u64>((u64)1 << 14, (u64)4 << bits); }
// enqueue the first batch of the search on the side stream; false: no side stream / no slot, search synchronously
static bool pow_enqueue(DeviceCtx* ctx, PowDefer* d, const u64 h[4], u32 bits, size_t at) {
    if (!d || !d->slots || d->jobs.size() >= d->nslots || d->jobs.size() >= DeviceCtx::kSideWords) return false;
    hipStream_t side = ctx->side_stream();
    if (!side) return false;
    if (!ctx->side_words) ctx->side_words = (unsigned long long*)ctx->alloc_persistent(DeviceCtx::kSideWords * 8);
    PowDefer::Job j;
    j.at = at; j.bits = bits;
    for (int i = 0; i < 4; i++) j.h[i] = h[i];
    j.h_best = d->slots + d->jobs.size();
    *j.h_best = ~0ull;
    j.d_best = ctx->side_words + d->jobs.size();
    HIP_CHECK(hipMemsetAsync(j.d_best, 0xFF, 8, side));
    hipLaunchKernelGGL(pow_kernel, dim3((unsigned)(pow_batch(bits) / 256)), dim3(256), 0, side, h[0], h[1], h[2], h[3], (u64)0, bits, j.d_best);
    HIP_CHECK(hipMemcpyAsync(j.h_best, j.d_best, 8, hipMemcpyDeviceToHost, side));
    d->jobs.push_back(j);
    return true;
}
// all searches done: patch the witnesses (8 little-endian bytes each, serialization.rs:50-52) into the proof
void pow_finish(DeviceCtx* ctx, PowDefer& d, std::vector<uint8_t>& bytes) {
    if (d.jobs.empty()) return;
    const hipError_t e = hipStreamSynchronize(ctx->side);
    if (e != hipSuccess) { d.jobs.clear(); HIP_CHECK(e); }
    for (PowDefer::Job& j : d.jobs) {
        u64 w = *j.h_best;
        if (w == ~0ull) w = run_pow(ctx, j.h, j.bits);      // not in the first batch (2 % of the searches): the plain loop
        w = gl_canon(w);
        for (int k = 0; k < 8; k++) bytes[j.at + (size_t)k] = (uint8_t)(w >> (8 * k));
    }
    d.jobs.clear();
}
// an exception unwound the proof: the searches may still be running
void pow_abandon(DeviceCtx* ctx, PowDefer& d) {
    if (d.jobs.empty()) return;
    if (ctx->side) (void)hipStreamSynchronize(ctx->side);
    d.jobs.clear();
}

// ------------------------------------------------------------------------------------------------ the pipeline
void open_and_prove(DeviceCtx* ctx, NttTables& tables, const OlaGpuConfig& cfg, const OlaBatch& trace, const OlaBatch& zs,
                    const OlaBatch& quot, uint32_t nperm, OlaChallenger& ch, std::vector<uint8_t>& bytes, size_t& openings_len,
                    PowDefer* pow_defer = nullptr) {
    DevBuf mem(ctx);
    const int degree_bits = (int)trace.log_n;
    const size_t n = trace.n();
    const int rate_bits = (int)cfg.rate_bits;
    const size_t N = n << rate_bits;
    const int W = (int)trace.ncols, Z = (int)zs.ncols, Q = (int)quot.ncols;
    const size_t len_cap = (size_t)1 << cfg.cap_height;
    std::vector<int> arities = fri_arities(cfg, degree_bits);
    {
        int tot = 0;
        for (int a : arities) tot += a;
        if (tot > degree_bits + rate_bits - (int)cfg.cap_height) throw OlaError(OLA_E_INVALID_ARG, "FRI total reduction arity is too large.");
    }

    // ---- zeta and the opening set (prover.rs:499-524) ----
    const Ext2 zeta = challenger_get_ext(ch);
    if (ext_eq(ext_pow(zeta, (u64)1 << degree_bits), ext_make(1, 0))) throw OlaError(OLA_E_ZETA_IN_SUBGROUP, "Opening point is in the subgroup.");
    const u64 g = gl_root_of_unity(degree_bits);
    const Ext2 zeta_next = ext_scalar_mul(zeta, g);
    const Ext2 g_inv = ext_make(gl_inv(g), 0);
    const Ext2 zpts[3] = {zeta, zeta_next, g_inv};
    ExtPow zpow[3], zinv[3];
    {
        const Ext2 six[6] = {zeta, zeta_next, g_inv, ext_inv(zeta), ext_inv(zeta_next), ext_inv(g_inv)};
        ExtPow tabs[6];
        make_ext_pows(mem, six, 6, degree_bits, tabs, 3);
        for (int b = 0; b < 3; b++) { zpow[b] = tabs[b]; zinv[b] = tabs[3 + b]; }
    }
    const ExtPow pz = zpow[0], pzn = zpow[1], pgi = zpow[2];
    std::vector<Ext2> local, next, zs_local, zs_next, q_local, zs_last_all, dummy;
    {
        EvalJob jobs[4];
        const OlaBatch* jb[4] = {&trace, &zs, &quot, &zs};
        const int jp[4] = {2, 2, 1, 1};
        const bool sharded = trace.is_shard() && ctx->shard.world > 1;
        if (!sharded) {
            {
                WorkScope ws(ctx, 3);      // the partition divides the columns among the ranks
                PhaseScope ph(ctx, PH_OPEN_EVAL, (double)n * (2 * W + 3 * Z + Q), (double)n * (W + 2 * Z + Q) * 8);
                eval_batch_launch(mem, trace, 2, pz, pzn, &local, &next, jobs[0]);
                eval_batch_launch(mem, zs, 2, pz, pzn, &zs_local, &zs_next, jobs[1]);
                eval_batch_launch(mem, quot, 1, pz, pz, &q_local, &dummy, jobs[2]);
                eval_batch_launch(mem, zs, 1, pgi, pgi, &zs_last_all, &dummy, jobs[3]);
            }
            if (ctx->acct.shardable) acct_exchange(ctx, (size_t)(2 * W + 3 * Z + Q) * 16);
            mem.sync_collect();
            for (auto& j : jobs) eval_batch_collect(j);
        } else {
            const uint32_t world = ctx->shard.world;
            size_t record = 0;
            for (int i = 0; i < 4; i++) {
                jobs[i].cpr = (jb[i]->ncols + world - 1) / world;
                jobs[i].send_off = record;
                record += (size_t)jp[i] * jobs[i].cpr * 2;
            }
            u64* d_send = mem.alloc(record);
            u64* d_recv = mem.alloc(record * world);
            eval_batch_launch(mem, trace, 2, pz, pzn, &local, &next, jobs[0], d_send);
            eval_batch_launch(mem, zs, 2, pz, pzn, &zs_local, &zs_next, jobs[1], d_send);
            eval_batch_launch(mem, quot, 1, pz, pz, &q_local, &dummy, jobs[2], d_send);
            eval_batch_launch(mem, zs, 1, pgi, pgi, &zs_last_all, &dummy, jobs[3], d_send);
            shard_all_gather(ctx, d_send, d_recv, record * 8);
            std::vector<u64> recv(record * world);
            HIP_CHECK(hipMemcpyAsync(recv.data(), d_recv, recv.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (auto& j : jobs) eval_batch_collect_sharded(j, recv.data(), record);
        }
    }
    std::vector<u64> ctl_last;
    for (int i = (int)nperm; i < Z; i++) ctl_last.push_back(zs_last_all[i].a);

    ByteWriter w{bytes, ctx->hasher == (int)OLA_HASH_BLAKE3};
    w.ext_vec(local); w.ext_vec(next); w.ext_vec(zs_local); w.ext_vec(zs_next);
    w.field_vec(ctl_last.data(), ctl_last.size());
    w.ext_vec(q_local);
    openings_len = bytes.size();

    // observe_openings (fri/challenges.rs:16-23) in to_fri_openings order (proof.rs:235-265)
    for (auto& e : local) challenger_observe_ext(ch, e);
    for (auto& e : zs_local) challenger_observe_ext(ch, e);
    for (auto& e : q_local) challenger_observe_ext(ch, e);
    for (auto& e : next) challenger_observe_ext(ch, e);
    for (auto& e : zs_next) challenger_observe_ext(ch, e);
    for (u64 x : ctl_last) challenger_observe_ext(ch, ext_make(x, 0));

    // ---- prove_openings: final polynomial (fri/oracle.rs:178-219) ----
    const Ext2 alpha = challenger_get_ext(ch);
    const int napow = W + Z + Q;
    HostSpan h_apow = mem.host(2 * (size_t)napow);
    {
        Ext2 acc = ext_make(1, 0);
        for (int i = 0; i < napow; i++) { h_apow[i] = acc.a; h_apow[napow + i] = acc.b; acc = ext_mul(acc, alpha); }
    }
    u64* d_apow = mem.alloc(h_apow.size());
    HIP_CHECK(hipMemcpyAsync(d_apow, h_apow.data(), h_apow.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    u64* S = mem.alloc(6 * n);
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, trace.coeffs, W, zs.coeffs, Z,
                       quot.coeffs, Q, (int)nperm, n, d_apow, napow, S);
    const int l0 = W + Z + Q, l1 = W + Z, l2 = Z - (int)nperm;
    (void)l0;
    const bool use2 = l2 > 0;
    (void)zpts;
    u64* tot = mem.alloc((n + SCAN_B - 1) / SCAN_B + 1);
    for (int b = 0; b < 3; b++) {
        if (b == 2 && !use2) continue;
        hipLaunchKernelGGL(weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, S + 2 * b * n, S + (2 * b + 1) * n,
                           n, zpow[b]);
        scan_plane(mem, S + 2 * b * n, n, tot);
        scan_plane(mem, S + (2 * b + 1) * n, n, tot);
    }
    // weights: ((q0 * alpha^l1) + q1) * alpha^l2 + q2   (oracle.rs:212-213)
    const Ext2 w2 = ext_make(1, 0);
    const Ext2 w1 = use2 ? ext_pow(alpha, (u64)l2) : ext_make(1, 0);
    const Ext2 w0 = ext_mul(ext_pow(alpha, (u64)l1), w1);
    // coefficient planes [fa | fb] of length N (zero padded: PolynomialCoeffs::lde, polynomial/mod.rs:215-217)
    u64* coef = mem.alloc(2 * N);
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, S, n, zinv[0], zinv[1], zinv[2], w0,
                       w1, w2, use2 ? 1 : 0, coef, coef + N, N);

    // ---- FRI commit phase (fri/prover.rs:72-121) ----
    // (OLA_TIMING scopes carry the reference's `timed!` names: fri/oracle.rs:221-225, fri/prover.rs:41-58)
    std::unique_ptr<PhaseTimer> t_fold;
    std::vector<FriLayer> layers;
    u64 shift = GL_GENERATOR;
    size_t len = N;
    u64* cur_coef = coef;  // planes [a | b] each `len`
    size_t nz = n;         // coefficients that can be non-zero (the planes are zero beyond them)
    // Under the coset partition the FIRST layer -- 15/16 of the commit phase's values and leaves -- is divided like a commitment
    // (fri/prover.rs:72-121): its bit-reversed values are the leaf-order LDE of the n coefficients (the planes
    // are zero beyond n), a rank extends and hashes its cosets only (n/arity leaves per coset), builds their sub-trees and the
    // cap slices are all-gathered; the layers after the first fold (1/16 of the data, and shrinking) stay replicated.
    const bool sharded = trace.is_shard() && ctx->shard.world > 1;
    for (size_t li = 0; li < arities.size(); li++) {
        const int ab = arities[li];
        const int arity = 1 << ab;
        int cur_bits = 0;
        while (((size_t)1 << cur_bits) < len) cur_bits++;
        // values = coset_fft(coeffs, shift), kept bit-reversed (reverse_index_bits_in_place, prover.rs:88)
        FriLayer L;
        L.len = len; L.arity_bits = ab;
        const size_t nleaves = len >> ab;
        L.cap.resize(len_cap * 4);
        const bool shard_layer = sharded && li == 0 && degree_bits >= ab;
        if (shard_layer) {
            const uint32_t lw = ctx->shard.log_world;
            const size_t coset_count = ((size_t)1 << rate_bits) >> lw, coset_first = (size_t)ctx->shard.rank * coset_count;
            const size_t len_loc = n * coset_count, nl_loc = len_loc >> ab, cap_loc = len_cap >> lw;
            L.shard_leaves = nl_loc;
            L.va = mem.alloc(2 * len_loc);
            L.vb = L.va + len_loc;
            {
                std::unique_ptr<PhaseTimer> t_fft(new PhaseTimer(ctx, "      perform final FFT " + std::to_string(len) + " (this rank's cosets)"));
                ntt_lde_leaf_order(tables, cur_coef, L.va, degree_bits, rate_bits, 2, coset_first, coset_count, len);
            }
            t_fold.reset(new PhaseTimer(ctx, "      fold codewords in the commitment phase"));
            L.heap = mem.alloc(2 * nl_loc * 4);
            launch_leaf_hash_ext(ctx, L.va, L.vb, arity, nl_loc, L.heap + 4 * nl_loc);
            launch_merkle_build(ctx, L.heap, nl_loc, cfg.cap_height - lw);
            u64* d_cap = mem.alloc(len_cap * 4);
            shard_all_gather(ctx, L.heap + 4 * cap_loc, d_cap, cap_loc * 32);
            HIP_CHECK(hipMemcpyAsync(L.cap.data(), d_cap, len_cap * 32, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            WorkScope ws(ctx, (li == 0 && degree_bits >= ab) ? 3 : 0);
            if (li == 0 && degree_bits >= ab && ctx->acct.shardable) acct_exchange(ctx, len_cap * 32);
            L.va = mem.alloc(2 * len);
            L.vb = L.va + len;
            {
                std::unique_ptr<PhaseTimer> t_fft(li == 0 ? new PhaseTimer(ctx, "      perform final FFT " + std::to_string(len)) : nullptr);
                ntt_coset_evaluate(tables, cur_coef, L.va, nullptr, cur_bits, 2, shift, false);
            }
            if (li == 0) t_fold.reset(new PhaseTimer(ctx, "      fold codewords in the commitment phase"));
            L.heap = mem.alloc(2 * nleaves * 4);
            launch_leaf_hash_ext(ctx, L.va, L.vb, arity, nleaves, L.heap + 4 * nleaves);
            launch_merkle_build(ctx, L.heap, nleaves, cfg.cap_height);
            mem.readback(L.cap.data(), L.heap + 4 * len_cap, len_cap * 32);
        }
        mem.sync_collect();
        challenger_observe_cap(ch, L.cap.data(), L.cap.size() / 4);
        const Ext2 beta = challenger_get_ext(ch);
        const size_t out_len = len >> ab;
        u64* folded = mem.alloc(2 * out_len);
        {
            static const bool fold16_off = getenv("OLA_FOLD16") && !strcmp(getenv("OLA_FOLD16"), "0");
            if (arity == 16 && !fold16_off && nz % 16 == 0 && nz >= 4096) {
                const size_t nz_out = nz / 16;
                PhaseScope ph(ctx, PH_FRI_FOLD, (double)(nz + nz_out) * 16, (double)nz);
                if (nz_out < out_len) {
                    HIP_CHECK(hipMemsetAsync(folded + nz_out, 0, (out_len - nz_out) * 8, ctx->stream));
                    HIP_CHECK(hipMemsetAsync(folded + out_len + nz_out, 0, (out_len - nz_out) * 8, ctx->stream));
                }
                const Ext2 b2 = ext_mul(beta, beta), b4 = ext_mul(b2, b2), b8 = ext_mul(b4, b4);
                hipLaunchKernelGGL(fold16_kernel, dim3((unsigned)((nz_out * 8 + 255) / 256)), dim3(256), 0, ctx->stream, cur_coef, cur_coef + len, nz_out, beta,
                                   b2, b4, b8, folded, folded + out_len);
                nz = nz_out;
            } else {
                PhaseScope ph(ctx, PH_FRI_FOLD, (double)(len + out_len) * 16, (double)len);
                hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((out_len + 255) / 256)), dim3(256), 0, ctx->stream, cur_coef, cur_coef + len, out_len,
                                   arity, beta, folded, folded + out_len);
                nz = std::min(out_len, (nz + arity - 1) / arity);
            }
        }
        cur_coef = folded;
        len = out_len;
        shift = gl_pow(shift, (u64)arity);
        layers.push_back(L);
    }
    // final polynomial: truncate to len / 2^rate_bits (prover.rs:114-119)
    const size_t final_len = len >> rate_bits;
    std::vector<u64> h_final(2 * len);
    mem.readback(h_final.data(), cur_coef, 2 * len * 8);
    mem.sync_collect();
    std::vector<Ext2> final_poly(final_len);
    for (size_t i = 0; i < final_len; i++) final_poly[i] = ext_make(h_final[i], h_final[len + i]);
    for (auto& e : final_poly) challenger_observe_ext(ch, e);

    // ---- proof of work (prover.rs:126-148), minimal witness ----
    u64 hsh[4];
    for (int i = 0; i < 4; i++) hsh[i] = challenger_get(ch);
    t_fold.reset();
    u64 pow_witness = 0;
    bool pow_deferred = false;
    PowDefer::Job pow_job = {};
    {
        PhaseTimer t_pow(ctx, "      find proof-of-work witness");
        // inside a whole proof: on the side stream, the witness is patched in at the end (its place in the bytes is known below)
        pow_deferred = pow_enqueue(ctx, pow_defer, hsh, cfg.proof_of_work_bits, 0);
        if (!pow_deferred) pow_witness = run_pow(ctx, hsh, cfg.proof_of_work_bits);
    }

    // ---- query rounds (prover.rs:150-204) ----
    const int nq = (int)cfg.num_query_rounds;
    std::vector<size_t> xs(nq);
    for (int r = 0; r < nq; r++) xs[r] = (size_t)(challenger_get(ch) % (u64)N);
    const OlaBatch* oracles[3] = {&trace, &zs, &quot};
    const int depth0 = degree_bits + rate_bits - (int)cfg.cap_height;
    std::vector<std::vector<u64>> rows(3), paths(3);
    for (int o = 0; o < 3; o++) {
        rows[o].resize((size_t)nq * oracles[o]->ncols);
        paths[o].resize((size_t)nq * (size_t)std::max(depth0, 1) * 4);
        query_leaves(mem, tables, *oracles[o], xs.data(), nq, depth0, rows[o].data(), paths[o].data());
    }
    // per layer: leaves (arity ext) and paths at x >> (sum of arity bits so far + this)
    std::vector<std::vector<u64>> lrows(layers.size()), lpaths(layers.size());
    std::vector<int> ldepth(layers.size());
    {
        std::vector<size_t> cur = xs;
        std::vector<std::vector<unsigned long long>> h_idx_all(layers.size());   // staging buffers live until the one sync below
        std::vector<u64> shard_recv;          // first layer on the partition: every rank's records [rank][rows nq*arity*2 | paths nq*depth*4]
        size_t shard_record = 0;
        for (size_t li = 0; li < layers.size(); li++) {
            unsigned long long* d_idx = (unsigned long long*)mem.alloc(nq);
            FriLayer& L = layers[li];
            const int arity = 1 << L.arity_bits;
            const size_t nleaves = L.len >> L.arity_bits;
            int lb = 0;
            while (((size_t)1 << lb) < nleaves) lb++;
            ldepth[li] = lb - (int)cfg.cap_height;
            std::vector<unsigned long long>& h_idx = h_idx_all[li];
            h_idx.resize(nq);
            for (int r = 0; r < nq; r++) { cur[r] >>= L.arity_bits; h_idx[r] = cur[r]; }
            lrows[li].resize((size_t)nq * arity * 2);
            if (ldepth[li] > 0) lpaths[li].resize((size_t)nq * ldepth[li] * 4);
            if (L.shard_leaves) {
                // every rank gathers all queries at the local index (meaningful on the owner only); the records are all-gathered
                // and each query is read from its owner's copy below
                for (int r = 0; r < nq; r++) h_idx[r] = cur[r] % L.shard_leaves;
                HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), nq * 8, hipMemcpyHostToDevice, ctx->stream));
                const size_t rows_w = (size_t)nq * arity * 2, paths_w = (size_t)nq * std::max(ldepth[li], 0) * 4;
                shard_record = rows_w + paths_w;
                u64* d_send = mem.alloc(shard_record);
                u64* d_recv = mem.alloc(shard_record * ctx->shard.world);
                hipLaunchKernelGGL(gather_ext_leaves_kernel, dim3((unsigned)nq), dim3(64), 0, ctx->stream, L.va, L.vb, arity, d_idx, d_send);
                if (ldepth[li] > 0)
                    hipLaunchKernelGGL(gather_paths_kernel, dim3((unsigned)nq), dim3(((ldepth[li] * 4 + 63) / 64) * 64), 0, ctx->stream, L.heap,
                                       L.shard_leaves, ldepth[li], d_idx, d_send + rows_w);
                shard_all_gather(ctx, d_send, d_recv, shard_record * 8);
                shard_recv.resize(shard_record * ctx->shard.world);
                HIP_CHECK(hipMemcpyAsync(shard_recv.data(), d_recv, shard_recv.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
                continue;
            }
            if (li == 0 && ctx->acct.shardable && ctx->shard.world <= 1) acct_exchange(ctx, ((size_t)nq * arity * 2 + (size_t)nq * std::max(ldepth[li], 0) * 4) * 8 * 8);
            HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), nq * 8, hipMemcpyHostToDevice, ctx->stream));
            u64* d_rows = mem.alloc((size_t)nq * arity * 2);
            hipLaunchKernelGGL(gather_ext_leaves_kernel, dim3((unsigned)nq), dim3(64), 0, ctx->stream, L.va, L.vb, arity, d_idx, d_rows);
            mem.readback(lrows[li].data(), d_rows, lrows[li].size() * 8);
            if (ldepth[li] > 0) {
                u64* d_paths = mem.alloc((size_t)nq * ldepth[li] * 4);
                hipLaunchKernelGGL(gather_paths_kernel, dim3((unsigned)nq), dim3(((ldepth[li] * 4 + 63) / 64) * 64), 0, ctx->stream, L.heap,
                                   nleaves, ldepth[li], d_idx, d_paths);
                mem.readback(lpaths[li].data(), d_paths, lpaths[li].size() * 8);
            }
        }
        mem.sync_collect();
        if (!shard_recv.empty()) {
            const FriLayer& L = layers[0];
            const int arity = 1 << L.arity_bits, dep = std::max(ldepth[0], 0);
            const size_t rows_w = (size_t)nq * arity * 2;
            for (int r = 0; r < nq; r++) {
                const size_t leaf = xs[r] >> L.arity_bits, owner = leaf / L.shard_leaves;
                const u64* rec = shard_recv.data() + owner * shard_record;
                std::copy(rec + (size_t)r * arity * 2, rec + (size_t)(r + 1) * arity * 2, lrows[0].begin() + (size_t)r * arity * 2);
                if (dep) std::copy(rec + rows_w + (size_t)r * dep * 4, rec + rows_w + (size_t)(r + 1) * dep * 4, lpaths[0].begin() + (size_t)r * dep * 4);
            }
        }
    }

    // ---- serialise the FRI proof (serialization.rs:305-317) ----
    w.u32((uint32_t)layers.size());
    for (auto& L : layers) w.cap(L.cap.data(), len_cap);
    w.u32((uint32_t)nq);
    for (int r = 0; r < nq; r++) {
        w.u32(3);
        for (int o = 0; o < 3; o++) {
            w.field_vec(rows[o].data() + (size_t)r * oracles[o]->ncols, oracles[o]->ncols);
            w.merkle_proof(paths[o].data() + (size_t)r * (size_t)std::max(depth0, 0) * 4, std::max(depth0, 0));
        }
        w.u32((uint32_t)layers.size());
        for (size_t li = 0; li < layers.size(); li++) {
            const int arity = 1 << layers[li].arity_bits;
            w.u32((uint32_t)arity);
            for (int k = 0; k < arity; k++) {
                w.field(lrows[li][((size_t)r * arity + k) * 2]);
                w.field(lrows[li][((size_t)r * arity + k) * 2 + 1]);
            }
            w.merkle_proof(ldepth[li] > 0 ? lpaths[li].data() + (size_t)r * ldepth[li] * 4 : nullptr, ldepth[li] > 0 ? ldepth[li] : 0);
        }
    }
    w.ext_vec(final_poly);
    if (pow_deferred) pow_defer->jobs.back().at = bytes.size();
    w.field(pow_witness);
}

// ------------------------------------------------------------------------------------------------ the same, one step per call
// "ola_open; ola_fri_commit_begin / next_layer(beta, cap_out) / finish(final_poly_out) -- layer-stepped because each
// beta depends on the previous cap through the host challenger (fri/prover.rs:98-101); ola_pow; ola_fri_query".  For a host that
// keeps the reference's own loops (StarkOpeningSet::new, prove_openings, fri_committed_trees, fri_proof_of_work,
// fri_prover_query_rounds) and its own Challenger, and hands only the device work over.  Same kernels and helpers as
// open_and_prove above, whose bytes the steps reassemble to (tests/test_gpu_fri_steps.py); single-device contexts only.
}  // namespace ola
struct OlaFri {      // (global, like OlaBatch: the C header names it)
    ola::DeviceCtx* ctx;
    ola::NttTables* tables;
    OlaGpuConfig cfg;
    const OlaBatch *trace, *zs, *quot;
    uint32_t nperm;
    ola::DevBuf mem;                  // everything that lives between the steps
    int degree_bits, rate_bits, stage = 0;   // 0 opened, 1 polynomial built / layers being committed, 2 finished
    size_t n, N, len, nz;
    ola::Ext2 zeta;
    ola::ExtPow zpow[3], zinv[3];
    std::vector<int> arities;
    std::vector<ola::FriLayer> layers;
    ola::u64* cur_coef = nullptr;
    ola::u64 shift = ola::GL_GENERATOR;
    void* owner = nullptr;            // the OlaCtx the batches belong to (the C wrappers make its device current)
    OlaFri(ola::DeviceCtx* c, ola::NttTables* t, const OlaGpuConfig& g, const OlaBatch* tr, const OlaBatch* z, const OlaBatch* q, uint32_t np)
        : ctx(c), tables(t), cfg(g), trace(tr), zs(z), quot(q), nperm(np), mem(c) {
        mem.use_pinned = false;
        degree_bits = (int)tr->log_n; rate_bits = (int)g.rate_bits;
        n = tr->n(); N = n << rate_bits; len = N; nz = n;
        arities = ola::fri_arities(cfg, degree_bits);
        int tot = 0;
        for (int a : arities) tot += a;
        if (tot > degree_bits + rate_bits - (int)cfg.cap_height) throw ola::OlaError(OLA_E_INVALID_ARG, "FRI total reduction arity is too large.");
    }
};
namespace ola {

// StarkOpeningSet::new (circuits/src/stark/proof.rs:198-233) at the caller's zeta: the opening set in wire format
// (serialization.rs write_stark_opening_set: local, next, permutation_ctl_zs, its next, ctl_zs_last, quotient_polys)
void fri_steps_open(OlaFri& f, const u64 zeta_in[2], std::vector<uint8_t>& bytes) {
    DeviceCtx* ctx = f.ctx;
    const int degree_bits = f.degree_bits;
    f.zeta = ext_make(gl_canon(zeta_in[0]), gl_canon(zeta_in[1]));
    if (ext_eq(ext_pow(f.zeta, (u64)1 << degree_bits), ext_make(1, 0))) throw OlaError(OLA_E_ZETA_IN_SUBGROUP, "Opening point is in the subgroup.");
    const u64 g = gl_root_of_unity(degree_bits);
    const Ext2 zeta_next = ext_scalar_mul(f.zeta, g), g_inv = ext_make(gl_inv(g), 0);
    {
        const Ext2 six[6] = {f.zeta, zeta_next, g_inv, ext_inv(f.zeta), ext_inv(zeta_next), ext_inv(g_inv)};
        ExtPow tabs[6];
        make_ext_pows(f.mem, six, 6, degree_bits, tabs, 3);
        for (int b = 0; b < 3; b++) { f.zpow[b] = tabs[b]; f.zinv[b] = tabs[3 + b]; }
    }
    std::vector<Ext2> local, next, zs_local, zs_next, q_local, zs_last_all, dummy;
    {
        DevBuf tmp(ctx);
        EvalJob jobs[4];
        eval_batch_launch(tmp, *f.trace, 2, f.zpow[0], f.zpow[1], &local, &next, jobs[0]);
        eval_batch_launch(tmp, *f.zs, 2, f.zpow[0], f.zpow[1], &zs_local, &zs_next, jobs[1]);
        eval_batch_launch(tmp, *f.quot, 1, f.zpow[0], f.zpow[0], &q_local, &dummy, jobs[2]);
        eval_batch_launch(tmp, *f.zs, 1, f.zpow[2], f.zpow[2], &zs_last_all, &dummy, jobs[3]);
        tmp.sync_collect();
        for (auto& j : jobs) eval_batch_collect(j);
    }
    std::vector<u64> ctl_last;
    for (int i = (int)f.nperm; i < (int)f.zs->ncols; i++) ctl_last.push_back(zs_last_all[i].a);
    ByteWriter w{bytes, ctx->hasher == (int)OLA_HASH_BLAKE3};
    w.ext_vec(local); w.ext_vec(next); w.ext_vec(zs_local); w.ext_vec(zs_next);
    w.field_vec(ctl_last.data(), ctl_last.size());
    w.ext_vec(q_local);
}

// prove_openings up to the final polynomial (fri/oracle.rs:178-219) with the caller's alpha
void fri_steps_begin(OlaFri& f, const u64 alpha_in[2]) {
    if (f.stage != 0) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_commit_begin: already begun");
    DeviceCtx* ctx = f.ctx;
    const OlaBatch &trace = *f.trace, &zs = *f.zs, &quot = *f.quot;
    const int W = (int)trace.ncols, Z = (int)zs.ncols, Q = (int)quot.ncols;
    const size_t n = f.n, N = f.N;
    const Ext2 alpha = ext_make(gl_canon(alpha_in[0]), gl_canon(alpha_in[1]));
    const int napow = W + Z + Q;
    u64* coef = f.mem.alloc(2 * N);
    {
        DevBuf tmp(ctx);
        HostSpan h_apow = tmp.host(2 * (size_t)napow);
        Ext2 acc = ext_make(1, 0);
        for (int i = 0; i < napow; i++) { h_apow[i] = acc.a; h_apow[napow + i] = acc.b; acc = ext_mul(acc, alpha); }
        u64* d_apow = tmp.alloc(h_apow.size());
        HIP_CHECK(hipMemcpyAsync(d_apow, h_apow.data(), h_apow.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        u64* S = tmp.alloc(6 * n);
        hipLaunchKernelGGL(compose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, trace.coeffs, W, zs.coeffs, Z, quot.coeffs, Q,
                           (int)f.nperm, n, d_apow, napow, S);
        const int l1 = W + Z, l2 = Z - (int)f.nperm;
        const bool use2 = l2 > 0;
        u64* tot = tmp.alloc((n + SCAN_B - 1) / SCAN_B + 1);
        for (int b = 0; b < 3; b++) {
            if (b == 2 && !use2) continue;
            hipLaunchKernelGGL(weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, S + 2 * b * n, S + (2 * b + 1) * n, n, f.zpow[b]);
            scan_plane(tmp, S + 2 * b * n, n, tot);
            scan_plane(tmp, S + (2 * b + 1) * n, n, tot);
        }
        const Ext2 w2 = ext_make(1, 0);
        const Ext2 w1 = use2 ? ext_pow(alpha, (u64)l2) : ext_make(1, 0);
        const Ext2 w0 = ext_mul(ext_pow(alpha, (u64)l1), w1);
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, S, n, f.zinv[0], f.zinv[1], f.zinv[2], w0, w1, w2,
                           use2 ? 1 : 0, coef, coef + N, N);
    }       // (the scope waits for the stream before its scratch goes back to the pool)
    f.cur_coef = coef;
    f.stage = 1;
}

static void fri_steps_fold(OlaFri& f, const u64 beta_in[2]) {
    DeviceCtx* ctx = f.ctx;
    const int ab = f.layers.back().arity_bits, arity = 1 << ab;
    const Ext2 beta = ext_make(gl_canon(beta_in[0]), gl_canon(beta_in[1]));
    const size_t len = f.len, out_len = len >> ab;
    u64* folded = f.mem.alloc(2 * out_len);
    if (arity == 16 && f.nz % 16 == 0 && f.nz >= 4096) {
        const size_t nz_out = f.nz / 16;
        if (nz_out < out_len) {
            HIP_CHECK(hipMemsetAsync(folded + nz_out, 0, (out_len - nz_out) * 8, ctx->stream));
            HIP_CHECK(hipMemsetAsync(folded + out_len + nz_out, 0, (out_len - nz_out) * 8, ctx->stream));
        }
        const Ext2 b2 = ext_mul(beta, beta), b4 = ext_mul(b2, b2), b8 = ext_mul(b4, b4);
        hipLaunchKernelGGL(fold16_kernel, dim3((unsigned)((nz_out * 8 + 255) / 256)), dim3(256), 0, ctx->stream, f.cur_coef, f.cur_coef + len, nz_out, beta, b2, b4, b8,
                           folded, folded + out_len);
        f.nz = nz_out;
    } else {
        hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((out_len + 255) / 256)), dim3(256), 0, ctx->stream, f.cur_coef, f.cur_coef + len, out_len, arity, beta, folded,
                           folded + out_len);
        f.nz = std::min(out_len, (f.nz + arity - 1) / arity);
    }
    f.cur_coef = folded;
    f.len = out_len;
    f.shift = gl_pow(f.shift, (u64)arity);
}

// one turn of fri_committed_trees (fri/prover.rs:72-121): fold by the previous layer's beta (none before the first layer), then
// commit the values of the current polynomial on its coset -> the layer's cap
void fri_steps_next_layer(OlaFri& f, const u64* beta, u64* cap_out) {
    if (f.stage != 1) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_commit_next_layer: call ola_fri_commit_begin first");
    if (f.layers.size() >= f.arities.size()) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_commit_next_layer: every layer of the reduction plan is committed");
    if (f.layers.empty() != (beta == nullptr)) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_commit_next_layer: beta is NULL for the first layer and only there");
    DeviceCtx* ctx = f.ctx;
    if (beta) fri_steps_fold(f, beta);
    const int ab = f.arities[f.layers.size()], arity = 1 << ab;
    const size_t len = f.len, len_cap = (size_t)1 << f.cfg.cap_height, nleaves = len >> ab;
    int cur_bits = 0;
    while (((size_t)1 << cur_bits) < len) cur_bits++;
    FriLayer L;
    L.len = len; L.arity_bits = ab;
    L.cap.resize(len_cap * 4);
    L.va = f.mem.alloc(2 * len);
    L.vb = L.va + len;
    ntt_coset_evaluate(*f.tables, f.cur_coef, L.va, nullptr, cur_bits, 2, f.shift, false);
    L.heap = f.mem.alloc(2 * nleaves * 4);
    launch_leaf_hash_ext(ctx, L.va, L.vb, arity, nleaves, L.heap + 4 * nleaves);
    launch_merkle_build(ctx, L.heap, nleaves, f.cfg.cap_height);
    HIP_CHECK(hipMemcpyAsync(L.cap.data(), L.heap + 4 * len_cap, len_cap * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    std::copy(L.cap.begin(), L.cap.end(), cap_out);
    f.layers.push_back(L);
}

// fold by the last beta and hand out the final polynomial, truncated to len / 2^rate_bits coefficients (prover.rs:114-119)
size_t fri_steps_finish(OlaFri& f, const u64* beta, u64* final_poly_out, size_t cap_elems) {
    if (f.stage != 1 || f.layers.size() != f.arities.size()) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_commit_finish: layers of the reduction plan are missing");
    if (f.layers.empty() != (beta == nullptr)) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_commit_finish: beta is NULL exactly when the plan has no layer");
    if (beta) fri_steps_fold(f, beta);
    const size_t final_len = f.len >> f.rate_bits;
    if (final_len > cap_elems || !final_poly_out) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_commit_finish: output buffer too small");
    std::vector<u64> h(2 * f.len);
    HIP_CHECK(hipMemcpyAsync(h.data(), f.cur_coef, 2 * f.len * 8, hipMemcpyDeviceToHost, f.ctx->stream));
    HIP_CHECK(hipStreamSynchronize(f.ctx->stream));
    for (size_t i = 0; i < final_len; i++) { final_poly_out[2 * i] = h[i]; final_poly_out[2 * i + 1] = h[f.len + i]; }
    f.stage = 2;
    return final_len;
}

// fri_prover_query_rounds (fri/prover.rs:150-204) for the caller's indices: the query round proofs in wire format
// (serialization.rs:305-317, the part between the caps and the final polynomial: count, then per query the three oracles' rows
// and paths and every layer's leaf and path)
void fri_steps_query(OlaFri& f, const u64* x_index, uint32_t nq_in, std::vector<uint8_t>& bytes) {
    if (f.stage != 2) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_query: call ola_fri_commit_finish first");
    DeviceCtx* ctx = f.ctx;
    const int nq = (int)nq_in;
    std::vector<size_t> xs(nq);
    for (int r = 0; r < nq; r++) {
        if (x_index[r] >= (u64)f.N) throw OlaError(OLA_E_INVALID_ARG, "ola_fri_query: index beyond the LDE");
        xs[r] = (size_t)x_index[r];
    }
    const OlaBatch* oracles[3] = {f.trace, f.zs, f.quot};
    const int depth0 = f.degree_bits + f.rate_bits - (int)f.cfg.cap_height;
    std::vector<std::vector<u64>> rows(3), paths(3), lrows(f.layers.size()), lpaths(f.layers.size());
    std::vector<int> ldepth(f.layers.size());
    {
        DevBuf tmp(ctx);
        for (int o = 0; o < 3; o++) {
            rows[o].resize((size_t)nq * oracles[o]->ncols);
            paths[o].resize((size_t)nq * (size_t)std::max(depth0, 1) * 4);
            query_leaves(tmp, *f.tables, *oracles[o], xs.data(), nq, depth0, rows[o].data(), paths[o].data());
        }
        std::vector<size_t> cur = xs;
        std::vector<std::vector<unsigned long long>> h_idx_all(f.layers.size());
        for (size_t li = 0; li < f.layers.size(); li++) {
            FriLayer& L = f.layers[li];
            const int arity = 1 << L.arity_bits;
            const size_t nleaves = L.len >> L.arity_bits;
            int lb = 0;
            while (((size_t)1 << lb) < nleaves) lb++;
            ldepth[li] = lb - (int)f.cfg.cap_height;
            std::vector<unsigned long long>& h_idx = h_idx_all[li];
            h_idx.resize(nq);
            for (int r = 0; r < nq; r++) { cur[r] >>= L.arity_bits; h_idx[r] = cur[r]; }
            lrows[li].resize((size_t)nq * arity * 2);
            if (ldepth[li] > 0) lpaths[li].resize((size_t)nq * ldepth[li] * 4);
            unsigned long long* d_idx = (unsigned long long*)tmp.alloc(nq);
            HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), nq * 8, hipMemcpyHostToDevice, ctx->stream));
            u64* d_rows = tmp.alloc((size_t)nq * arity * 2);
            hipLaunchKernelGGL(gather_ext_leaves_kernel, dim3((unsigned)nq), dim3(64), 0, ctx->stream, L.va, L.vb, arity, d_idx, d_rows);
            tmp.readback(lrows[li].data(), d_rows, lrows[li].size() * 8);
            if (ldepth[li] > 0) {
                u64* d_paths = tmp.alloc((size_t)nq * ldepth[li] * 4);
                hipLaunchKernelGGL(gather_paths_kernel, dim3((unsigned)nq), dim3(((ldepth[li] * 4 + 63) / 64) * 64), 0, ctx->stream, L.heap, nleaves, ldepth[li], d_idx,
                                   d_paths);
                tmp.readback(lpaths[li].data(), d_paths, lpaths[li].size() * 8);
            }
        }
        tmp.sync_collect();
    }
    ByteWriter w{bytes, ctx->hasher == (int)OLA_HASH_BLAKE3};
    w.u32((uint32_t)nq);
    for (int r = 0; r < nq; r++) {
        w.u32(3);
        for (int o = 0; o < 3; o++) {
            w.field_vec(rows[o].data() + (size_t)r * oracles[o]->ncols, oracles[o]->ncols);
            w.merkle_proof(paths[o].data() + (size_t)r * (size_t)std::max(depth0, 0) * 4, std::max(depth0, 0));
        }
        w.u32((uint32_t)f.layers.size());
        for (size_t li = 0; li < f.layers.size(); li++) {
            const int arity = 1 << f.layers[li].arity_bits;
            w.u32((uint32_t)arity);
            for (int k = 0; k < arity; k++) {
                w.field(lrows[li][((size_t)r * arity + k) * 2]);
                w.field(lrows[li][((size_t)r * arity + k) * 2 + 1]);
            }
            w.merkle_proof(ldepth[li] > 0 ? lpaths[li].data() + (size_t)r * ldepth[li] * 4 : nullptr, ldepth[li] > 0 ? ldepth[li] : 0);
        }
    }
}

}  // namespace ola
