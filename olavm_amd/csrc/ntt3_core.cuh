// Third-generation NTT pass (2^18-point transforms and larger): index maps and per-thread phases, host + gfx950 device.
//
// Contract = plonky2/field/src/cfft/mod.rs:22-231 (fft / ifft / coset variants), computed as an in-place decimation-in-
// frequency transform cut into PASSES over HBM (each reads and writes every element once) and, inside a pass, ROUNDS over
// registers separated by an exchange through LDS:
//   * a workgroup of 256 threads owns a TILE of 2^13 elements, 32 per thread, held in T-form (tform.cuh): a round is a
//     radix-32 (or smaller) transform whose butterflies are plain 32-bit adds and whose twiddles are limb rotations;
//   * tile index t (13 bits) = R transform ("digit") bits d and 13-R batch bits s.  STRIDED passes put d on top of s:
//     the tile is 2^R rows of 2^(13-R) >= 16 consecutive elements; the closing pass of a bit-reversed-order transform is one
//     contiguous run of 2^13; the closing pass of a natural-order transform takes R = 9 with the 16 batch rows 2^(L-4)
//     apart, so that its stores are 16 consecutive natural-order outputs;
//   * round r handles k_r digit bits (top first); the 5 - k_r spare register bits are batch bits (first round) or digits
//     already done (later rounds), so the multiplier after a round depends on the thread and a COMPILE-TIME register index;
//   * between rounds elements move as two 8-byte halves (limbs 0,1 then 2,3) through one 64 KB LDS buffer whose slot map is
//     an XOR swizzle, conflict-free for every round's lane set (checked by tests/host_ntt3_check.cpp);
//   * multipliers: inside a pass, w_{2^R}^(m * q) from a 2^R-entry table; after a strided pass, w_{2^(lo+R)}^(M * Q) (times
//     the inverse transform's 2^-L and the coset factor s^M) from a table laid out like the data itself -- the element's own
//     in-place offset is the index, so the loads are the same 128-byte segments as the stores and the workgroups that share a
//     tile (one per column) find the 64 KB slice in L2; a coset transform multiplies the loaded element by s^(d * 2^lo) from
//     a 2^R-entry table per coset.
// Every function below is per-thread code between two barriers; ntt3.hip strings them together with __syncthreads(), the
// host test with loops over the 256 threads.
#pragma once
#include <type_traits>

#include "gl.cuh"
#include "tform.cuh"

namespace ola {

enum { N3_STRIDED = 0, N3_LAST_BITREV = 1, N3_LAST_NATURAL = 2 };
static const int N3_TILE_BITS = 13;
static const int N3_THREADS = 256;
static const int N3_REGS = 32;

GL_HD constexpr int n3_bitrev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

template <int R, int MODE>
struct N3Cfg {
    static_assert(R >= 5 && R <= 13, "pass width");
    static_assert(MODE != N3_LAST_BITREV || R == 13, "the contiguous closing pass owns the whole tile");
    static_assert(MODE != N3_LAST_NATURAL || R == 9, "the natural-order closing pass is 2^9 x 16");
    static constexpr int SB = N3_TILE_BITS - R;
    static constexpr int NR = (R + 4) / 5;
    GL_HD static constexpr int k(int r) {   // digit bits of round r: as even as possible, larger first
        int total = R, kk = 0;
        for (int i = 0; i <= r; i++) { kk = (total + (NR - i) - 1) / (NR - i); total -= kk; }
        return kk;
    }
    GL_HD static constexpr int g(int r) {   // lowest digit of round r
        int total = R;
        for (int i = 0; i <= r; i++) total -= k(i);
        return total;
    }
    GL_HD static constexpr int dpos(int i) { return MODE == N3_LAST_NATURAL ? (i < 4 ? i : i + 4) : SB + i; }
    GL_HD static constexpr int spos(int i) { return MODE == N3_LAST_NATURAL ? 4 + i : i; }
    GL_HD static constexpr int fill0() { return 5 - k(0); }
    // t-bit position of register bit b in round r
    GL_HD static constexpr int regpos(int r, int b) {
        if (r == 0) return b < fill0() ? spos(SB - fill0() + b) : dpos(g(0) + b - fill0());
        return dpos(g(r) + b);
    }
    GL_HD static constexpr bool is_reg(int r, int pos) {
        for (int b = 0; b < 5; b++) if (regpos(r, b) == pos) return true;
        return false;
    }
    GL_HD static constexpr int tidpos(int r, int b) {
        int cnt = 0;
        for (int pos = 0; pos < N3_TILE_BITS; pos++) {
            if (is_reg(r, pos)) continue;
            if (cnt == b) return pos;
            cnt++;
        }
        return -1;
    }
    // register index -> its bits of t
    GL_HD static constexpr int reg_t(int r, int j) {
        int t = 0;
        for (int b = 0; b < 5; b++) t |= ((j >> b) & 1) << regpos(r, b);
        return t;
    }
    GL_HD static int tid_t(int r, int tid) {
        int t = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) t |= ((tid >> b) & 1) << tidpos(r, b);
        return t;
    }
    // digit / batch value of (part of) a tile index
    GL_HD static constexpr int d_of(int t) {
        int d = 0;
        for (int i = 0; i < R; i++) d |= ((t >> dpos(i)) & 1) << i;
        return d;
    }
    GL_HD static constexpr int s_of(int t) {
        int s = 0;
        for (int i = 0; i < SB; i++) s |= ((t >> spos(i)) & 1) << i;
        return s;
    }
    // natural output index of register j's round-r transform (the in-place DIF leaves X[bitrev(p)] at position p)
    GL_HD static constexpr int qhat(int r, int j) {
        const int p = (r == 0) ? (j >> fill0()) : (j & ((1 << k(r)) - 1));
        return n3_bitrev(p, k(r));
    }
};

// LDS slot of tile element t (8-byte slots): GF(2)-linear, so slot(a | b) = slot(a) ^ slot(b) for disjoint a, b
// (the low five slot bits select the pair of LDS banks of an 8-byte access: bit b >= 5 of t flips them by N3_SWZ[b - 5], chosen
// by search so that the five lowest lane bits of every round of every pass shape hit 32 different bank pairs)
GL_HD constexpr int n3_slot(int t) {
    constexpr int SWZ[8] = {18, 23, 30, 14, 24, 7, 31, 6};
    int s = t;
    for (int b = 5; b < N3_TILE_BITS; b++) s ^= ((t >> b) & 1) ? SWZ[b - 5] : 0;
    return s;
}

struct N3Params {
    const u64* in;
    u64* out;
    size_t in_col_stride, out_col_stride, in_coset_stride, out_coset_stride;
    int log_n;          // L
    int lo;             // the pass transforms index bits [lo, lo + R)
    size_t ncols;
    const u64* tw;      // w_{2^R}^e, e < 2^R (inverse root for an inverse transform)
    // strided passes: the pass multiplier of the element at in-place offset o (within its 2^(lo+R) block):
    //   ptw[o] = w_{2^(lo+R)}^(M * bitrev_R(d)) * scale * s^M,   o = d * 2^lo + M
    // (scale = 2^-L on the last strided pass of an inverse transform, s = the coset shift of a coset transform)
    const u64* ptw;
    size_t ptw_coset_stride;
    // coset transforms (first pass only): s^(d * 2^lo), [coset][2^R]
    const u64* sc_dig;
};

// scheduling fence for the device compiler: keeps the table loads of one group of registers from being hoisted above the
// arithmetic of the previous groups (32 loads in flight would cost 64 more VGPRs than the kernel has)
#if defined(__HIP_DEVICE_COMPILE__)
#define N3_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define N3_SCHED_FENCE() ((void)0)
#endif
static const int N3_GROUP = 8;   // registers per group of table loads

// Element offset (within a column / coset slice) of tile element t = uniform part + lane part: offsets are linear in the
// bits of t, a thread's registers differ in bits whose contribution is the same for the whole workgroup (scalar registers),
// and the thread's own bits contribute a 32-bit byte offset -- every access is "SGPR base + 32-bit lane offset".
template <int R, int MODE>
struct N3Addr {
    typedef N3Cfg<R, MODE> C;
    size_t tile_base;   // STRIDED / BITREV reads and writes, NATURAL reads
    size_t nat_base;    // NATURAL writes
    int L, lo;
    GL_HD void init(int L_, int lo_, u32 tile) {
        L = L_; lo = lo_;
        nat_base = 0;
        if (MODE == N3_STRIDED) {
            const int mb = lo - C::SB;   // bits of `mid`
            const size_t mid = tile & (((size_t)1 << mb) - 1), hi = tile >> mb;
            tile_base = (hi << (lo + R)) + (mid << C::SB);
        } else if (MODE == N3_LAST_BITREV) {
            tile_base = (size_t)tile << N3_TILE_BITS;
        } else {
            tile_base = (size_t)tile << 9;
            u32 rv = 0;
            for (int i = 0; i < L - 13; i++) rv |= ((tile >> i) & 1u) << (L - 14 - i);
            nat_base = (size_t)rv << 4;
        }
    }
    // linear parts (no tile base)
    GL_HD size_t rd_lin(int t) const {
        if (MODE == N3_STRIDED) return ((size_t)C::d_of(t) << lo) + (size_t)C::s_of(t);
        if (MODE == N3_LAST_BITREV) return (size_t)t;
        return ((size_t)C::s_of(t) << (L - 4)) + (size_t)C::d_of(t);
    }
    GL_HD size_t wr_lin(int t) const {
        if (MODE != N3_LAST_NATURAL) return rd_lin(t);
        return ((size_t)n3_bitrev(C::d_of(t), 9) << (L - 9)) + (size_t)n3_bitrev(C::s_of(t), 4);
    }
    GL_HD size_t wr_base() const { return MODE == N3_LAST_NATURAL ? nat_base : tile_base; }
};
template <class P>
GL_HD P* n3_at(P* uniform_base, u32 lane_elems) {   // uniform pointer + 32-bit lane offset (in elements of 8 bytes)
    typedef typename std::conditional<std::is_const<P>::value, const char, char>::type B;
    return reinterpret_cast<P*>(reinterpret_cast<B*>(uniform_base) + (u32)(lane_elems << 3));
}

// ---------------------------------------------------------------------------------------------------- per-thread phases
template <int R, int MODE, class I, int N3G = N3_GROUP>
GL_HD void n3_load(const N3Params& p, const N3Addr<R, MODE>& a, const u64* in, int tid, u32 coset, T4<I> (&x)[N3_REGS]) {
    typedef N3Cfg<R, MODE> C;
    const int tt = C::tid_t(0, tid);
    const u32 lane = (u32)a.rd_lin(tt);
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) x[j] = tf_from_u64<I>(*n3_at(in + a.tile_base + a.rd_lin(C::reg_t(0, j)), lane));
    if (MODE == N3_STRIDED && p.sc_dig) {
        const u64* dig = p.sc_dig + ((size_t)coset << R);
        const u32 dl = (u32)C::d_of(tt);
        u64 w[2][N3G];
        auto fetch = [&](int g, u64 (&dst)[N3G]) {
#pragma unroll
            for (int j = 0; j < N3G; j++) dst[j] = *n3_at(dig + C::d_of(C::reg_t(0, g + j)), dl);
        };
        fetch(0, w[0]);
#pragma unroll
        for (int g = 0; g < N3_REGS; g += N3G) {
            const int cur = (g / N3G) & 1;
            if (g + N3G < N3_REGS) fetch(g + N3G, w[cur ^ 1]);
            N3_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < N3G; j++) x[g + j] = tf_mul(x[g + j], tf_split_u64(w[cur][j]));
            N3_SCHED_FENCE();
        }
    }
}

template <int R, int MODE, bool INV, int RND, class I, int N3G = N3_GROUP>
GL_HD void n3_round(const N3Params& p, const N3Addr<R, MODE>& a, int tid, u32 coset, T4<I> (&x)[N3_REGS]) {
    typedef N3Cfg<R, MODE> C;
    constexpr int K = C::k(RND);
    if (RND == 0) {
        constexpr int F = 1 << C::fill0();
#pragma unroll
        for (int f = 0; f < F; f++) tf_dft<K, INV, F>(&x[f]);
    } else {
#pragma unroll
        for (int grp = 0; grp < (N3_REGS >> K); grp++) tf_dft<K, INV, 1>(&x[grp << K]);
    }
    if (RND < C::NR - 1) {
        // inside the pass: w_{2^(g+K)}^(m * qhat) = tw[m * qhat << (R - g - K)], m = the digits below this round's
        const int tt = C::tid_t(RND, tid);
        const int m = C::d_of(tt) & ((1 << C::g(RND)) - 1);
        // table loads run one group of registers ahead of the multiplications that use them
        u64 w[2][N3G];
        auto fetch = [&](int g, u64 (&dst)[N3G]) {
#pragma unroll
            for (int j = 0; j < N3G; j++) {
                const int q = C::qhat(RND, g + j);
                dst[j] = q ? *n3_at(p.tw, (u32)(m * q) << (R - C::g(RND) - K)) : 1;
            }
        };
        fetch(0, w[0]);
#pragma unroll
        for (int g = 0; g < N3_REGS; g += N3G) {
            const int cur = (g / N3G) & 1;
            if (g + N3G < N3_REGS) fetch(g + N3G, w[cur ^ 1]);
            N3_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < N3G; j++) {
                if (C::qhat(RND, g + j) == 0) x[g + j] = tf_norm(x[g + j]);   // multiplier 1: only bring the limbs back below 2^25
                else x[g + j] = tf_mul(x[g + j], tf_split_u64(w[cur][j]));
            }
            N3_SCHED_FENCE();
        }
    } else if (MODE == N3_STRIDED) {
        // pass multiplier, read at the element's own in-place offset
        const size_t mask = ((size_t)1 << (p.lo + R)) - 1;
        const u64* ptw = p.ptw + coset * p.ptw_coset_stride + (a.tile_base & mask);
        const u32 lane = (u32)a.wr_lin(C::tid_t(RND, tid));
        u64 w[2][N3G];
        auto fetch = [&](int g, u64 (&dst)[N3G]) {
#pragma unroll
            for (int j = 0; j < N3G; j++) dst[j] = *n3_at(ptw + a.wr_lin(C::reg_t(RND, g + j)), lane);
        };
        fetch(0, w[0]);
#pragma unroll
        for (int g = 0; g < N3_REGS; g += N3G) {
            const int cur = (g / N3G) & 1;
            if (g + N3G < N3_REGS) fetch(g + N3G, w[cur ^ 1]);
            N3_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < N3G; j++) x[g + j] = tf_mul(x[g + j], tf_split_u64(w[cur][j]));
            N3_SCHED_FENCE();
        }
    }
}

// exchange between round RND and RND + 1, half H (0: limbs 0,1; 1: limbs 2,3)
template <int R, int MODE, int RND, int H, class I>
GL_HD void n3_xchg_write(int tid, const T4<I> (&x)[N3_REGS], u64* lds) {
    typedef N3Cfg<R, MODE> C;
    typedef TfTraits<I> Tr;
    const int sl = n3_slot(C::tid_t(RND, tid));
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) {
        const u32 a = Tr::to_u32_biased(x[j].v[2 * H], 0u), b = Tr::to_u32_biased(x[j].v[2 * H + 1], 0u);
        lds[sl ^ n3_slot(C::reg_t(RND, j))] = (u64)a | ((u64)b << 32);
    }
}
template <int R, int MODE, int RND, int H, class I>
GL_HD void n3_xchg_read(int tid, T4<I> (&x)[N3_REGS], const u64* lds) {
    typedef N3Cfg<R, MODE> C;
    typedef TfTraits<I> Tr;
    const int sl = n3_slot(C::tid_t(RND + 1, tid));
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) {
        const u64 v = lds[sl ^ n3_slot(C::reg_t(RND + 1, j))];
        x[j].v[2 * H] = Tr::from_u32((u32)v);
        x[j].v[2 * H + 1] = Tr::from_u32((u32)(v >> 32));
    }
}

// the same exchange one limb (4 bytes) at a time: a 32 KB buffer, so that three workgroups fit a CU
template <int R, int MODE, int RND, int H, class I>
GL_HD void n3_xchg4_write(int tid, const T4<I> (&x)[N3_REGS], u32* lds) {
    typedef N3Cfg<R, MODE> C;
    typedef TfTraits<I> Tr;
    const int sl = n3_slot(C::tid_t(RND, tid));
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) lds[sl ^ n3_slot(C::reg_t(RND, j))] = Tr::to_u32_biased(x[j].v[H], 0u);
}
template <int R, int MODE, int RND, int H, class I>
GL_HD void n3_xchg4_read(int tid, T4<I> (&x)[N3_REGS], const u32* lds) {
    typedef N3Cfg<R, MODE> C;
    typedef TfTraits<I> Tr;
    const int sl = n3_slot(C::tid_t(RND + 1, tid));
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) x[j].v[H] = Tr::from_u32(lds[sl ^ n3_slot(C::reg_t(RND + 1, j))]);
}
// ... and the closing transpose of the contiguous pass in two 4-byte halves (HALF = 0: low words)
template <int R, int MODE, int HALF>
GL_HD void n3_final4_write(int tid, const u64 (&c)[N3_REGS], u32* lds) {
    typedef N3Cfg<R, MODE> C;
    constexpr int last = C::NR - 1;
    const int sl = n3_slot(C::tid_t(last, tid));
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) lds[sl ^ n3_slot(C::reg_t(last, j))] = (u32)(c[j] >> (32 * HALF));
}
template <int HALF>
GL_HD void n3_final4_read(int tid, u32 (&w)[N3_REGS], const u32* lds) {
#pragma unroll
    for (int jj = 0; jj < N3_REGS; jj++) w[jj] = lds[n3_slot(tid) ^ n3_slot(jj << 8)];
}

// strided and natural-order passes store straight from the last round's registers (lanes run along the batch bits)
template <int R, int MODE, class I>
GL_HD void n3_store_direct(const N3Addr<R, MODE>& a, u64* out, int tid, const T4<I> (&x)[N3_REGS]) {
    typedef N3Cfg<R, MODE> C;
    constexpr int last = C::NR - 1;
    const u32 lane = (u32)a.wr_lin(C::tid_t(last, tid));
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) *n3_at(out + a.wr_base() + a.wr_lin(C::reg_t(last, j)), lane) = tf_to_u64(x[j]);
}
// the contiguous closing pass transposes once more through LDS (canonical words) so that lanes store consecutive addresses
template <int R, int MODE, class I>
GL_HD void n3_final_write(int tid, const T4<I> (&x)[N3_REGS], u64* lds) {
    typedef N3Cfg<R, MODE> C;
    constexpr int last = C::NR - 1;
    const int sl = n3_slot(C::tid_t(last, tid));
#pragma unroll
    for (int j = 0; j < N3_REGS; j++) lds[sl ^ n3_slot(C::reg_t(last, j))] = tf_to_u64(x[j]);
}
template <int R, int MODE>
GL_HD void n3_final_store(const N3Addr<R, MODE>& a, u64* out, int tid, const u64* lds) {
#pragma unroll
    for (int jj = 0; jj < N3_REGS; jj++) {
        *n3_at(out + a.tile_base + (jj << 8), (u32)tid) = lds[n3_slot(tid) ^ n3_slot(jj << 8)];
    }
}

}  // namespace ola
