// Batched Goldilocks NTT / iNTT / coset-LDE for gfx950 (MI355X).
//
// Replaces, for whole column batches resident in HBM, the reference's per-column CPU transforms
//   plonky2/field/src/cfft/mod.rs:22-231   evaluate_poly / evaluate_poly_with_offset / interpolate_poly(_with_offset)
//   plonky2/field/src/cfft/serial.rs:9-78  (index conventions: natural in, natural out; LDE point m = offset*g^m)
// and the dead CUDA hop plonky2/field/src/cfft/ntt/mod.rs:123-388.
//
// Design (not a translation of the reference's recursion):
//   * a length-2^L transform is cut into "passes" of R bits each (top bits first, decimation in frequency);
//     one workgroup owns a tile of 2^R x T elements in LDS, T = 16 consecutive elements (128 B) per strided row so
//     that every global access is a full 128-byte line; the last pass is contiguous (or, for natural-order output,
//     reads 16 rows and writes 16-wide columns);
//   * inside a tile the R bits are done in rounds of <= 4 bits held in registers (16 values per thread), with one LDS
//     exchange per round; LDS rows are padded by one element per 16 to stay bank-conflict free;
//   * each pass is a plain 2^R-point transform followed by one twiddle multiplication per element
//     (w_{2^(lo+R)}^(low*q), from a two-level table), so passes compose exactly into the big transform;
//   * in-place decimation in frequency leaves the result bit-reversed, which is precisely the order the
//     commitment's Merkle leaves want, so the LDE never needs a bit-reversal pass.
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

#include "device_ctx.h"
#include "gl.cuh"

namespace ola {

enum { MODE_STRIDED = 0, MODE_CONTIG = 1, MODE_ROWS = 2 };

struct NttPassParams {
    const u64* in;
    u64* out;
    size_t in_col_stride, out_col_stride;  // elements between consecutive columns
    size_t in_coset_stride;                // elements between consecutive cosets on the input side (0: shared input)
    size_t out_coset_stride;               // elements between consecutive cosets of one column (LDE), else 0
    int log_n;                             // transform length 2^log_n
    int lo;                                // this pass handles index bits [lo, lo+R)
    int natural_out;                       // MODE_CONTIG: write natural order instead of bit-reversed
    const u64* tw_small;                   // w_{2^R}^j, j < 2^(R-1)
    const u64* tw_lo;                      // two-level table of w_{2^(lo+R)}: w^j (j < 2^tw_h) and w^(j<<tw_h)
    const u64* tw_hi;
    int tw_h;
    const u64* sc_lo;                      // optional pre-scale tables s^j / s^(j<<sc_h), per coset
    const u64* sc_hi;
    int sc_h;
    size_t sc_coset_stride;
    u64 out_scale;                         // multiply every output by this (1 = skip)
};

template <int K, int S, int R>
__device__ __forceinline__ void dif_radix(u64 (&x)[1 << K], u32 m_below, const u64* __restrict__ tw) {
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        const int b = S + i;
#pragma unroll
        for (int j = 0; j < (1 << K); ++j) {
            if (j & (1 << i)) continue;
            const u32 e = ((u32)(j & ((1 << i) - 1)) << S) + m_below;  // m mod 2^b
            const u64 a = x[j], c = x[j | (1 << i)];
            x[j] = gl_add(a, c);
            u64 d = gl_sub(a, c);
            if (e != 0) d = gl_mul(d, tw[(size_t)e << (R - 1 - b)]);
            x[j | (1 << i)] = d;
        }
    }
}

__device__ __forceinline__ int lds_idx(int e) { return e + (e >> 4); }

// one round: bits [S, S+K) of the tile's m index, 2^K values per group in registers
template <int R, int LOGT, int K, int S>
__device__ __forceinline__ void tile_round(u64* lds, const u64* __restrict__ tw, int tid, int nthreads) {
    constexpr int E = 1 << (R + LOGT);
    constexpr int GROUPS = E >> K;
    // group id g enumerates the "other" bits of the LDS element index e = m*T + u with bits [SP, SP+K) removed
    constexpr int SP = S + LOGT;
    for (int g = tid; g < GROUPS; g += nthreads) {
        const int low = g & ((1 << SP) - 1);
        const int high = g >> SP;
        const int e0 = (high << (SP + K)) | low;
        const u32 m_below = (u32)(low >> LOGT);
        u64 x[1 << K];
#pragma unroll
        for (int j = 0; j < (1 << K); ++j) x[j] = lds[lds_idx(e0 + (j << SP))];
        dif_radix<K, S, R>(x, m_below, tw);
#pragma unroll
        for (int j = 0; j < (1 << K); ++j) lds[lds_idx(e0 + (j << SP))] = x[j];
    }
}

// rounds of as-even-as-possible width, top bits first; S = bits still to do, NR = rounds left
template <int R, int LOGT, int S, int NR>
struct TileRounds {
    static __device__ __forceinline__ void run(u64* lds, const u64* __restrict__ tw, int tid, int nthreads) {
        constexpr int K = (S + NR - 1) / NR;
        tile_round<R, LOGT, K, S - K>(lds, tw, tid, nthreads);
        __syncthreads();
        TileRounds<R, LOGT, S - K, NR - 1>::run(lds, tw, tid, nthreads);
    }
};
template <int R, int LOGT, int S>
struct TileRounds<R, LOGT, S, 0> {
    static __device__ __forceinline__ void run(u64*, const u64* __restrict__, int, int) {}
};

template <int R, int LOGT>
__device__ __forceinline__ void tile_ntt(u64* lds, const u64* __restrict__ tw, int tid, int nthreads) {
    TileRounds<R, LOGT, R, (R + 3) / 4>::run(lds, tw, tid, nthreads);
}

template <int R, int LOGT>
constexpr int ntt_threads() {
    return ((1 << (R + LOGT)) / 16) < 64 ? 64 : (((1 << (R + LOGT)) / 16) > 1024 ? 1024 : ((1 << (R + LOGT)) / 16));
}

template <int R, int LOGT, int MODE>
__global__ __launch_bounds__((ntt_threads<R, LOGT>())) void ntt_pass_kernel(NttPassParams p) {
    constexpr int E = 1 << (R + LOGT);
    constexpr int T = 1 << LOGT;
    constexpr int NT = ntt_threads<R, LOGT>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* lds = reinterpret_cast<u64*>(smem_raw);
    const int tid = threadIdx.x;
    const size_t tile = blockIdx.x;
    const size_t col = blockIdx.y;
    const size_t coset = blockIdx.z;
    const u64* in = p.in + col * p.in_col_stride + coset * p.in_coset_stride;
    u64* out = p.out + col * p.out_col_stride + coset * p.out_coset_stride;
    const int L = p.log_n;

    // ---- tile geometry ----
    size_t base = 0;      // STRIDED / CONTIG: index of (m=0,u=0)
    size_t low0 = 0;      // STRIDED: first "low" index of the tile
    size_t rowsA = 0;     // ROWS: tile id
    if (MODE == MODE_STRIDED) {
        const size_t lowblks = ((size_t)1 << p.lo) >> LOGT;
        const size_t lb = tile % lowblks, hi = tile / lowblks;
        low0 = lb << LOGT;
        base = (hi << (p.lo + R)) + low0;
    } else if (MODE == MODE_CONTIG) {
        base = tile << R;
    } else {
        rowsA = tile;
    }

    // ---- copy in (with optional pre-scale by s^index) ----
    const u64* sc_lo = p.sc_lo ? p.sc_lo + coset * p.sc_coset_stride : nullptr;
    const u64* sc_hi = p.sc_hi ? p.sc_hi + coset * p.sc_coset_stride : nullptr;
    for (int e = tid; e < E; e += NT) {
        int m, u;
        size_t g;
        if (MODE == MODE_STRIDED) {
            u = e & (T - 1); m = e >> LOGT;
            g = base + ((size_t)m << p.lo) + u;
        } else if (MODE == MODE_CONTIG) {
            u = 0; m = e;
            g = base + m;
        } else {
            m = e & ((1 << R) - 1); u = e >> R;
            const int ub = L - R - LOGT;  // bits of the tile id
            const size_t row = ((size_t)bitrev32((u32)u, LOGT) << ub) + (ub ? bitrev32((u32)rowsA, ub) : 0);
            g = (row << R) + m;
        }
        u64 v = in[g];
        if (sc_lo) {
            const u64 s = gl_mul(sc_lo[g & (((size_t)1 << p.sc_h) - 1)], sc_hi[g >> p.sc_h]);
            v = gl_mul(gl_canon(v), s);
        } else {
            v = gl_canon(v);
        }
        lds[lds_idx((m << LOGT) + u)] = v;
    }
    __syncthreads();

    tile_ntt<R, LOGT>(lds, p.tw_small, tid, NT);

    // ---- copy out (with inter-pass twiddle / scaling) ----
    for (int e = tid; e < E; e += NT) {
        u64 v;
        size_t g;
        if (MODE == MODE_STRIDED) {
            const int u = e & (T - 1), mp = e >> LOGT;
            v = lds[lds_idx(e)];
            const u64 q = bitrev32((u32)mp, R);
            const u64 ex = (low0 + u) * q;  // < 2^(lo+R)
            if (ex) {
                const u64 w = gl_mul(p.tw_lo[ex & (((u64)1 << p.tw_h) - 1)], p.tw_hi[ex >> p.tw_h]);
                v = gl_mul(v, w);
            }
            g = base + ((size_t)mp << p.lo) + u;
        } else if (MODE == MODE_CONTIG) {
            v = lds[lds_idx(p.natural_out ? (int)bitrev32((u32)e, R) : e)];
            g = base + e;
        } else {
            const int u = e & (T - 1), q = e >> LOGT;
            v = lds[lds_idx(((int)bitrev32((u32)q, R) << LOGT) + u)];
            g = (rowsA << LOGT) + u + ((size_t)q << (L - R));
        }
        if (p.out_scale != 1) v = gl_mul(v, p.out_scale);
        out[g] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side: tables, planner, launcher
// ------------------------------------------------------------------------------------------------
struct TwoLevel {
    u64* lo = nullptr;
    u64* hi = nullptr;
    int h = 0;
};

struct NttTables {
    DeviceCtx* ctx;
    std::map<std::pair<int, int>, u64*> small;       // (R, inverse) -> w_{2^R}^j, j < 2^(R-1)
    std::map<std::pair<int, int>, TwoLevel> two;     // (k, inverse) -> two-level powers of w_{2^k}
    std::map<std::pair<int, int>, TwoLevel> coset;   // (log_n, rate_bits) -> per-coset scale tables (blowup cosets)
    std::map<std::pair<int, u64>, TwoLevel> shift;   // (log_n, shift) -> s^k tables (single coset, arbitrary shift)
    std::map<std::tuple<int, int, int, u64>, const u64*> coset_steps;  // (log_n, rate_bits, e, shift) -> per-coset s^(2^e) (ntt2.hip)
    // (log_n, rate_bits, lo/R/inverse, shift) -> per-coset pre-scale tables of a coset transform's first pass (ntt2.hip)
    // ola_gpu_ntt_pass_times: every T-form pass launch bracketed by two events on the context's stream, summed per kernel
    // instantiation when read (the per-launch duration of the dominant kernel for bench.py's roofline; off by default)
    struct PassRec { int R, mode, lm; bool inv; size_t elems; hipEvent_t a, b; };
    bool pass_timing = false;
    std::vector<PassRec> pass_recs;
    ~NttTables() { for (PassRec& r : pass_recs) { if (r.a) (void)hipEventDestroy(r.a); if (r.b) (void)hipEventDestroy(r.b); } }
};

static u64* upload(DeviceCtx* ctx, const std::vector<u64>& v) {
    u64* d = (u64*)ctx->alloc_persistent(v.size() * 8);
    HIP_CHECK(hipMemcpyAsync(d, v.data(), v.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return d;
}

static std::vector<u64> powers(u64 base, size_t count) {
    std::vector<u64> v(count);
    u64 acc = 1;
    for (size_t i = 0; i < count; i++) { v[i] = acc; acc = gl_mul(acc, base); }
    return v;
}

static u64 root_for(int k, int inverse) {
    u64 w = gl_root_of_unity(k);
    return inverse ? gl_inv(w) : w;
}

static const u64* get_small(NttTables& t, int R, int inverse) {
    auto key = std::make_pair(R, inverse);
    auto it = t.small.find(key);
    if (it != t.small.end()) return it->second;
    std::vector<u64> v = powers(root_for(R, inverse), R ? ((size_t)1 << (R - 1)) : 1);
    return t.small[key] = upload(t.ctx, v);
}

static TwoLevel make_two_level(DeviceCtx* ctx, u64 w, int k) {
    TwoLevel tl;
    tl.h = (k + 1) / 2;
    std::vector<u64> lo = powers(w, (size_t)1 << tl.h);
    std::vector<u64> hi = powers(gl_pow(w, (u64)1 << tl.h), (size_t)1 << (k - tl.h));
    tl.lo = upload(ctx, lo);
    tl.hi = upload(ctx, hi);
    return tl;
}

static TwoLevel get_two(NttTables& t, int k, int inverse) {
    auto key = std::make_pair(k, inverse);
    auto it = t.two.find(key);
    if (it != t.two.end()) return it->second;
    return t.two[key] = make_two_level(t.ctx, root_for(k, inverse), k);
}

// per-coset tables for the LDE: coset c scales coefficient k by (7 * g^bitrev(c))^k, g of order n*blowup
// (cfft/serial.rs:32-41).  Layout: [coset][lo 2^h | hi 2^(log_n-h)] with a common stride.
static TwoLevel get_coset(NttTables& t, int log_n, int rate_bits, size_t* stride) {
    auto key = std::make_pair(log_n, rate_bits);
    const int h = (log_n + 1) / 2;
    const size_t nlo = (size_t)1 << h, nhi = (size_t)1 << (log_n - h);
    *stride = nlo + nhi;
    auto it = t.coset.find(key);
    if (it != t.coset.end()) return it->second;
    const int blow = 1 << rate_bits;
    std::vector<u64> all((nlo + nhi) * blow);
    const u64 g = gl_root_of_unity(log_n + rate_bits);
    for (int c = 0; c < blow; c++) {
        const u64 s = gl_mul(gl_pow(g, bitrev32((u32)c, rate_bits)), GL_GENERATOR);
        std::vector<u64> lo = powers(s, nlo), hi = powers(gl_pow(s, (u64)1 << h), nhi);
        std::copy(lo.begin(), lo.end(), all.begin() + c * (nlo + nhi));
        std::copy(hi.begin(), hi.end(), all.begin() + c * (nlo + nhi) + nlo);
    }
    TwoLevel tl;
    tl.h = h;
    tl.lo = upload(t.ctx, all);
    tl.hi = tl.lo + nlo;
    return t.coset[key] = tl;
}

static TwoLevel get_shift(NttTables& t, int log_n, u64 shift) {
    auto key = std::make_pair(log_n, shift);
    auto it = t.shift.find(key);
    if (it != t.shift.end()) return it->second;
    return t.shift[key] = make_two_level(t.ctx, shift, log_n);
}

NttTables* ntt_tables_create(DeviceCtx* ctx) {
    NttTables* t = new NttTables();
    t->ctx = ctx;
    return t;
}
void ntt_tables_destroy(NttTables* t) { delete t; }  // device memory is owned by the ctx persistent pool

template <int R, int LOGT, int MODE>
static void launch_pass(const NttPassParams& p, size_t tiles, size_t cols, size_t cosets, hipStream_t stream) {
    constexpr int E = 1 << (R + LOGT);
    constexpr int NT = ntt_threads<R, LOGT>();
    const size_t lds_bytes = (size_t)(E + (E >> 4) + 1) * 8;
    auto kern = ntt_pass_kernel<R, LOGT, MODE>;
    // the attribute is a property of the kernel ON ONE DEVICE: a context that spans several GPUs sets it once per device
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (lds_bytes > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev].load()) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set[dev].store(true);
    }
    dim3 grid((unsigned)tiles, (unsigned)cols, (unsigned)cosets);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes, stream, p);
}

template <int MODE, int LOGT>
static void dispatch_pass(int R, const NttPassParams& p, size_t tiles, size_t cols, size_t cosets, hipStream_t s) {
    switch (R) {
#define OLA_CASE(r) case r: launch_pass<r, LOGT, MODE>(p, tiles, cols, cosets, s); break;
        OLA_CASE(1) OLA_CASE(2) OLA_CASE(3) OLA_CASE(4) OLA_CASE(5) OLA_CASE(6) OLA_CASE(7) OLA_CASE(8) OLA_CASE(9)
#undef OLA_CASE
        default: break;
    }
}
static void dispatch_contig(int R, const NttPassParams& p, size_t tiles, size_t cols, size_t cosets, hipStream_t s) {
    switch (R) {
#define OLA_CASE(r) case r: launch_pass<r, 0, MODE_CONTIG>(p, tiles, cols, cosets, s); break;
        OLA_CASE(0) OLA_CASE(1) OLA_CASE(2) OLA_CASE(3) OLA_CASE(4) OLA_CASE(5) OLA_CASE(6) OLA_CASE(7) OLA_CASE(8) OLA_CASE(9)
        OLA_CASE(10) OLA_CASE(11) OLA_CASE(12) OLA_CASE(13)
#undef OLA_CASE
        default: break;
    }
}

static std::vector<int> split_even(int total, int parts) {
    std::vector<int> v;
    for (int i = 0; i < parts; i++) {
        int k = (total + (parts - i) - 1) / (parts - i);
        v.push_back(k);
        total -= k;
    }
    return v;
}

// Pass plan for a 2^L transform.  Strided passes carry <= 9 bits (tile 2^R x 16), the closing contiguous pass <= 13
// bits, the closing natural-order ("rows") pass <= 9 bits.
struct NttPlan {
    std::vector<int> strided;  // R of each strided pass, top bits first
    int last_R = 0;
    int last_mode = MODE_CONTIG;
};

static NttPlan make_plan(int L, bool natural_out) {
    NttPlan pl;
    if (L <= 13) { pl.last_R = L; pl.last_mode = MODE_CONTIG; return pl; }
    if (!natural_out) {
        int best_rc = 12, best_p = (L - 12 + 8) / 9;
        int p13 = (L - 13 + 8) / 9;
        if (p13 < best_p) { best_rc = 13; best_p = p13; }
        pl.last_R = best_rc;
        pl.last_mode = MODE_CONTIG;
        pl.strided = split_even(L - best_rc, best_p);
    } else {
        int best_r = 8, best_p = (L - 8 + 8) / 9;
        int p9 = (L - 9 + 8) / 9;
        if (p9 < best_p) { best_r = 9; best_p = p9; }
        pl.last_R = best_r;
        pl.last_mode = MODE_ROWS;
        pl.strided = split_even(L - best_r, best_p);
    }
    return pl;
}

// Generic driver.  `in` may equal `out` only when the plan has no ROWS pass (bit-reversed or single-tile output).
// With a ROWS pass, `scratch` (same shape as out) receives the intermediate passes.
//   inverse:      use w^-1 and scale by 2^-L at the end
//   prescale:     nullptr, or two-level tables of s^k applied to input element k (coset transforms)
//   cosets:       number of output cosets (LDE); coset c writes at out + c * out_coset_stride and uses
//                 prescale tables at + c * sc_coset_stride
void ntt_run(NttTables& t, const u64* in, size_t in_col_stride, u64* out, size_t out_col_stride, u64* scratch,
             size_t scratch_col_stride, int L, size_t cols, bool inverse, bool natural_out, const TwoLevel* prescale,
             size_t sc_coset_stride, size_t cosets, size_t out_coset_stride, u64 extra_scale) {
    if (cols == 0) return;
    hipStream_t stream = t.ctx->stream;
    NttPlan pl = make_plan(L, natural_out);
    u64 final_scale = extra_scale;
    if (inverse) final_scale = gl_mul(final_scale, gl_inv(((u64)1 << L) % GL_P));

    const bool rows = pl.last_mode == MODE_ROWS;
    // buffer the strided passes work in: `scratch` when a ROWS pass follows (it is out of place), else `out`
    u64* work = rows ? scratch : out;
    size_t work_col_stride = rows ? scratch_col_stride : out_col_stride;
    size_t work_coset_stride = rows ? 0 : out_coset_stride;

    const u64* cur_in = in;
    size_t cur_in_stride = in_col_stride;
    size_t cur_in_coset_stride = 0;
    bool first = true;
    int lo = L;
    for (size_t i = 0; i < pl.strided.size(); i++) {
        const int R = pl.strided[i];
        lo -= R;
        NttPassParams p = {};
        p.in = cur_in; p.out = work;
        p.in_col_stride = cur_in_stride; p.out_col_stride = work_col_stride;
        p.in_coset_stride = cur_in_coset_stride; p.out_coset_stride = work_coset_stride;
        p.log_n = L; p.lo = lo;
        p.tw_small = get_small(t, R, inverse);
        TwoLevel tw = get_two(t, lo + R, inverse);
        p.tw_lo = tw.lo; p.tw_hi = tw.hi; p.tw_h = tw.h;
        if (first && prescale) { p.sc_lo = prescale->lo; p.sc_hi = prescale->hi; p.sc_h = prescale->h; p.sc_coset_stride = sc_coset_stride; }
        p.out_scale = 1;
        const size_t tiles = ((size_t)1 << (L - R)) >> 4;
        dispatch_pass<MODE_STRIDED, 4>(R, p, tiles, cols, cosets, stream);
        // after the first pass every coset reads its own slice of the work buffer
        if (first) {
            first = false;
            cur_in = work; cur_in_stride = work_col_stride; cur_in_coset_stride = work_coset_stride;
        }
    }
    // closing pass
    {
        const int R = pl.last_R;
        NttPassParams p = {};
        p.log_n = L; p.lo = 0;
        p.tw_small = get_small(t, R, inverse);
        p.out_scale = final_scale;
        p.natural_out = natural_out ? 1 : 0;
        if (first && prescale) { p.sc_lo = prescale->lo; p.sc_hi = prescale->hi; p.sc_h = prescale->h; p.sc_coset_stride = sc_coset_stride; }
        p.in = cur_in; p.in_col_stride = cur_in_stride; p.in_coset_stride = cur_in_coset_stride;
        p.out = out; p.out_col_stride = out_col_stride; p.out_coset_stride = out_coset_stride;
        if (rows) {
            const size_t tiles = ((size_t)1 << (L - R)) >> 4;
            dispatch_pass<MODE_ROWS, 4>(R, p, tiles, cols, cosets, stream);
        } else {
            dispatch_contig(R, p, (size_t)1 << (L - R), cols, cosets, stream);
        }
    }
}

// ---- public (library-internal) entry points -----------------------------------------------------
// Transforms of 2^14 points and more run on the second-generation passes (ntt2.hip); smaller ones fit one tile here.
void ntt2_run(NttTables& t, const u64* in, size_t in_col_stride, u64* out, size_t out_col_stride, u64* scratch,
              size_t scratch_col_stride, int L, size_t cols, bool inverse, bool natural_out, int sc_rate_bits, u64 sc_shift,
              size_t cosets, size_t out_coset_stride, size_t coset_first = 0);
static const int NTT2_MIN_LOG = 14;

// values (natural) -> coefficients (natural), per column.  scratch must hold cols * 2^L elements when L > 13.
void ntt_interpolate(NttTables& t, const u64* values, u64* coeffs, u64* scratch, int L, size_t cols) {
    const size_t n = (size_t)1 << L;
    PhaseScope ph(t.ctx, PH_INTT, (double)cols * n * 16, (double)cols);
    if (L >= NTT2_MIN_LOG) { ntt2_run(t, values, n, coeffs, n, scratch, n, L, cols, true, true, -2, 0, 1, 0); return; }
    ntt_run(t, values, n, coeffs, n, scratch, n, L, cols, true, true, nullptr, 0, 1, 0, 1);
}
// coefficients (natural) -> values at w^i (natural)
void ntt_evaluate(NttTables& t, const u64* coeffs, u64* values, u64* scratch, int L, size_t cols) {
    const size_t n = (size_t)1 << L;
    if (L >= NTT2_MIN_LOG) { ntt2_run(t, coeffs, n, values, n, scratch, n, L, cols, false, true, -2, 0, 1, 0); return; }
    ntt_run(t, coeffs, n, values, n, scratch, n, L, cols, false, true, nullptr, 0, 1, 0, 1);
}
// coefficients (natural, n per column) -> LDE on 7*<g>, in commitment leaf order: out[col][c*n + r] =
// P(7 * g^bitrev(c) * w_n^bitrev_n(r)); equals natural LDE row bitrev_N(c*n + r).
void ntt_lde_leaf_order(NttTables& t, const u64* coeffs, u64* lde, int L, int rate_bits, size_t cols, size_t coset_first,
                        size_t coset_count, size_t in_col_stride = 0) {
    // cosets [coset_first, coset_first + coset_count) of the 2^rate_bits (leaf-order blocks of n); lde holds only those.
    // in_col_stride: distance between the coefficient columns (default n)
    const size_t n = (size_t)1 << L;
    if (!in_col_stride) in_col_stride = n;
    PhaseScope ph(t.ctx, PH_LDE, (double)cols * n * 8 * (1 + coset_count), (double)cols * coset_count);
    if (L >= NTT2_MIN_LOG) {
        ntt2_run(t, coeffs, in_col_stride, lde, n * coset_count, nullptr, 0, L, cols, false, false, rate_bits, 0, coset_count, n, coset_first);
        return;
    }
    size_t stride = 0;
    TwoLevel sc = get_coset(t, L, rate_bits, &stride);
    sc.lo += coset_first * stride;
    sc.hi += coset_first * stride;
    ntt_run(t, coeffs, in_col_stride, lde, n * coset_count, nullptr, 0, L, cols, false, false, &sc, stride, coset_count, n, 1);
}
void ntt_lde_leaf_order(NttTables& t, const u64* coeffs, u64* lde, int L, int rate_bits, size_t cols) {
    ntt_lde_leaf_order(t, coeffs, lde, L, rate_bits, cols, 0, (size_t)1 << rate_bits);
}
// coefficients -> values on shift*<w_n> in natural order (coset_fft with blowup 1), or bit-reversed order
void ntt_coset_evaluate(NttTables& t, const u64* coeffs, u64* values, u64* scratch, int L, size_t cols, u64 shift,
                        bool natural_out) {
    const size_t n = (size_t)1 << L;
    if (L >= NTT2_MIN_LOG) { ntt2_run(t, coeffs, n, values, n, scratch, n, L, cols, false, natural_out, -1, shift, 1, 0); return; }
    TwoLevel sc = get_shift(t, L, shift);
    ntt_run(t, coeffs, n, values, n, scratch, n, L, cols, false, natural_out, &sc, 0, 1, 0, 1);
}

}  // namespace ola
