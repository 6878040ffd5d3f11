// Host check of the third-generation NTT pass (olavm_amd/csrc/ntt3_core.cuh, tform.cuh): the per-thread phases the GPU kernels
// run are executed here thread by thread on a RANGE-CHECKED limb type and compared with a plain radix-2 transform built from
// the canonical field functions (gl.cuh) -- the same recursion as the reference's cfft/serial.rs:89-127.
//   * every 32-bit limb operation asserts that its result fits a signed 32-bit register, every 64-bit sum that it fits 64 bits;
//   * every LDS access pattern (per round, per half-wave of 32 lanes, 8-byte accesses) is checked for bank conflicts;
//   * forward / inverse, natural / bit-reversed output, coset (LDE) transforms, all pass widths the planner uses.
// build: g++ -O2 -std=c++17 -o host_ntt3_check tests/host_ntt3_check.cpp ; run: ./host_ntt3_check
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../olavm_amd/csrc/ntt3_core.cuh"

using namespace ola;

// ------------------------------------------------------------------------------------------------ checked limb types
static long long g_max_abs = 0;
struct Chk64 {
    __int128 v;
    Chk64() : v(0) {}
    explicit Chk64(long long x) : v(x) {}
};
struct Chk32 {
    long long v;
    Chk32() : v(0) {}
    Chk32(long long x) : v(x) {
        if (x < -(1ll << 31) || x >= (1ll << 31)) { fprintf(stderr, "limb overflow: %lld\n", x); abort(); }
        const long long a = x < 0 ? -x : x;
        if (a > g_max_abs) g_max_abs = a;
    }
    friend Chk32 operator+(Chk32 a, Chk32 b) { return Chk32(a.v + b.v); }
    friend Chk32 operator-(Chk32 a, Chk32 b) { return Chk32(a.v - b.v); }
    Chk32 operator-() const { return Chk32(-v); }
    friend Chk32 operator&(Chk32 a, Chk32 b) { return Chk32((long long)(int)((unsigned)a.v & (unsigned)b.v)); }
    friend Chk32 operator<<(Chk32 a, int s) { return Chk32(a.v * (1ll << s)); }   // must not overflow either
    friend Chk32 operator>>(Chk32 a, int s) { return Chk32(a.v >> s); }           // arithmetic
};
namespace ola {
template <> struct TfTraits<Chk32> {
    typedef Chk64 W;
    static W mad(Chk32 a, Chk32 b, W c) {
        W r;
        r.v = (__int128)a.v * b.v + c.v;
        if (r.v < -((__int128)1 << 63) || r.v >= ((__int128)1 << 63)) { fprintf(stderr, "64-bit sum overflow\n"); abort(); }
        return r;
    }
    static u32 lo32(W z) { return (u32)(u64)(long long)z.v; }
    static Chk32 hi32(W z) { return Chk32((long long)z.v >> 32); }
    static Chk32 from_u32(u32 x) { return Chk32((long long)(int)x); }
    static u32 to_u32_biased(Chk32 v, u32 bias) {
        if (bias) {   // the conversion's bias must land in [0, 2^32)
            const long long s = v.v + (long long)bias;
            if (s < 0 || s >= (1ll << 32)) { fprintf(stderr, "bias overflow\n"); abort(); }
            return (u32)s;
        }
        return (u32)(int)v.v;
    }
};
}  // namespace ola

// ------------------------------------------------------------------------------------------------ reference transform
static u32 brev(u32 x, int bits) { u32 r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i); return r; }

// natural in -> natural out, X[k] = sum x[i] w^(ik), w = root (forward) or its inverse (then scaled by 1/n)
static std::vector<u64> ref_ntt(std::vector<u64> x, int L, bool inverse) {
    const size_t n = (size_t)1 << L;
    for (size_t i = 0; i < n; i++) { size_t j = brev((u32)i, L); if (i < j) std::swap(x[i], x[j]); }
    u64 root = gl_root_of_unity(L);
    if (inverse) root = gl_inv(root);
    for (int s = 1; s <= L; s++) {
        const size_t m = (size_t)1 << s, half = m >> 1;
        const u64 wm = gl_pow(root, n >> s);
        std::vector<u64> tw(half);
        tw[0] = 1;
        for (size_t j = 1; j < half; j++) tw[j] = gl_mul(tw[j - 1], wm);
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                const u64 t = gl_mul(tw[j], x[k + j + half]), u = x[k + j];
                x[k + j] = gl_add(u, t);
                x[k + j + half] = gl_sub(u, t);
            }
    }
    if (inverse) { const u64 ninv = gl_inv((u64)n % GL_P); for (auto& v : x) v = gl_mul(v, ninv); }
    return x;
}

// ------------------------------------------------------------------------------------------------ tables
struct Two { std::vector<u64> lo, hi; int h; };
static std::vector<u64> powers(u64 b, size_t n) { std::vector<u64> v(n); u64 w = 1; for (size_t i = 0; i < n; i++) { v[i] = w; w = gl_mul(w, b); } return v; }
static Two two_level(u64 base, int k) {
    Two t; t.h = (k + 1) / 2;
    t.lo = powers(base, (size_t)1 << t.h);
    t.hi = powers(gl_pow(base, (u64)1 << t.h), (size_t)1 << (k - t.h));
    return t;
}

// ------------------------------------------------------------------------------------------------ LDS conflict check
static long g_conflicts = 0, g_lds_ops = 0;
static void check_lanes(const int (&slot)[64]) {   // one 8-byte access per lane: two half-waves, 32 bank pairs
    for (int half = 0; half < 2; half++) {
        int seen[32] = {0};
        for (int l = 0; l < 32; l++) seen[slot[half * 32 + l] & 31]++;
        for (int b = 0; b < 32; b++) if (seen[b] > 1) g_conflicts += seen[b] - 1;
        g_lds_ops++;
    }
}
template <int R, int MODE>
static void check_conflicts() {
    typedef N3Cfg<R, MODE> C;
    for (int r = 0; r < C::NR; r++)
        for (int j = 0; j < 32; j++)
            for (int wave = 0; wave < 4; wave++) {
                int s[64];
                for (int l = 0; l < 64; l++) s[l] = n3_slot(C::tid_t(r, wave * 64 + l)) ^ n3_slot(C::reg_t(r, j));
                check_lanes(s);   // written after round r (r < NR-1), read before round r (r > 0), final transpose (last)
            }
    if (MODE == N3_LAST_BITREV)
        for (int jj = 0; jj < 32; jj++)
            for (int wave = 0; wave < 4; wave++) {
                int s[64];
                for (int l = 0; l < 64; l++) s[l] = n3_slot((jj << 8) | (wave * 64 + l));
                check_lanes(s);
            }
    // the maps must be bijections of the tile
    for (int r = 0; r < C::NR; r++) {
        std::vector<char> hit(8192, 0);
        for (int tid = 0; tid < 256; tid++)
            for (int j = 0; j < 32; j++) {
                const int t = C::tid_t(r, tid) | C::reg_t(r, j);
                assert(!hit[t]);
                hit[t] = 1;
                assert(n3_slot(t) == (n3_slot(C::tid_t(r, tid)) ^ n3_slot(C::reg_t(r, j))));
            }
    }
}

// ------------------------------------------------------------------------------------------------ pass emulation
template <int R, int MODE, bool INV, int RND, class I>
static void rounds_from(const N3Params& p, const N3Addr<R, MODE>& a, u32 coset, std::vector<std::vector<T4<I>>>& regs, std::vector<u64>& lds) {
    typedef N3Cfg<R, MODE> C;
    for (int tid = 0; tid < 256; tid++) {
        T4<I>(&x)[N3_REGS] = *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data());
        n3_round<R, MODE, INV, RND, I>(p, a, tid, coset, x);
    }
    if constexpr (RND + 1 < C::NR) {
        for (int tid = 0; tid < 256; tid++) n3_xchg_write<R, MODE, RND, 0, I>(tid, *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data()), lds.data());
        for (int tid = 0; tid < 256; tid++) n3_xchg_read<R, MODE, RND, 0, I>(tid, *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data()), lds.data());
        for (int tid = 0; tid < 256; tid++) n3_xchg_write<R, MODE, RND, 1, I>(tid, *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data()), lds.data());
        for (int tid = 0; tid < 256; tid++) n3_xchg_read<R, MODE, RND, 1, I>(tid, *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data()), lds.data());
        rounds_from<R, MODE, INV, RND + 1, I>(p, a, coset, regs, lds);
    }
}

template <int R, int MODE, bool INV, class I>
static void run_pass(N3Params p, size_t cols, size_t cosets) {
    typedef N3Cfg<R, MODE> C;
    const size_t tiles = (size_t)1 << (p.log_n - 13);
    if (MODE == N3_STRIDED) assert(p.lo >= C::SB);
    std::vector<u64> lds(8192);
    for (size_t coset = 0; coset < cosets; coset++)
        for (size_t tile = 0; tile < tiles; tile++) {
            N3Addr<R, MODE> a;
            a.init(p.log_n, p.lo, (u32)tile);
            for (size_t col = 0; col < cols; col++) {
                const u64* in = p.in + col * p.in_col_stride + coset * p.in_coset_stride;
                u64* out = p.out + col * p.out_col_stride + coset * p.out_coset_stride;
                std::vector<std::vector<T4<I>>> regs(256, std::vector<T4<I>>(N3_REGS));
                for (int tid = 0; tid < 256; tid++) n3_load<R, MODE, I>(p, a, in, tid, (u32)coset, *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data()));
                rounds_from<R, MODE, INV, 0, I>(p, a, (u32)coset, regs, lds);
                if (MODE == N3_LAST_BITREV) {
                    for (int tid = 0; tid < 256; tid++) n3_final_write<R, MODE, I>(tid, *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data()), lds.data());
                    for (int tid = 0; tid < 256; tid++) n3_final_store<R, MODE>(a, out, tid, lds.data());
                } else {
                    for (int tid = 0; tid < 256; tid++) n3_store_direct<R, MODE, I>(a, out, tid, *reinterpret_cast<T4<I>(*)[N3_REGS]>(regs[tid].data()));
                }
            }
        }
}

template <int MODE, bool INV, class I>
static void dispatch_pass(int R, const N3Params& p, size_t cols, size_t cosets) {
    if constexpr (MODE == N3_STRIDED) {
        switch (R) {
            case 5: run_pass<5, MODE, INV, I>(p, cols, cosets); break;
            case 6: run_pass<6, MODE, INV, I>(p, cols, cosets); break;
            case 7: run_pass<7, MODE, INV, I>(p, cols, cosets); break;
            case 8: run_pass<8, MODE, INV, I>(p, cols, cosets); break;
            case 9: run_pass<9, MODE, INV, I>(p, cols, cosets); break;
            default: assert(0);
        }
    } else if constexpr (MODE == N3_LAST_BITREV) run_pass<13, MODE, INV, I>(p, cols, cosets);
    else run_pass<9, MODE, INV, I>(p, cols, cosets);
}

static std::vector<int> split_even(int total, int parts) {
    std::vector<int> v;
    for (int i = 0; i < parts; i++) { int k = (total + (parts - i) - 1) / (parts - i); v.push_back(k); total -= k; }
    return v;
}

// the driver the GPU library runs (ntt3_run), restated on host vectors.  coset_shifts empty: plain transform.
template <class I>
static std::vector<u64> ntt3_host(const std::vector<u64>& in, int L, size_t cols, bool inverse, bool natural_out, const std::vector<u64>& coset_shifts) {
    const size_t n = (size_t)1 << L, cosets = coset_shifts.empty() ? 1 : coset_shifts.size();
    const int lastR = natural_out ? 9 : 13;
    const int sb = L - lastR;
    std::vector<int> Rs = split_even(sb, (sb + 8) / 9);
    std::vector<u64> work(cols * n * cosets), out(cols * n * cosets);
    const u64 scale = inverse ? gl_inv((u64)n % GL_P) : 1;
    const u64* cur_in = in.data(); size_t cur_coset_stride = 0;
    int lo = L;
    std::vector<std::vector<u64>> keep;   // tables stay alive
    for (size_t i = 0; i < Rs.size(); i++) {
        const int R = Rs[i];
        lo -= R;
        N3Params p = {};
        p.in = cur_in; p.in_col_stride = n * (cur_in == in.data() ? 1 : cosets); p.in_coset_stride = cur_coset_stride;
        p.out = work.data(); p.out_col_stride = n * cosets; p.out_coset_stride = n;
        p.log_n = L; p.lo = lo; p.ncols = cols;
        u64 rootR = gl_root_of_unity(R), rootP = gl_root_of_unity(lo + R);
        if (inverse) { rootR = gl_inv(rootR); rootP = gl_inv(rootP); }
        keep.push_back(powers(rootR, (size_t)1 << R)); p.tw = keep.back().data();
        // pass multipliers at the in-place offset o = d * 2^lo + M:  w^(M * bitrev_R(d)) * scale * s^M
        const u64 ps = (i + 1 == Rs.size()) ? scale : 1;
        const size_t blk = (size_t)1 << (lo + R);
        std::vector<u64> ptw(blk * cosets);
        for (size_t cs = 0; cs < cosets; cs++) {
            std::vector<u64> sM;
            if (i == 0 && !coset_shifts.empty()) sM = powers(coset_shifts[cs], (size_t)1 << lo);
            for (size_t d = 0; d < ((size_t)1 << R); d++) {
                const u64 wq = gl_pow(rootP, brev((u32)d, R));
                u64 w = ps;
                for (size_t M = 0; M < ((size_t)1 << lo); M++) {
                    ptw[cs * blk + (d << lo) + M] = sM.empty() ? w : gl_mul(w, sM[M]);
                    w = gl_mul(w, wq);
                }
            }
        }
        keep.push_back(ptw); p.ptw = keep.back().data(); p.ptw_coset_stride = blk;
        if (i == 0 && !coset_shifts.empty()) {
            std::vector<u64> dig;
            for (u64 sft : coset_shifts) { std::vector<u64> dd = powers(gl_pow(sft, (u64)1 << lo), (size_t)1 << R); dig.insert(dig.end(), dd.begin(), dd.end()); }
            keep.push_back(dig); p.sc_dig = keep.back().data();
        }
        if (inverse) dispatch_pass<N3_STRIDED, true, I>(R, p, cols, cosets); else dispatch_pass<N3_STRIDED, false, I>(R, p, cols, cosets);
        cur_in = work.data(); cur_coset_stride = n;
    }
    N3Params p = {};
    p.in = cur_in; p.in_col_stride = n * cosets; p.in_coset_stride = n;
    p.out = out.data(); p.out_col_stride = n * cosets; p.out_coset_stride = n;
    p.log_n = L; p.lo = 0; p.ncols = cols;
    u64 rootR = gl_root_of_unity(lastR);
    if (inverse) rootR = gl_inv(rootR);
    keep.push_back(powers(rootR, (size_t)1 << lastR)); p.tw = keep.back().data();
    if (natural_out) { if (inverse) dispatch_pass<N3_LAST_NATURAL, true, I>(9, p, cols, cosets); else dispatch_pass<N3_LAST_NATURAL, false, I>(9, p, cols, cosets); }
    else { if (inverse) dispatch_pass<N3_LAST_BITREV, true, I>(13, p, cols, cosets); else dispatch_pass<N3_LAST_BITREV, false, I>(13, p, cols, cosets); }
    return out;
}

static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() { u64 z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <class I>
static int check_case(int L, bool inverse, bool natural_out, bool coset, const char* what) {
    const size_t n = (size_t)1 << L, cols = 2;
    std::vector<u64> in(cols * n);
    static const u64 edge[6] = {0, 1, GL_P - 1, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFFFFFFFFFFull};   // incl. a non-canonical word
    for (size_t i = 0; i < n; i++) { in[i] = rnd(); in[n + i] = edge[(i * 7 + i / 5) % 6]; }
    std::vector<u64> shifts;
    if (coset) { const u64 g = gl_root_of_unity(L + 1); shifts = {GL_GENERATOR, gl_mul(GL_GENERATOR, g)}; }
    std::vector<u64> got = ntt3_host<I>(in, L, cols, inverse, natural_out, shifts);
    const size_t cosets = coset ? shifts.size() : 1;
    int bad = 0;
    for (size_t c = 0; c < cols; c++)
        for (size_t cs = 0; cs < cosets; cs++) {
            std::vector<u64> x(in.begin() + c * n, in.begin() + (c + 1) * n);
            for (auto& v : x) v = gl_canon(v);
            if (coset) { u64 w = 1; for (size_t i = 0; i < n; i++) { x[i] = gl_mul(x[i], w); w = gl_mul(w, shifts[cs]); } }
            std::vector<u64> want = ref_ntt(x, L, inverse);
            const u64* g = got.data() + c * n * cosets + cs * n;
            for (size_t i = 0; i < n; i++) {
                const u64 wv = natural_out ? want[i] : want[brev((u32)i, L)];
                if (g[i] != wv) { if (bad < 5) fprintf(stderr, "  mismatch %s L=%d col=%zu coset=%zu i=%zu got=%llx want=%llx\n", what, L, c, cs, i, g[i], wv); bad++; }
            }
        }
    printf("%-34s L=%2d inverse=%d natural=%d coset=%d : %s\n", what, L, inverse, natural_out, coset, bad ? "FAIL" : "ok");
    return bad != 0;
}

#ifndef NTT3_NO_MAIN
int main(int argc, char** argv) {
    int fails = 0;
    // arithmetic: T-form <-> canonical, multiply, shifts
    for (int e = 1; e <= 6; e++) { u64 w = gl_root_of_unity(e); u64 p2 = gl_pow(2, (u64)tf_root_exp(e, false)); if (w != p2) { printf("root exponent %d wrong\n", e); fails++; } if (gl_mul(w, gl_pow(2, (u64)tf_root_exp(e, true))) != 1) { printf("inverse root exponent %d wrong\n", e); fails++; } }
    for (int it = 0; it < 200000; it++) {
        const u64 a = rnd(), b = rnd();
        T4<Chk32> ta = tf_from_u64<Chk32>(a), tb = tf_from_u64<Chk32>(b);
        if (tf_to_u64(ta) != gl_canon(a)) { printf("round trip\n"); fails++; break; }
        if (tf_to_u64(tf_add(ta, tb)) != gl_add(gl_canon(a), gl_canon(b))) { printf("add\n"); fails++; break; }
        if (tf_to_u64(tf_mul(tf_sub(ta, tb), tf_split_u64(b))) != gl_mul(gl_sub(gl_canon(a), gl_canon(b)), gl_canon(b))) { printf("mul\n"); fails++; break; }
        if (tf_to_u64(tf_sub_mul_pow2<78>(ta, tb)) != gl_mul(gl_sub(gl_canon(a), gl_canon(b)), gl_pow(2, 78))) { printf("pow2 78\n"); fails++; break; }
        if (tf_to_u64(tf_sub_mul_pow2<186>(ta, tb)) != gl_mul(gl_sub(gl_canon(a), gl_canon(b)), gl_pow(2, 186))) { printf("pow2 186\n"); fails++; break; }
        if (tf_to_u64(tf_sub_mul_pow2<120>(ta, tb)) != gl_mul(gl_sub(gl_canon(a), gl_canon(b)), gl_pow(2, 120))) { printf("pow2 120\n"); fails++; break; }
    }
    {   // extreme limb magnitudes through the conversion
        const long long E[6] = {-(1ll << 31) + 129, (1ll << 31) - 129, 0, 1, -1, (1ll << 30)};
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) for (int c = 0; c < 6; c++) for (int d = 0; d < 6; d++) {
            T4<Chk32> t; t.v[0] = Chk32(E[a]); t.v[1] = Chk32(E[b]); t.v[2] = Chk32(E[c]); t.v[3] = Chk32(E[d]);
            __int128 val = 0; const long long L4[4] = {E[a], E[b], E[c], E[d]};
            // value mod p with python-style big arithmetic
            unsigned __int128 acc = 0; const u64 T = 1ull << 24; u64 pw = 1;
            u64 sum = 0;
            for (int i = 0; i < 4; i++) { const u64 limb = L4[i] >= 0 ? (u64)L4[i] % GL_P : gl_neg((u64)(-L4[i]) % GL_P); sum = gl_add(sum, gl_mul(limb, pw)); pw = gl_mul(pw, T); }
            (void)val; (void)acc;
            if (tf_to_u64(t) != sum) { printf("extreme conversion\n"); fails++; }
        }
    }
    printf("T-form arithmetic: %s\n", fails ? "FAIL" : "ok");
    g_max_abs = 0;
    check_conflicts<5, N3_STRIDED>(); check_conflicts<6, N3_STRIDED>(); check_conflicts<7, N3_STRIDED>(); check_conflicts<8, N3_STRIDED>();
    check_conflicts<9, N3_STRIDED>(); check_conflicts<13, N3_LAST_BITREV>(); check_conflicts<9, N3_LAST_NATURAL>();
    printf("LDS: %ld half-wave accesses, %ld bank conflicts\n", g_lds_ops, g_conflicts);
    if (g_conflicts) fails++;
    const int maxL = argc > 1 ? atoi(argv[1]) : 20;
    for (int L = 18; L <= maxL; L++) {
        fails += check_case<Chk32>(L, false, false, false, "forward, bit-reversed");
        fails += check_case<Chk32>(L, false, true, false, "forward, natural");
        fails += check_case<Chk32>(L, true, true, false, "inverse, natural");
        fails += check_case<Chk32>(L, false, false, true, "coset LDE, bit-reversed");
        if (L <= 19) { fails += check_case<Chk32>(L, true, false, false, "inverse, bit-reversed"); fails += check_case<Chk32>(L, false, true, true, "coset, natural"); }
    }
    if (maxL >= 20) fails += check_case<i32>(20, false, true, false, "plain int limbs");
    printf("largest |limb| seen: 2^%.2f\n", __builtin_log2((double)g_max_abs));
    printf(fails ? "FAILED\n" : "all ok\n");
    return fails ? 1 : 0;
}
#endif
