// Multi-table STARK prover on the device: CTL / permutation Z columns, constraint quotient, per-table proof assembly.
//
// Replaces (reference paths relative to /root/reference/circuits/src/stark):
//   prover.rs:79-327        prove_with_traces         -> prove_with_traces() below (host orchestration, device data)
//   prover.rs:330-567       prove_single_table
//   prover.rs:571-705       compute_quotient_polys    -> quotient_kernel (one thread per LDE point, constraint program
//                                                        interpreted out of an LDS register file) + coset iNTT
//   cross_table_lookup.rs:224-311  cross_table_lookup_data / partial_products -> ctl_factor_kernel + product scan
//   permutation.rs:103-155  compute_permutation_z_polys -> perm_factor_kernel + product scan
//   constraint_consumer.rs:34-78, vanishing_poly.rs:20-45, permutation.rs:302-360, cross_table_lookup.rs:380-421
//                           ConstraintConsumer / eval_vanishing_poly / permutation + CTL checks (inside quotient_kernel)
//   serialization.rs:349-358,377-393   write_proof / write_all_proof
// The table descriptions (constraint programs, permutation pairs, CTLs) are data: the AIR-set blob produced by
// olavm_amd/air/dsl.py.  The sequential prefix products of the reference become three-phase parallel scans; the
// quotient is evaluated directly on the resident LDE in leaf order (thread j <-> leaf j <-> natural LDE row bitrev(j)),
// which for a table of quotient-degree 2^qdb are exactly the first n*2^qdb leaves (cosets 0..2^qdb-1).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ola_gpu.h"
#include "device_ctx.h"
#include "gl.cuh"
#include "airq.cuh"
#include "upload.h"

namespace ola {

// ------------------------------------------------------------------------------------------------ AIR-set (host)
enum { AOP_LOCAL = 0, AOP_NEXT, AOP_CONST, AOP_PARAM, AOP_ADD, AOP_SUB, AOP_MUL, AOP_EMIT, AOP_ISZERO };
enum { AK_ALL = 0, AK_TRANSITION, AK_FIRST, AK_LAST };

struct HLinCol { std::vector<std::pair<u64, u64>> terms; u64 constant = 0; };
struct HTwc { int table = 0; std::vector<HLinCol> columns; bool has_filter = false; HLinCol filter; };
struct HCtl { std::vector<HTwc> looking; HTwc looked; };
struct HTable {
    int ncols = 0, constraint_degree = 0, n_regs = 0, n_params = 0;
    std::vector<std::vector<std::pair<u64, u64>>> perm_pairs;
    std::vector<u64> ops;  // 2 words per op
    int quotient_degree_factor() const { return std::max(1, constraint_degree - 1); }
    int permutation_batch_size() const { return quotient_degree_factor(); }
    int num_permutation_batches(int nch) const {
        const int inst = (int)perm_pairs.size() * nch;
        return inst ? (inst + permutation_batch_size() - 1) / permutation_batch_size() : 0;
    }
};
struct HAirSet { std::vector<HTable> tables; std::vector<HCtl> ctls; };

static HAirSet parse_airset(const u64* w, size_t n) {
    size_t p = 0;
    auto next = [&]() -> u64 {
        if (p >= n) throw OlaError(OLA_E_INVALID_ARG, "AIR-set blob truncated");
        return w[p++];
    };
    if (next() != 0x4F4C41414952ull || next() != 1) throw OlaError(OLA_E_INVALID_ARG, "AIR-set blob: bad magic/version");
    const size_t nt = next(), nc = next();
    HAirSet s;
    for (size_t t = 0; t < nt; t++) {
        HTable a;
        a.ncols = (int)next(); a.constraint_degree = (int)next(); a.n_regs = (int)next(); a.n_params = (int)next();
        const size_t np = next();
        for (size_t i = 0; i < np; i++) {
            const size_t len = next();
            std::vector<std::pair<u64, u64>> pr;
            for (size_t k = 0; k < len; k++) { const u64 l = next(); const u64 r = next(); pr.push_back({l, r}); }
            a.perm_pairs.push_back(pr);
        }
        const size_t nops = next();
        for (size_t i = 0; i < 2 * nops; i++) a.ops.push_back(next());
        s.tables.push_back(a);
    }
    auto col = [&]() { HLinCol c; const size_t k = next(); for (size_t i = 0; i < k; i++) { const u64 cc = next(); const u64 f = next(); c.terms.push_back({cc, f}); } c.constant = next(); return c; };
    auto twc = [&]() { HTwc t; t.table = (int)next(); const size_t k = next(); for (size_t i = 0; i < k; i++) t.columns.push_back(col()); t.has_filter = next() != 0; if (t.has_filter) t.filter = col(); return t; };
    for (size_t c = 0; c < nc; c++) {
        HCtl ctl;
        const size_t nl = next();
        for (size_t i = 0; i < nl; i++) ctl.looking.push_back(twc());
        ctl.looked = twc();
        for (auto& t : ctl.looking) if (t.table < 0 || t.table >= (int)nt) throw OlaError(OLA_E_INVALID_ARG, "CTL table index");
        if (ctl.looked.table < 0 || ctl.looked.table >= (int)nt) throw OlaError(OLA_E_INVALID_ARG, "CTL table index");
        s.ctls.push_back(ctl);
    }
    if (p != n) throw OlaError(OLA_E_INVALID_ARG, "AIR-set blob has trailing words");
    return s;
}

static void push_lincol(std::vector<u64>& d, const HLinCol& c) {
    d.push_back(c.terms.size());
    for (auto& t : c.terms) { d.push_back(t.first); d.push_back(t.second); }
    d.push_back(c.constant);
}
// [beta, gamma, ncols, lincol*, has_filter, lincol?]
static void push_ctl_desc(std::vector<u64>& d, const HTwc& t, u64 beta, u64 gamma) {
    d.push_back(beta); d.push_back(gamma); d.push_back(t.columns.size());
    for (auto& c : t.columns) push_lincol(d, c);
    d.push_back(t.has_filter ? 1 : 0);
    if (t.has_filter) push_lincol(d, t.filter);
}


// FNV-1a over the little-endian bytes of the words a specialised kernel depends on (olavm_amd/air/dsl.py
// AirSet.signature_words): the table description and the static part of each of its CTL Z columns.
struct SigHasher {
    u64 h = 0xCBF29CE484222325ull;
    void word(u64 x) { for (int k = 0; k < 8; k++) { h ^= (x >> (8 * k)) & 0xFF; h *= 0x100000001B3ull; } }
    void lincol(const HLinCol& c) { word(c.terms.size()); for (auto& t : c.terms) { word(t.first); word(t.second); } word(c.constant); }
};
static u64 air_signature(const HTable& a, const std::vector<const HTwc*>& jobs) {
    SigHasher s;
    s.word((u64)a.ncols); s.word((u64)a.constraint_degree); s.word((u64)a.n_regs); s.word((u64)a.n_params);
    s.word(a.perm_pairs.size());
    for (auto& pr : a.perm_pairs) { s.word(pr.size()); for (auto& lr : pr) { s.word(lr.first); s.word(lr.second); } }
    s.word(a.ops.size() / 2);
    for (u64 w : a.ops) s.word(w);
    s.word(jobs.size());
    for (const HTwc* t : jobs) {
        s.word(t->columns.size());
        for (auto& c : t->columns) s.lincol(c);
        s.word(t->has_filter ? 1 : 0);
        if (t->has_filter) s.lincol(t->filter);
    }
    return s.h;
}

// ------------------------------------------------------------------------------------------------ device helpers
// evaluate a linear combination of columns at row `row` of a column-major table (stride `cs`); advances the cursor
__device__ __forceinline__ u64 dev_lincol(const u64* __restrict__ d, u32& p, const u64* __restrict__ tab, size_t cs, size_t row) {
    const u32 nt = (u32)d[p++];
    u64 s = 0;
    for (u32 i = 0; i < nt; i++) {
        const u64 c = d[p++], f = d[p++];
        s = gl_add(s, gl_mul(gl_canon(tab[c * cs + row]), f));
    }
    return gl_add(s, d[p++]);
}

// CTL factor columns: out[i] = filter(i) ? combine(i) : 1   (cross_table_lookup.rs:284-311)
// A table carries one Z column per (lookup side, challenge); the num_challenges columns of one side combine the SAME linear
// combinations of trace columns with different (beta, gamma).  One blockIdx.y handles such a pair: descriptors at
// desc_all[offs[pairs[2y]]] and desc_all[offs[pairs[2y+1]]] (equal indices: an unpaired column), outputs at out_all + index*n.
// The trace cells and the linear combinations are read / evaluated once for both.
__device__ __forceinline__ u64 dev_lincol_fast(const u64* __restrict__ d, u32& p, const u64* __restrict__ tab, size_t cs, size_t row) {
    const u32 nt = (u32)d[p++];
    u64 s = 0;
    for (u32 i = 0; i < nt; i++) {
        const u64 c = d[p++], f = d[p++];
        const u64 v = gl_canon(tab[c * cs + row]);
        s = gl_add(s, f == 1 ? v : gl_mul(v, f));          // the coefficient is uniform: no divergence; most are 1
    }
    return gl_add(s, d[p++]);
}
__global__ __launch_bounds__(256) void ctl_factor_kernel(const u64* __restrict__ trace, size_t n, const u64* __restrict__ desc_all,
                                                         const u64* __restrict__ offs, const u64* __restrict__ pairs, u64* __restrict__ out_all,
                                                         unsigned* __restrict__ bad_filter) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 ya = pairs[2 * blockIdx.y], yb = pairs[2 * blockIdx.y + 1];
    const u64* __restrict__ desc = desc_all + offs[ya];
    const u64* __restrict__ descb = desc_all + offs[yb];
    u32 p = 0;
    const u64 beta_a = desc[0], gamma_a = desc[1], beta_b = descb[0], gamma_b = descb[1];
    p = 2;
    const u32 ncol = (u32)desc[p++];
    // combine = reduce_with_powers(evals, beta) + gamma = sum_k beta^k e_k + gamma
    u64 acc_a = 0, acc_b = 0, bp_a = 1, bp_b = 1;
    for (u32 k = 0; k < ncol; k++) {
        const u64 e = dev_lincol_fast(desc, p, trace, n, i);
        acc_a = gl_add(acc_a, k == 0 ? e : gl_mul(bp_a, e));
        acc_b = gl_add(acc_b, k == 0 ? e : gl_mul(bp_b, e));
        if (k + 1 < ncol) { bp_a = gl_mul(bp_a, beta_a); bp_b = gl_mul(bp_b, beta_b); }
    }
    acc_a = gl_add(acc_a, gamma_a);
    acc_b = gl_add(acc_b, gamma_b);
    u64 f = 1;
    if (desc[p++]) f = dev_lincol_fast(desc, p, trace, n, i);
    if (f > 1) atomicOr(bad_filter, 1u);  // cross_table_lookup.rs:303-305 panics "Non-binary filter?"
    out_all[ya * n + i] = (f == 1) ? acc_a : 1;
    if (yb != ya) out_all[yb * n + i] = (f == 1) ? acc_b : 1;
}

// permutation quotient column: out[i] = prod_inst (gamma + sum beta^k lhs_k) / prod_inst (gamma + sum beta^k rhs_k)
// desc: [n_inst, (beta, gamma, npairs, (lhs, rhs)*)*]
__global__ __launch_bounds__(256) void perm_factor_kernel(const u64* __restrict__ trace, size_t n, const u64* __restrict__ desc,
                                                          u64* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 p = 0;
    const u32 ninst = (u32)desc[p++];
    u64 num = 1, den = 1;
    for (u32 s = 0; s < ninst; s++) {
        const u64 beta = desc[p++], gamma = desc[p++];
        const u32 np = (u32)desc[p++];
        u64 l = gamma, r = gamma, w = 1;
        for (u32 k = 0; k < np; k++) {
            const u64 cl = desc[p++], cr = desc[p++];
            l = gl_add(l, gl_mul(gl_canon(trace[cl * n + i]), w));
            r = gl_add(r, gl_mul(gl_canon(trace[cr * n + i]), w));
            w = gl_mul(w, beta);
        }
        num = gl_mul(num, l);
        den = gl_mul(den, r);
    }
    out[i] = gl_mul(num, gl_inv(den));
}

// multiplicative inclusive prefix scan, 3 phases, blocks of 2048
#define PSCAN_B 2048
// (the scan kernels take one column per blockIdx.y: data at d + y*n, block totals at block_tot + y*tot_stride)
__global__ __launch_bounds__(256) void pscan_local_kernel(u64* __restrict__ d_all, size_t n, u64* __restrict__ tot_all, size_t tot_stride) {
    __shared__ u64 sh[256];
    u64* __restrict__ d = d_all + (size_t)blockIdx.y * n;
    u64* __restrict__ block_tot = tot_all + (size_t)blockIdx.y * tot_stride;
    const size_t b0 = (size_t)blockIdx.x * PSCAN_B;
    const int t = threadIdx.x;
    u64 v[8];
    u64 run = 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t k = b0 + (size_t)t * 8 + i;
        run = gl_mul(run, k < n ? d[k] : 1);
        v[i] = run;
    }
    sh[t] = run;
    __syncthreads();
    u64 incl = run;
    for (int s = 1; s < 256; s <<= 1) {
        const u64 other = (t >= s) ? sh[t - s] : 1;
        __syncthreads();
        incl = gl_mul(incl, other);
        sh[t] = incl;
        __syncthreads();
    }
    const u64 excl = (t > 0) ? sh[t - 1] : 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t k = b0 + (size_t)t * 8 + i;
        if (k < n) d[k] = gl_mul(v[i], excl);
    }
    if (t == 255) block_tot[blockIdx.x] = incl;
}
// exclusive running product of the block totals, one workgroup: each thread owns a contiguous chunk, chunk totals
// are combined with a Hillis-Steele scan in LDS
__global__ __launch_bounds__(256) void pscan_totals_kernel(u64* __restrict__ tot_all, size_t nblocks, size_t tot_stride) {
    __shared__ u64 sh[256];
    u64* __restrict__ tot = tot_all + (size_t)blockIdx.y * tot_stride;
    const int t = threadIdx.x;
    const size_t per = (nblocks + 255) / 256;
    const size_t lo = (size_t)t * per, hi = lo + per < nblocks ? lo + per : nblocks;
    u64 run = 1;
    for (size_t i = lo; i < hi; i++) run = gl_mul(run, tot[i]);
    sh[t] = run;
    __syncthreads();
    u64 incl = run;
    for (int s = 1; s < 256; s <<= 1) {
        const u64 other = (t >= s) ? sh[t - s] : 1;
        __syncthreads();
        incl = gl_mul(incl, other);
        sh[t] = incl;
        __syncthreads();
    }
    u64 acc = (t > 0) ? sh[t - 1] : 1;
    for (size_t i = lo; i < hi; i++) {
        const u64 v = tot[i];
        tot[i] = acc;
        acc = gl_mul(acc, v);
    }
}
__global__ __launch_bounds__(256) void pscan_apply_kernel(u64* __restrict__ d_all, size_t n, const u64* __restrict__ tot_all, size_t tot_stride) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    u64* __restrict__ d = d_all + (size_t)blockIdx.y * n;
    const u64* __restrict__ tot = tot_all + (size_t)blockIdx.y * tot_stride;
    d[k] = gl_mul(d[k], tot[k / PSCAN_B]);
}
// exclusive form: out[0] = 1, out[i] = incl[i-1]
__global__ __launch_bounds__(256) void shift_right_kernel(const u64* __restrict__ incl, u64* __restrict__ out, size_t n) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    out[k] = k ? incl[k - 1] : 1;
}

// running products of `ncols` adjacent columns (n each) in place; tot: scratch of ncols * pscan_tot_stride(n) words
static size_t pscan_tot_stride(size_t n) { return (n + PSCAN_B - 1) / PSCAN_B + 1; }
static void product_scan_inclusive(DeviceCtx* ctx, u64* cols, size_t n, u64* tot, size_t ncols = 1) {
    if (ncols == 0) return;
    const size_t nblocks = (n + PSCAN_B - 1) / PSCAN_B, ts = pscan_tot_stride(n);
    hipLaunchKernelGGL(pscan_local_kernel, dim3((unsigned)nblocks, (unsigned)ncols), dim3(256), 0, ctx->stream, cols, n, tot, ts);
    if (nblocks > 1) {
        hipLaunchKernelGGL(pscan_totals_kernel, dim3(1, (unsigned)ncols), dim3(256), 0, ctx->stream, tot, nblocks, ts);
        hipLaunchKernelGGL(pscan_apply_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)ncols), dim3(256), 0, ctx->stream, cols, n, tot, ts);
    }
}

// Lagrange selector polynomials in coefficient form: L_0 = (1/n) sum X^k ; L_{n-1} = (1/n) sum g^k X^k
__global__ __launch_bounds__(256) void lagrange_coeffs_kernel(u64* __restrict__ out, size_t n, u64 n_inv, const u64* __restrict__ g_lo,
                                                              const u64* __restrict__ g_hi, int g_h) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    out[k] = n_inv;
    const u64 gk = gl_mul(g_lo[k & (((size_t)1 << g_h) - 1)], g_hi[k >> g_h]);
    out[n + k] = gl_mul(n_inv, gk);
}

// any non-zero element in data[lo, hi)?
__global__ __launch_bounds__(256) void any_nonzero_kernel(const u64* __restrict__ data, size_t lo, size_t hi, unsigned* __restrict__ flag) {
    const size_t k = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < hi && gl_canon(data[k]) != 0) atomicOr(flag, 1u);
}

// ------------------------------------------------------------------------------------------------ quotient kernel

#define QW 64  // one wavefront per workgroup; register file = n_regs x 64 lanes in LDS
__global__ __launch_bounds__(QW) void quotient_kernel(QuotParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* regs = reinterpret_cast<u64*>(smem_raw);
    const int lane = threadIdx.x;
    const size_t size = P.npoints ? P.npoints : P.out_plane;   // points this launch evaluates (the rank's cosets of the quotient domain)
    const size_t j = (size_t)blockIdx.x * QW + lane;
    const bool active = j < size;
    const size_t jj = active ? j : 0;
    // leaf j = c*n + r  <->  natural row m = bitrev3(c) + 8*bitrev_n(r); next row m + 8 -> r' = bitrev_n(bitrev_n(r) + 1)
    // (buffers hold cosets from P.coset_first on: `cl` indexes memory, `c` is the coset of the evaluation point)
    const size_t cl = jj >> P.log_n, r = jj & (P.n - 1);
    const size_t c = cl + P.coset_first;
    const u32 rr = bitrev32((u32)r, P.log_n);
    const size_t rn = bitrev32((rr + 1) & (u32)(P.n - 1), P.log_n);
    const size_t jn = (cl << P.log_n) + rn;
    const u64 m = ((u64)rr << (P.log_N - P.log_n)) + bitrev32((u32)c, P.log_N - P.log_n);
    const u64 x = gl_mul(GL_GENERATOR, gl_mul(P.gN_lo[m & (((u64)1 << P.gN_h) - 1)], P.gN_hi[m >> P.gN_h]));
    const u64 z_last = gl_sub(x, P.g_inv);
    const u64 lag_first = P.lag_lde[jj], lag_last = P.lag_lde[P.N + jj];

    const u64* __restrict__ D = P.desc;
    const u32 n_ops = (u32)D[0], ops_off = (u32)D[1], nperm = (u32)D[2], perm_off = (u32)D[3], nctl = (u32)D[4], ctl_off = (u32)D[5];
    const u32 params_off = (u32)D[7];
    const u64 alpha0 = D[8], alpha1 = D[9];
    const u64 zh_inv = D[10 + c];
    u64 acc0 = 0, acc1 = 0;
    auto emit = [&](int kind, u64 v) {
        if (kind == AK_TRANSITION) v = gl_mul(v, z_last);
        else if (kind == AK_FIRST) v = gl_mul(v, lag_first);
        else if (kind == AK_LAST) v = gl_mul(v, lag_last);
        acc0 = gl_add(gl_mul(acc0, alpha0), v);
        acc1 = gl_add(gl_mul(acc1, alpha1), v);
    };
    const size_t N = P.N;

    // ---- the table's constraint program ----
    for (u32 i = 0; i < n_ops; i++) {
        const u64 w0 = D[ops_off + 2 * i], w1 = D[ops_off + 2 * i + 1];
        const int op = (int)(w0 & 0xff), kind = (int)((w0 >> 8) & 0xff);
        const u32 dst = (u32)((w0 >> 16) & 0xffff), a = (u32)((w0 >> 32) & 0xffff), b = (u32)((w0 >> 48) & 0xffff);
        u64 v;
        switch (op) {
            case AOP_LOCAL: v = gl_canon(P.trace_lde[(size_t)a * N + jj]); break;
            case AOP_NEXT: v = gl_canon(P.trace_lde[(size_t)a * N + jn]); break;
            case AOP_CONST: v = w1; break;
            case AOP_PARAM: v = D[params_off + a]; break;
            case AOP_ADD: v = gl_add(regs[a * QW + lane], regs[b * QW + lane]); break;
            case AOP_SUB: v = gl_sub(regs[a * QW + lane], regs[b * QW + lane]); break;
            case AOP_MUL: v = gl_mul(regs[a * QW + lane], regs[b * QW + lane]); break;
            case AOP_ISZERO: v = (regs[a * QW + lane] == 0) ? 1 : 0; break;
            default: emit(kind, regs[a * QW + lane]); continue;
        }
        regs[dst * QW + lane] = v;
    }
    // ---- permutation checks (permutation.rs:302-360) ----
    {
        for (u32 i = 0; i < nperm; i++) emit(AK_FIRST, gl_sub(P.zs_lde[(size_t)i * N + jj], 1));
        u32 p = perm_off;
        for (u32 bI = 0; bI < nperm; bI++) {
            const u32 ninst = (u32)D[p++];
            u64 prod_l = 1, prod_r = 1;
            for (u32 s = 0; s < ninst; s++) {
                const u64 beta = D[p++], gamma = D[p++];
                const u32 np = (u32)D[p++];
                u64 l = 0, rr2 = 0, w = 1;
                for (u32 k = 0; k < np; k++) {
                    const u64 cl = D[p++], cr = D[p++];
                    l = gl_add(l, gl_mul(gl_canon(P.trace_lde[cl * N + jj]), w));
                    rr2 = gl_add(rr2, gl_mul(gl_canon(P.trace_lde[cr * N + jj]), w));
                    w = gl_mul(w, beta);
                }
                prod_l = gl_mul(prod_l, gl_add(l, gamma));
                prod_r = gl_mul(prod_r, gl_add(rr2, gamma));
            }
            const u64 zl = P.zs_lde[(size_t)bI * N + jj], zn = P.zs_lde[(size_t)bI * N + jn];
            emit(AK_ALL, gl_sub(gl_mul(zn, prod_r), gl_mul(zl, prod_l)));
        }
    }
    // ---- cross-table lookup checks (cross_table_lookup.rs:380-421) ----
    {
        u32 p = ctl_off;
        for (u32 i = 0; i < nctl; i++) {
            const u64 beta = D[p++], gamma = D[p++];
            const u32 ncol = (u32)D[p++];
            u64 cl = 0, cn = 0, bp = 1;
            for (u32 k = 0; k < ncol; k++) {
                u32 p2 = p;
                const u64 el = dev_lincol(D, p, P.trace_lde, N, jj);
                const u64 en = dev_lincol(D, p2, P.trace_lde, N, jn);
                cl = gl_add(cl, gl_mul(bp, el));
                cn = gl_add(cn, gl_mul(bp, en));
                bp = gl_mul(bp, beta);
            }
            cl = gl_add(cl, gamma);
            cn = gl_add(cn, gamma);
            u64 fl = 1, fn = 1;
            if (D[p++]) {
                u32 p2 = p;
                fl = dev_lincol(D, p, P.trace_lde, N, jj);
                fn = dev_lincol(D, p2, P.trace_lde, N, jn);
            }
            // select(f, x) = f*x + 1 - f
            const u64 sl = gl_sub(gl_add(gl_mul(fl, cl), 1), fl), sn = gl_sub(gl_add(gl_mul(fn, cn), 1), fn);
            const u64 zl = P.zs_lde[(size_t)(nperm + i) * N + jj], zn = P.zs_lde[(size_t)(nperm + i) * N + jn];
            emit(AK_FIRST, gl_sub(zl, sl));
            emit(AK_TRANSITION, gl_sub(zn, gl_mul(zl, sn)));
        }
    }
    if (active) {
        P.out[j] = gl_mul(acc0, zh_inv);
        P.out[P.out_plane + j] = gl_mul(acc1, zh_inv);
    }
}

// ------------------------------------------------------------------------------------------------ specialised quotient kernels
// (airq.cuh; one translation unit per table signature under gen/, printed by olavm_amd/air/codegen.py)
#include "gen/air_registry.inc"

static const AirKernelEntry* find_air_kernel(u64 sig) {
    for (const AirKernelEntry* e : AIR_KERNELS)
        if (e->signature == sig) return e;
    return nullptr;
}

// any element of a[0,len) != b[0,len)?
__global__ __launch_bounds__(256) void any_diff_kernel(const u64* __restrict__ a, const u64* __restrict__ b, size_t len, unsigned* __restrict__ flag) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < len && a[k] != b[k]) atomicOr(flag, 1u);
}

// ------------------------------------------------------------------------------------------------ host orchestration
struct GpChallenge { u64 beta, gamma; };

static GpChallenge get_gp(OlaChallenger& ch) { const u64 b = challenger_get(ch); const u64 g = challenger_get(ch); return {b, g}; }

struct CtlJob { const HTwc* twc; GpChallenge ch; };

// Phase-level entry points (ola_perm_z / ola_ctl_z / ola_quotient) run prove_single_table with a tap: challenges come from the
// caller instead of the transcript, and the function hands back an intermediate result and stops.
struct PhaseTap {
    const u64* perm_challenges = nullptr;    // [batch_size][num_challenges][2] (beta, gamma)
    const u64* alphas = nullptr;             // [num_challenges]
    const OlaBatch* zs = nullptr;            // the caller's Z commitment (quotient phase: the Z columns are not recomputed)
    int stop_after = 0;                      // 1: after the Z columns, 2: after the quotient chunks
    std::vector<u64>* zs_out = nullptr;      // [nz][n] column-major
    std::vector<u64>* chunks_out = nullptr;  // [num_challenges * q][n] coefficients
    int nperm = 0, nz = 0, q = 0;
};

struct BatchHolder {
    DeviceCtx* ctx;
    OlaBatch* b = nullptr;
    explicit BatchHolder(DeviceCtx* c) : ctx(c) {}
    ~BatchHolder() { if (b) { (void)hipStreamSynchronize(ctx->stream); batch_destroy(ctx, b); } }
};

// values of one table resident on the device (column-major, n per column)
struct DevTable { u64* vals = nullptr; uint32_t log_n = 0; size_t n() const { return (size_t)1 << log_n; } };

static void write_cap(ByteWriter& w, const std::vector<u64>& cap) { w.cap(cap.data(), cap.size() / 4); }

// PolynomialBatch::from_values / from_coeffs for the whole table, or -- under the coset partition -- this rank's share of
// it; either way `cap_full` receives the complete Merkle cap (all-gathered from the ranks' slices when sharded).
static OlaBatch* commit_shared(DeviceCtx* ctx, NttTables& t, const u64* dev_cols, uint32_t ncols, uint32_t log_n, const OlaGpuConfig& cfg,
                               bool from_values, bool sharded, const ColumnFeed* feed, std::vector<u64>& cap_full, bool lean = false) {
    const size_t len_cap = (size_t)1 << cfg.cap_height;
    cap_full.assign(len_cap * 4, 0);
    if (!sharded) {
        if (ctx->acct.shardable) acct_exchange(ctx, len_cap * 32);   // the cap slices a partitioned run gathers
        OlaBatch* b = batch_commit(ctx, t, nullptr, dev_cols, ncols, log_n, cfg.rate_bits, cfg.cap_height, from_values, 0, 0, feed, lean);
        try { batch_read_cap(ctx, *b, cap_full.data()); } catch (...) { batch_destroy(ctx, b); throw; }
        return b;
    }
    OlaBatch* b = batch_commit(ctx, t, nullptr, dev_cols, ncols, log_n, cfg.rate_bits, cfg.cap_height, from_values, ctx->shard.rank,
                               ctx->shard.log_world, feed);
    u64* d_full = nullptr;
    try {
        const size_t len_local = len_cap >> ctx->shard.log_world;   // the local tree's cap level (batch_read_cap)
        d_full = (u64*)ctx->alloc(len_cap * 32);
        shard_all_gather(ctx, b->heap + 4 * len_local, d_full, len_local * 32);
        HIP_CHECK(hipMemcpyAsync(cap_full.data(), d_full, len_cap * 32, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->free(d_full);
    } catch (...) { if (d_full) ctx->free(d_full); batch_destroy(ctx, b); throw; }
    return b;
}

// Which tables run on the coset partition: every table large enough to be worth the exchanges.  The COMMITMENTS of a table
// split over the 2^rate_bits LDE cosets whatever its constraint degree; its quotient lives on the first 2^qdb cosets of the
// leaf order and is evaluated by the ranks that own them (all ranks when 2^qdb = 2^rate_bits: CPU, memory, Poseidon tables).
static bool table_is_sharded(const DeviceCtx* ctx, const OlaGpuConfig& cfg, const HTable& air, uint32_t log_n) {
    (void)air;
    if (ctx->shard.world <= 1) return false;
    return log_n >= ctx->shard.min_log_n && ctx->shard.log_world <= cfg.rate_bits && ctx->shard.log_world <= cfg.cap_height;
}

static void prove_single_table(DeviceCtx* ctx, NttTables& tables, const OlaGpuConfig& cfg, const HTable& air, const DevTable& tv,
                               const OlaBatch& trace_c, const std::vector<u64>& trace_cap, const std::vector<CtlJob>& ctl,
                               const u64* params, OlaChallenger& ch, std::vector<uint8_t>& bytes, bool sharded, PhaseTap* tap = nullptr,
                               PowDefer* pow_defer = nullptr) {
    DevBuf mem(ctx);
    const int nch = (int)cfg.num_challenges;
    if (nch != 2) throw OlaError(OLA_E_INVALID_ARG, "num_challenges must be 2");
    const int degree_bits = (int)tv.log_n;
    const size_t n = tv.n();
    const int rate_bits = (int)cfg.rate_bits;
    const size_t N = n << rate_bits;
    const size_t len_cap = (size_t)1 << cfg.cap_height;
    // this rank's part of the LDE domain (everything when not sharded)
    const uint32_t log_world = sharded ? ctx->shard.log_world : 0;
    const size_t coset_count = ((size_t)1 << rate_bits) >> log_world, coset_first = sharded ? ctx->shard.rank * coset_count : 0;
    const size_t N_loc = n * coset_count;

    challenger_compact(ch);
    // ---- permutation challenges + Z polys (prover.rs:360-377) ----
    const int nperm = air.num_permutation_batches(nch);
    const int bs = air.permutation_batch_size();
    std::vector<std::vector<GpChallenge>> perm_sets;
    if (!air.perm_pairs.empty())
        for (int i = 0; i < bs; i++) {
            std::vector<GpChallenge> s;
            for (int c = 0; c < nch; c++) {
                if (tap && tap->perm_challenges) s.push_back({gl_canon(tap->perm_challenges[(i * nch + c) * 2]), gl_canon(tap->perm_challenges[(i * nch + c) * 2 + 1])});
                else s.push_back(get_gp(ch));
            }
            perm_sets.push_back(s);
        }
    const bool have_zs = tap && tap->zs;     // quotient phase on a caller-held Z commitment: descriptors only
    const int nz = nperm + (int)ctl.size();
    if (nz == 0) throw OlaError(OLA_E_INVALID_ARG, "No CTL?");
    // OLA_TIMING scopes carry the reference's `timed!` names (prover.rs:374-544)
    // (the reference times only the permutation Z's, and only for tables that have permutation arguments, prover.rs:371-377; its
    // CTL Z's are computed untimed in cross_table_lookup_data, cross_table_lookup.rs:224-311 -- here both happen in this scope)
    std::unique_ptr<PhaseTimer> ph(new PhaseTimer(ctx, nperm > 0 ? "    compute permutation Z(x) polys" : "    compute CTL Z(x) polys"));
    u64* zvals = mem.alloc((size_t)nz * n);
    u64* tot = mem.alloc(pscan_tot_stride(n) * std::max<size_t>(1, ctl.size()));
    u64* tmpcol = mem.alloc(n);
    // perm descriptors (also reused by the quotient kernel)
    std::vector<u64> perm_desc;
    {
        const int total = (int)air.perm_pairs.size() * nch;
        int inst = 0;
        for (int b = 0; b < nperm; b++) {
            const size_t at = perm_desc.size();
            perm_desc.push_back(0);
            u64 cnt = 0;
            for (int i = 0; i < bs && inst < total; i++, inst++, cnt++) {
                const auto& pair = air.perm_pairs[inst / nch];
                const GpChallenge c = perm_sets[i][inst % nch];
                perm_desc.push_back(c.beta); perm_desc.push_back(c.gamma); perm_desc.push_back(pair.size());
                for (auto& pr : pair) { perm_desc.push_back(pr.first); perm_desc.push_back(pr.second); }
            }
            perm_desc[at] = cnt;
        }
    }
    if (nperm && !have_zs) {
        u64* d_pd = mem.alloc(perm_desc.size());
        HIP_CHECK(hipMemcpyAsync(d_pd, perm_desc.data(), perm_desc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        // offsets of each batch descriptor
        size_t off = 0;
        for (int b = 0; b < nperm; b++) {
            hipLaunchKernelGGL(perm_factor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, tv.vals, n, d_pd + off, tmpcol);
            product_scan_inclusive(ctx, tmpcol, n, tot);
            hipLaunchKernelGGL(shift_right_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, tmpcol, zvals + (size_t)b * n, n);
            // advance to the next batch descriptor
            const u64 cnt = perm_desc[off];
            size_t p = off + 1;
            for (u64 s = 0; s < cnt; s++) { const u64 np = perm_desc[p + 2]; p += 3 + 2 * np; }
            off = p;
        }
    }
    // ---- CTL Z polys (cross_table_lookup.rs:224-311) ----
    std::vector<u64> ctl_desc;
    std::vector<size_t> ctl_off;
    for (auto& j : ctl) { ctl_off.push_back(ctl_desc.size()); push_ctl_desc(ctl_desc, *j.twc, j.ch.beta, j.ch.gamma); }
    u64* d_cd = mem.alloc(ctl_desc.size() + 1);
    if (!ctl_desc.empty()) HIP_CHECK(hipMemcpyAsync(d_cd, ctl_desc.data(), ctl_desc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    unsigned* d_bad_filter = (unsigned*)mem.alloc(1);
    HIP_CHECK(hipMemsetAsync(d_bad_filter, 0, 8, ctx->stream));
    if (!ctl.empty() && !have_zs) {   // all CTL Z columns of the table in one launch per step
        std::vector<u64> offs(ctl_off.begin(), ctl_off.end());
        // the columns of one lookup side under the two challenges share their linear combinations: pair them (ctl_factor_kernel)
        std::vector<u64> pairs;
        {
            std::vector<char> taken(ctl.size(), 0);
            for (size_t a = 0; a < ctl.size(); a++) {
                if (taken[a]) continue;
                size_t b = a;
                for (size_t c = a + 1; c < ctl.size(); c++)
                    if (!taken[c] && ctl[c].twc == ctl[a].twc) { b = c; break; }
                taken[a] = taken[b] = 1;
                pairs.push_back(a); pairs.push_back(b);
            }
        }
        offs.insert(offs.end(), pairs.begin(), pairs.end());
        u64* d_offs = mem.upload(offs);                  // staged in the scope's keep-alive list: no synchronisation for its sake
        u64* d_pairs = d_offs + ctl_off.size();
        u64* zc = zvals + (size_t)nperm * n;
        hipLaunchKernelGGL(ctl_factor_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(pairs.size() / 2)), dim3(256), 0, ctx->stream, tv.vals, n, d_cd,
                           d_offs, d_pairs, zc, d_bad_filter);
        product_scan_inclusive(ctx, zc, n, tot, ctl.size());
    }
    // the filter flag travels with the next read-back the host waits for anyway (the Z commitment's cap)
    HostSpan h_flags = mem.host(2);
    h_flags[0] = h_flags[1] = 0;
    const bool check_filter = !ctl.empty() && !have_zs;
    if (check_filter) HIP_CHECK(hipMemcpyAsync(&h_flags[0], d_bad_filter, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (tap) { tap->nperm = nperm; tap->nz = nz; tap->q = air.quotient_degree_factor(); }
    if (tap && tap->stop_after == 1) {
        tap->zs_out->resize((size_t)nz * n);
        HIP_CHECK(hipMemcpyAsync(tap->zs_out->data(), zvals, (size_t)nz * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (check_filter && (unsigned)h_flags[0]) throw OlaError(OLA_E_INVALID_ARG, "Non-binary filter?");
        return;
    }
    // ---- Zs commitment ----
    ph.reset();
    ph.reset(new PhaseTimer(ctx, "    compute Zs commitment"));
    struct ZsRef { BatchHolder own; const OlaBatch* b = nullptr; explicit ZsRef(DeviceCtx* c) : own(c) {} } zs_c(ctx);
    std::vector<u64> zs_cap;
    const bool lean = trace_c.lean;   // the table is proven memory-lean: no LDE of its three batches is kept
    if (lean && sharded) throw OlaError(OLA_E_INTERNAL, "a memory-lean table cannot be on the coset partition");
    if (have_zs) {
        if (tap->zs->ncols != (uint32_t)nz || tap->zs->log_n != (uint32_t)degree_bits || tap->zs->lean != lean || tap->zs->is_shard())
            throw OlaError(OLA_E_INVALID_ARG, "Z commitment does not match the table");
        zs_c.b = tap->zs;
    } else {
        zs_c.own.b = commit_shared(ctx, tables, zvals, (uint32_t)nz, (uint32_t)degree_bits, cfg, true, sharded, nullptr, zs_cap, lean);
        zs_c.b = zs_c.own.b;
        if (check_filter && (unsigned)h_flags[0]) throw OlaError(OLA_E_INVALID_ARG, "Non-binary filter?");   // commit_shared has waited for the cap
        challenger_observe_cap(ch, zs_cap.data(), zs_cap.size() / 4);
    }
    const u64 alpha0 = (tap && tap->alphas) ? gl_canon(tap->alphas[0]) : challenger_get(ch);
    const u64 alpha1 = (tap && tap->alphas) ? gl_canon(tap->alphas[1]) : challenger_get(ch);

    // ---- quotient (prover.rs:571-705) ----
    ph.reset();
    ph.reset(new PhaseTimer(ctx, "    compute quotient polys"));
    const int q = air.quotient_degree_factor();
    int qdb = 0;
    while ((1 << qdb) < q) qdb++;
    if (qdb > rate_bits) throw OlaError(OLA_E_INVALID_ARG, "Having constraints of degree higher than the rate is not supported yet.");
    const size_t size = n << qdb;
    // Lagrange first/last on the LDE domain (leaf order)
    u64* lag_coef = mem.alloc(2 * n);
    u64* lag_lde = mem.alloc(2 * (lean ? n : N_loc));
    {
        TwoLevel gt = get_two(tables, degree_bits, 0);
        const u64 n_inv = gl_inv(((u64)1 << degree_bits) % GL_P);
        hipLaunchKernelGGL(lagrange_coeffs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lag_coef, n, n_inv, gt.lo, gt.hi, gt.h);
        WorkScope ws(ctx, rate_bits);
        if (!lean) ntt_lde_leaf_order(tables, lag_coef, lag_lde, degree_bits, rate_bits, 2, coset_first, coset_count);
    }
    // descriptor for the kernel
    std::vector<u64> desc(18, 0);
    {
        // Z_H(x)^-1 per coset (zero_poly_coset.rs:18-33): x^n = 7^n * v^i, i = natural index mod 2^qdb; coset c <-> i = bitrev
        u64 g_pow_n = GL_GENERATOR;
        for (int i = 0; i < degree_bits; i++) g_pow_n = gl_mul(g_pow_n, g_pow_n);
        const u64 v = gl_root_of_unity(qdb);
        for (int c = 0; c < (1 << qdb); c++) {
            const u32 i = bitrev32((u32)c, qdb);
            desc[10 + c] = gl_inv(gl_sub(gl_mul(g_pow_n, gl_pow(v, i)), 1));
        }
        desc[0] = air.ops.size() / 2;
        desc[1] = desc.size();
        desc.insert(desc.end(), air.ops.begin(), air.ops.end());
        desc[2] = (u64)nperm;
        desc[3] = desc.size();
        desc.insert(desc.end(), perm_desc.begin(), perm_desc.end());
        desc[4] = ctl.size();
        desc[5] = desc.size();
        desc.insert(desc.end(), ctl_desc.begin(), ctl_desc.end());
        desc[6] = (u64)air.n_params;
        desc[7] = desc.size();
        for (int i = 0; i < air.n_params; i++) desc.push_back(gl_canon(params[i]));
        desc[8] = alpha0; desc[9] = alpha1;
    }
    u64* d_desc = mem.alloc(desc.size());
    HIP_CHECK(hipMemcpyAsync(d_desc, desc.data(), desc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    u64* qv = mem.alloc(2 * size);
    // points this rank evaluates: the whole quotient domain, or (sharded) those of its cosets that belong to the quotient
    // domain -- the first 2^qdb cosets of the leaf order; a rank that owns none of them only takes part in the exchange
    const size_t plane = sharded ? N_loc : size;
    u64* qloc = sharded ? mem.alloc(2 * plane) : qv;
    const size_t q_cosets = (size_t)1 << qdb;
    const size_t my_q_cosets = !sharded ? q_cosets : (coset_first >= q_cosets ? 0 : std::min(coset_count, q_cosets - coset_first));
    {
        QuotParams P = {};
        P.trace_lde = trace_c.lde; P.zs_lde = zs_c.b->lde; P.lag_lde = lag_lde;
        P.N = N_loc; P.n = n; P.log_n = degree_bits; P.log_N = degree_bits + rate_bits; P.qdb = qdb;
        P.coset_first = (u32)coset_first; P.out_plane = plane;
        TwoLevel gN = get_two(tables, degree_bits + rate_bits, 0);
        P.gN_lo = gN.lo; P.gN_hi = gN.hi; P.gN_h = gN.h;
        P.desc = d_desc;
        P.g_inv = gl_inv(gl_root_of_unity(degree_bits));
        P.out = qloc;
        P.n_regs = air.n_regs;
        // OLA_AIR_KERNELS=interpreter forces the generic kernel, =crosscheck runs both and compares (tests)
        const char* mode = getenv("OLA_AIR_KERNELS");
        const bool force_interp = mode && !strcmp(mode, "interpreter"), crosscheck = mode && !strcmp(mode, "crosscheck");
        std::vector<const HTwc*> twcs;
        for (auto& jb : ctl) twcs.push_back(jb.twc);
        // the specialised kernels address rows as "workgroup base + lane": workgroups of min(AIRQ_THREADS, n) threads (airq.cuh)
        const AirKernelEntry* spec = force_interp ? nullptr : find_air_kernel(air_signature(air, twcs));
        const unsigned spec_bs = (unsigned)std::min<size_t>(AIRQ_THREADS, n);
        // memory-lean: the trace and Z values of one coset at a time, re-derived from the coefficients; the quotient of a table
        // with 2^qdb < 2^rate_bits cosets lives on the first 2^qdb cosets of the leaf order
        u64 *slice_t = nullptr, *slice_z = nullptr;
        const size_t lean_cosets = lean ? ((size_t)1 << qdb) : 1;
        if (lean) {
            slice_t = mem.alloc((size_t)trace_c.ncols * n);
            slice_z = mem.alloc((size_t)zs_c.b->ncols * n);
            P.trace_lde = slice_t; P.zs_lde = slice_z; P.N = n; P.npoints = n;
        }
        u64* d_sd = nullptr;
        int K = 0;
        if (spec) {
            K = spec->n_emits;
            std::vector<u64> sd(8 + 2 * (size_t)K, 0);
            for (int c = 0; c < (1 << qdb); c++) sd[c] = desc[10 + c];
            const u64 al[2] = {alpha0, alpha1};
            for (int c = 0; c < 2; c++) {
                u64 w = 1;
                for (int i = K - 1; i >= 0; i--) { sd[8 + (size_t)c * K + i] = w; w = gl_mul(w, al[c]); }
            }
            for (int i = 0; i < air.n_params; i++) sd.push_back(gl_canon(params[i]));
            for (int b = 0; b < nperm; b++)
                for (int i = 0; i < bs; i++)
                    for (int k = 0; k < 2; k++) {
                        const int inst = b * bs + i;
                        const bool live = inst < (int)air.perm_pairs.size() * nch;
                        sd.push_back(live ? (k ? perm_sets[i][inst % nch].gamma : perm_sets[i][inst % nch].beta) : 0);
                    }
            for (auto& jb : ctl) {
                sd.push_back(jb.ch.gamma);
                u64 bp = 1;
                for (size_t k = 0; k < jb.twc->columns.size(); k++) { sd.push_back(bp); bp = gl_mul(bp, jb.ch.beta); }
            }
            // limb forms of the uniform multipliers (airq.cuh Acc3), one slot per use in the kernel's own order
            auto push_limbs = [&sd](u64 w) {
                w = gl_canon(w);
                const u64 v = gl_mul(w, (u64)1 << 32), M = 0x3FFFFF;
                sd.push_back((w & M) | (((w >> 22) & M) << 32));
                sd.push_back((w >> 44) | ((v & M) << 32));
                sd.push_back(((v >> 22) & M) | ((v >> 44) << 32));
            };
            const size_t u64_words = sd.size();
            sd.reserve(u64_words + 3 * (size_t)spec->n_limbs);
            for (int sl = 0; sl < spec->n_limbs; sl++) {
                if ((size_t)spec->limb_src[sl] >= u64_words) throw OlaError(OLA_E_INVALID_ARG, "generated kernel and descriptor disagree");
                push_limbs(sd[(size_t)spec->limb_src[sl]]);
            }
            d_sd = mem.upload(sd);
        }
        u64* qv2 = (spec && crosscheck) ? mem.alloc(2 * plane) : nullptr;
        WorkScope ws(ctx, qdb);   // the quotient lives on 2^qdb cosets: that many ranks share it
        // units: LDE points evaluated; bytes a point streams -- local and next row of the trace and Z batches, two planes written
        // (quotient row streaming is priced against HBM)
        const double q_points = (double)(sharded ? my_q_cosets * n : size);
        PhaseScope phq(ctx, PH_QUOTIENT, q_points, q_points * (8.0 * 2 * ((double)trace_c.ncols + (double)zs_c.b->ncols) + 16.0));
        if (!sharded && ctx->acct.shardable) { acct_exchange(ctx, N * 8); acct_exchange(ctx, N * 8); }   // the two planes a partitioned run gathers
        for (size_t lc = 0; lc < lean_cosets; lc++) {
            u64* out = qloc;
            size_t points = sharded ? my_q_cosets * n : plane;
            if (sharded) { P.npoints = points; if (points == 0) break; }
            if (lean) {
                batch_lde_slice(ctx, tables, trace_c, lc, slice_t);
                batch_lde_slice(ctx, tables, *zs_c.b, lc, slice_z);
                ntt_lde_leaf_order(tables, lag_coef, lag_lde, degree_bits, rate_bits, 2, lc, 1);
                P.coset_first = (u32)lc;
                out = qloc + lc * n;
                points = n;
            }
            auto run_interp_on = [&](u64* o) {
                P.out = o;
                const size_t lds = (size_t)air.n_regs * QW * 8;
                if (lds > 160 * 1024) throw OlaError(OLA_E_INVALID_ARG, "constraint program needs too many registers");
                if (lds > 48 * 1024)
                    HIP_CHECK(hipFuncSetAttribute((const void*)quotient_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(quotient_kernel, dim3((unsigned)((points + QW - 1) / QW)), dim3(QW), lds, ctx->stream, P);
            };
            if (!spec) {
                run_interp_on(out);
            } else {
                QuotParams S = P;
                S.desc = d_sd;
                S.out = out;
                hipLaunchKernelGGL(spec->kernel, dim3((unsigned)((points + spec_bs - 1) / spec_bs)), dim3(spec_bs), 0, ctx->stream, S);
                if (crosscheck) run_interp_on(qv2 + (lean ? lc * n : 0));
            }
        }
        if (spec && crosscheck) {
            unsigned* d_flag = (unsigned*)mem.alloc(1);
            HIP_CHECK(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
            const size_t evaluated = sharded ? my_q_cosets * n : plane;       // per plane
            for (int c = 0; c < 2 && evaluated; c++)
                hipLaunchKernelGGL(any_diff_kernel, dim3((unsigned)((evaluated + 255) / 256)), dim3(256), 0, ctx->stream, qloc + (size_t)c * plane,
                                   qv2 + (size_t)c * plane, evaluated, d_flag);
            unsigned flag = 0;
            HIP_CHECK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (flag) throw OlaError(OLA_E_INTERNAL, std::string("specialised quotient kernel disagrees with the interpreter: ") + spec->name);
        }
    }
    if (sharded) {
        // every rank needs the whole quotient for the inverse transform: gather the ranks' planes (rank order = leaf order) and
        // keep the first n * 2^qdb values of each
        if (N == size) {
            for (int c = 0; c < 2; c++) shard_all_gather(ctx, qloc + (size_t)c * plane, qv + (size_t)c * size, plane * 8);
        } else {
            u64* gath = mem.alloc(N);
            for (int c = 0; c < 2; c++) {
                shard_all_gather(ctx, qloc + (size_t)c * plane, gath, plane * 8);
                HIP_CHECK(hipMemcpyAsync(qv + (size_t)c * size, gath, size * 8, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
    }
    // qv is in bit-reversed order of the size-domain: un-reverse, coset iNTT (prover.rs:700-704)
    const int size_bits = degree_bits + qdb;
    u64* qnat = mem.alloc(2 * size);
    u64* qcoef = mem.alloc(2 * size);
    u64* qscratch = size_bits > 13 ? mem.alloc(2 * size) : nullptr;
    launch_bitrev_rows(ctx, qv, qnat, size_bits, 2);
    ntt_coset_interpolate(tables, qnat, qcoef, qscratch, size_bits, 2, GL_GENERATOR);
    // trim_to_len(n*q) (prover.rs:469-473)
    const size_t keep = n * (size_t)q;
    if (keep < size) {
        unsigned* d_flag = (unsigned*)mem.alloc(1);
        HIP_CHECK(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
        for (int c = 0; c < 2; c++)
            hipLaunchKernelGGL(any_nonzero_kernel, dim3((unsigned)((size - keep + 255) / 256)), dim3(256), 0, ctx->stream, qcoef + (size_t)c * size,
                               keep, size, d_flag);
        // read with the next read-back the host waits for (the quotient commitment's cap, or the tap's copy): no wait of its own
        HIP_CHECK(hipMemcpyAsync(&h_flags[1], d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    auto check_degree = [&] {
#ifndef AIRQ_TIMING_ONLY_L2_LOADS      // (timing builds of airq.cuh's experiment produce garbage on purpose)
        if ((unsigned)h_flags[1] && !ctx->priming) throw OlaError(OLA_E_QUOTIENT_DEGREE, "Quotient has failed, the vanishing polynomial is not divisible by Z_H");
#endif
    };
    // chunks of n coefficients: [challenge][k] -> column challenge*q + k
    ph.reset();
    ph.reset(new PhaseTimer(ctx, "    split quotient polys"));
    u64* chunks = mem.alloc((size_t)2 * q * n);
    for (int c = 0; c < 2; c++)
        HIP_CHECK(hipMemcpyAsync(chunks + (size_t)c * q * n, qcoef + (size_t)c * size, keep * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (tap && tap->stop_after == 2) {
        tap->chunks_out->resize((size_t)2 * q * n);
        HIP_CHECK(hipMemcpyAsync(tap->chunks_out->data(), chunks, (size_t)2 * q * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        check_degree();
        return;
    }
    ph.reset();
    ph.reset(new PhaseTimer(ctx, "    compute quotient commitment"));
    BatchHolder q_c(ctx);
    std::vector<u64> q_cap;
    q_c.b = commit_shared(ctx, tables, chunks, (uint32_t)(2 * q), (uint32_t)degree_bits, cfg, false, sharded, nullptr, q_cap, lean);
    check_degree();                                      // commit_shared has waited for the cap
    challenger_observe_cap(ch, q_cap.data(), q_cap.size() / 4);

    // ---- write_proof (serialization.rs:349-358): caps, then opening set + FRI proof ----
    ByteWriter w{bytes, ctx->hasher == (int)OLA_HASH_BLAKE3};
    write_cap(w, trace_cap);
    write_cap(w, zs_cap);
    write_cap(w, q_cap);
    size_t olen = 0;
    ph.reset();
    ph.reset(new PhaseTimer(ctx, "    compute openings proof"));
    open_and_prove(ctx, tables, cfg, trace_c, *zs_c.b, *q_c.b, (uint32_t)nperm, ch, bytes, olen, pow_defer);
}

// Per table, the running products it must carry: one per (lookup it takes part in, challenge), looking sides before the
// looked side, lookups in declaration order (cross_table_lookup_data, cross_table_lookup.rs:224-311).
static std::vector<std::vector<CtlJob>> ctl_jobs(const HAirSet& set, const std::vector<GpChallenge>& ctl_ch) {
    std::vector<std::vector<CtlJob>> jobs(set.tables.size());
    for (const HCtl& ctl : set.ctls)
        for (const GpChallenge& c : ctl_ch) {
            for (const HTwc& twc : ctl.looking) jobs[twc.table].push_back({&twc, c});
            jobs[ctl.looked.table].push_back({&ctl.looked, c});
        }
    return jobs;
}

// One table's StarkProof with the trace commitment and the transcript supplied by the caller (ola_prove_single_table).
void prove_single_table_host(DeviceCtx* ctx, NttTables& tables, const OlaGpuConfig& cfg, const u64* airset, size_t airset_words,
                             uint32_t table, const u64* const* trace_cols, const OlaBatch& trace_c, const u64* trace_cap,
                             const u64* ctl_challenges, const u64* params, OlaChallenger& ch, std::vector<uint8_t>& bytes) {
    HAirSet set = parse_airset(airset, airset_words);
    if (table >= set.tables.size()) throw OlaError(OLA_E_INVALID_ARG, "table index out of range");
    if (ctx->shard.world != 1) throw OlaError(OLA_E_INVALID_ARG, "per-table proving needs an unsharded context");
    const HTable& air = set.tables[table];
    if (trace_c.ncols != (uint32_t)air.ncols || trace_c.is_shard()) throw OlaError(OLA_E_INVALID_ARG, "trace commitment does not match the table");
    std::vector<GpChallenge> ctl_ch;
    for (uint32_t c = 0; c < cfg.num_challenges; c++) ctl_ch.push_back({ctl_challenges[2 * c], ctl_challenges[2 * c + 1]});
    const std::vector<std::vector<CtlJob>> jobs = ctl_jobs(set, ctl_ch);
    DevBuf mem(ctx);
    DevTable tv;
    tv.log_n = trace_c.log_n;
    const size_t n = tv.n();
    tv.vals = mem.alloc((size_t)air.ncols * n);
    for (int c = 0; c < air.ncols; c++)
        HIP_CHECK(hipMemcpyAsync(tv.vals + (size_t)c * n, trace_cols[c], n * 8, hipMemcpyHostToDevice, ctx->stream));
    canonicalize(ctx, tv.vals, (size_t)air.ncols * n);
    const size_t len_cap = (size_t)1 << cfg.cap_height;
    const std::vector<u64> cap(trace_cap, trace_cap + 4 * len_cap);
    std::vector<u64> zero_params(64, 0);
    if (!params && air.n_params > 64) throw OlaError(OLA_E_INVALID_ARG, "params required");
    prove_single_table(ctx, tables, cfg, air, tv, trace_c, cap, jobs[table], params ? params : zero_params.data(), ch, bytes, false);
}

// Widths of a table's three batches (trace, Z, quotient chunks)
struct TableWidths { size_t w, wz, wq; int qdb; };
static TableWidths table_widths(const HAirSet& set, size_t t, int nch) {
    const HTable& air = set.tables[t];
    size_t nctl = 0;
    for (const HCtl& c : set.ctls) { for (const HTwc& w2 : c.looking) nctl += ((size_t)w2.table == t) ? nch : 0; nctl += ((size_t)c.looked.table == t) ? nch : 0; }
    TableWidths r;
    r.w = (size_t)air.ncols; r.wz = air.num_permutation_batches(nch) + nctl; r.wq = (size_t)nch * air.quotient_degree_factor();
    r.qdb = 0;
    while ((1 << r.qdb) < air.quotient_degree_factor()) r.qdb++;
    return r;
}

// ---- phase-level entry points (one `timed!` scope of the reference at a time) --------------------------------
// out[6] = ncols, n_params, permutation Z columns, CTL Z columns, quotient_degree_factor, permutation batch size
void table_shape_host(const OlaGpuConfig& cfg, const u64* airset, size_t airset_words, uint32_t table, uint32_t out[6]) {
    HAirSet set = parse_airset(airset, airset_words);
    if (table >= set.tables.size()) throw OlaError(OLA_E_INVALID_ARG, "table index out of range");
    const TableWidths tw = table_widths(set, table, (int)cfg.num_challenges);
    const HTable& air = set.tables[table];
    const uint32_t nperm = (uint32_t)air.num_permutation_batches((int)cfg.num_challenges);
    out[0] = (uint32_t)air.ncols; out[1] = (uint32_t)air.n_params; out[2] = nperm; out[3] = (uint32_t)tw.wz - nperm;
    out[4] = (uint32_t)air.quotient_degree_factor(); out[5] = (uint32_t)air.permutation_batch_size();
}

// compute_permutation_z_polys (permutation.rs:103-187) and the table's share of cross_table_lookup_data
// (cross_table_lookup.rs:224-311): the Z columns as values over the trace domain.  which = 0: permutation columns, 1: CTL columns.
void phase_zs_host(DeviceCtx* ctx, NttTables& tables, const OlaGpuConfig& cfg, const u64* airset, size_t airset_words, uint32_t table,
                   uint32_t log_n, const u64* const* trace_cols, const u64* perm_challenges, const u64* ctl_challenges, int which,
                   std::vector<u64>& out_cols) {
    HAirSet set = parse_airset(airset, airset_words);
    if (table >= set.tables.size()) throw OlaError(OLA_E_INVALID_ARG, "table index out of range");
    const HTable& air = set.tables[table];
    std::vector<GpChallenge> ctl_ch;
    for (uint32_t c = 0; c < cfg.num_challenges; c++) ctl_ch.push_back(ctl_challenges ? GpChallenge{gl_canon(ctl_challenges[2 * c]), gl_canon(ctl_challenges[2 * c + 1])} : GpChallenge{1, 1});
    const std::vector<std::vector<CtlJob>> jobs = ctl_jobs(set, ctl_ch);
    DevBuf mem(ctx);
    DevTable tv;
    tv.log_n = log_n;
    const size_t n = tv.n();
    tv.vals = mem.alloc((size_t)air.ncols * n);
    for (int c = 0; c < air.ncols; c++)
        HIP_CHECK(hipMemcpyAsync(tv.vals + (size_t)c * n, trace_cols[c], n * 8, hipMemcpyHostToDevice, ctx->stream));
    canonicalize(ctx, tv.vals, (size_t)air.ncols * n);
    std::vector<u64> dummy_perm((size_t)air.permutation_batch_size() * cfg.num_challenges * 2, 1);
    PhaseTap tap;
    tap.perm_challenges = perm_challenges ? perm_challenges : dummy_perm.data();
    tap.stop_after = 1;
    std::vector<u64> zs;
    tap.zs_out = &zs;
    OlaBatch shape;            // only its flags are looked at before the Z columns exist
    shape.ncols = (uint32_t)air.ncols; shape.log_n = log_n;
    OlaChallenger ch;
    challenger_init(ch, (uint32_t)ctx->hasher);
    std::vector<uint8_t> bytes;
    std::vector<u64> zero_params(64, 0), cap;
    prove_single_table(ctx, tables, cfg, air, tv, shape, cap, jobs[table], zero_params.data(), ch, bytes, false, &tap);
    const size_t first = which == 0 ? 0 : (size_t)tap.nperm, count = which == 0 ? (size_t)tap.nperm : (size_t)(tap.nz - tap.nperm);
    out_cols.assign(zs.begin() + first * n, zs.begin() + (first + count) * n);
}

// compute_quotient_polys + the split into degree-n chunks (prover.rs:441-480, 571-705): coefficients of the 2 * q chunk
// polynomials, ready for PolynomialBatch::from_coeffs.
void phase_quotient_host(DeviceCtx* ctx, NttTables& tables, const OlaGpuConfig& cfg, const u64* airset, size_t airset_words, uint32_t table,
                         const OlaBatch& trace_c, const OlaBatch& zs_c, const u64* perm_challenges, const u64* ctl_challenges, const u64* alphas,
                         const u64* params, std::vector<u64>& chunks) {
    HAirSet set = parse_airset(airset, airset_words);
    if (table >= set.tables.size()) throw OlaError(OLA_E_INVALID_ARG, "table index out of range");
    if (ctx->shard.world != 1) throw OlaError(OLA_E_INVALID_ARG, "per-phase proving needs an unsharded context");
    const HTable& air = set.tables[table];
    if (trace_c.ncols != (uint32_t)air.ncols || trace_c.is_shard()) throw OlaError(OLA_E_INVALID_ARG, "trace commitment does not match the table");
    if (!air.perm_pairs.empty() && !perm_challenges) throw OlaError(OLA_E_INVALID_ARG, "the table has permutation arguments: perm_challenges required");
    std::vector<GpChallenge> ctl_ch;
    for (uint32_t c = 0; c < cfg.num_challenges; c++) ctl_ch.push_back({gl_canon(ctl_challenges[2 * c]), gl_canon(ctl_challenges[2 * c + 1])});
    const std::vector<std::vector<CtlJob>> jobs = ctl_jobs(set, ctl_ch);
    DevTable tv;
    tv.log_n = trace_c.log_n;
    PhaseTap tap;
    tap.perm_challenges = perm_challenges; tap.alphas = alphas; tap.zs = &zs_c; tap.stop_after = 2; tap.chunks_out = &chunks;
    OlaChallenger ch;
    challenger_init(ch, (uint32_t)ctx->hasher);
    std::vector<uint8_t> bytes;
    std::vector<u64> zero_params(64, 0), cap;
    if (!params && air.n_params > 64) throw OlaError(OLA_E_INVALID_ARG, "params required");
    prove_single_table(ctx, tables, cfg, air, tv, trace_c, cap, jobs[table], params ? params : zero_params.data(), ch, bytes, false, &tap);
}

// Memory-lean tables (OLA_LEAN=1 forces, =0 forbids, default: when keeping every LDE resident would not fit): the LDEs of the
// large tables are streamed coset by coset instead of kept (batch_commit lean) -- a 2^24-row CPU table then proves on one GPU
// (fri/oracle.rs:66-99 holds all of it; the reference's GPU shim was sized for 2^24, cfft/ntt/mod.rs:13).
static std::vector<char> plan_lean_tables(DeviceCtx* ctx, const OlaGpuConfig& cfg, const HAirSet& set, const uint32_t* log_n) {
    const size_t nt = set.tables.size();
    const int nch = (int)cfg.num_challenges;
    std::vector<char> lean(nt, 0);
    const char* e = getenv("OLA_LEAN");
    const int mode = e ? atoi(e) : -1;
    size_t need = 0;
    for (size_t t = 0; t < nt; t++) {
        const TableWidths tw = table_widths(set, t, nch);
        const size_t n_t = (size_t)1 << log_n[t], w_all = tw.w + tw.wz + tw.wq;
        need += w_all * n_t * 8 * ((size_t)1 << cfg.rate_bits) + 3 * 2 * (n_t << cfg.rate_bits) * 32;   // LDEs + digest heaps
        need += (tw.w + w_all) * n_t * 8;                                                                // + values + coefficients
    }
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // what the pool holds is available to the proof; blocks the reserver has not delivered yet are still part of free_b
    size_t cached;
    { std::lock_guard<std::mutex> lk(ctx->mu); cached = ctx->cached_bytes; }
    const size_t budget = (size_t)(0.80 * (double)(free_b + cached));
    const bool want = mode == 1 || (mode != 0 && need > budget);
    if (want && ctx->shard.world <= 1)
        for (size_t t = 0; t < nt; t++) lean[t] = (mode == 1) ? (log_n[t] >= 1) : (log_n[t] >= 20);
    if (ctx->timing) fprintf(stderr, "[ola-timing] resident proof would need about %.1f GB, %.1f GB available: %s\n", need / 1e9, budget / 1e9, want ? "memory-lean (coset-streamed) large tables" : "all LDEs resident");
    return lean;
}

// ola_gpu_reserve: the large device buffers prove_with_traces will ask for with these table heights, allocated by a helper
// thread into the context's pool while the caller is still busy elsewhere (reading the trace file, say).
void reserve_for_proof(DeviceCtx* ctx, const OlaGpuConfig& cfg, const u64* airset, size_t airset_words, const uint32_t* log_n) {
    HAirSet set = parse_airset(airset, airset_words);
    const size_t nt = set.tables.size();
    const int nch = (int)cfg.num_challenges;
    const std::vector<char> lean = plan_lean_tables(ctx, cfg, set, log_n);
    std::vector<size_t> sizes;
    size_t big = 0;
    for (size_t t = 0; t < nt; t++) if (log_n[t] > log_n[big]) big = t;
    auto batch_blocks = [&](size_t w, size_t n, bool is_lean) {
        const size_t N = n << cfg.rate_bits;
        sizes.push_back(w * n * 8);                      // coefficients
        sizes.push_back(2 * N * 32);                     // digest heap
        if (!is_lean) sizes.push_back(w * N * 8);        // LDE
    };
    // in the order the proof asks for them: the trace values of every large table (allocated before the upload starts), then the
    // commitment of each large table from the largest down (the order prove_with_traces commits in)
    std::vector<size_t> large;
    for (size_t t = 0; t < nt; t++)
        if (table_widths(set, t, nch).w * ((size_t)1 << log_n[t]) * 8 >= (64u << 20)) large.push_back(t);   // small tables allocate in microseconds
    for (size_t t : large) sizes.push_back(table_widths(set, t, nch).w * ((size_t)1 << log_n[t]) * 8);       // the trace values
    std::stable_sort(large.begin(), large.end(), [&](size_t a, size_t b) {
        return table_widths(set, a, nch).w << log_n[a] > table_widths(set, b, nch).w << log_n[b]; });
    for (size_t t : large) {
        const TableWidths tw = table_widths(set, t, nch);
        const size_t n = (size_t)1 << log_n[t];
        batch_blocks(tw.w, n, lean[t] != 0);
        if (lean[t]) sizes.push_back(tw.w * n * 8);      // the coset being hashed / transform scratch
    }
    // Z and quotient buffers go back to the pool when a table is done and are handed to the next table that asks for a size they
    // fit (alloc's rule: at most a quarter larger than wanted): reserve, in proving order, what no earlier table leaves behind
    (void)big;
    std::vector<size_t> left_behind;
    std::sort(large.begin(), large.end());
    for (size_t t : large) {
        const TableWidths tw = table_widths(set, t, nch);
        const size_t n = (size_t)1 << log_n[t], size = n << tw.qdb;
        std::vector<size_t> mine;
        { std::vector<size_t> keep; keep.swap(sizes);
          sizes.push_back(tw.wz * n * 8);              // Z values
          batch_blocks(tw.wz, n, lean[t] != 0);
          for (int i = 0; i < 4; i++) sizes.push_back(2 * size * 8);   // quotient values, natural order, coefficients, scratch
          sizes.push_back(tw.wq * n * 8);              // chunks
          batch_blocks(tw.wq, n, lean[t] != 0);
          if (lean[t]) { sizes.push_back(tw.w * n * 8); sizes.push_back(tw.wz * n * 8); }
          mine.swap(sizes); sizes.swap(keep); }
        std::vector<size_t> avail = left_behind;
        for (size_t raw : mine) {
            const size_t want = DeviceCtx::round_size(raw);
            if (want < (32u << 20)) continue;
            size_t best = avail.size();
            for (size_t i = 0; i < avail.size(); i++)
                if (avail[i] >= want && avail[i] <= want + want / 4 && (best == avail.size() || avail[i] < avail[best])) best = i;
            if (best < avail.size()) { avail.erase(avail.begin() + (long)best); continue; }
            sizes.push_back(raw);
            left_behind.push_back(want);
        }
    }
    ctx->reserve_async(sizes);                            // delivered in this order
}

// prove_with_traces (prover.rs:79-327).  traces[t]: where table t's columns are -- one column-major ncols x 2^log_n[t] block, or one
// pointer per column (the reference's Vec<PolynomialValues<F>>, prover.rs:79-83) -- see upload.h.
void prove_with_traces(DeviceCtx* ctx, NttTables& tables, const OlaGpuConfig& cfg, const u64* airset, size_t airset_words,
                       const TraceSource* traces, const uint32_t* log_n, const u64* params, const u64* compress,
                       std::vector<uint8_t>& bytes) {
    HAirSet set = parse_airset(airset, airset_words);
    const size_t nt = set.tables.size();
    const int nch = (int)cfg.num_challenges;
    const size_t len_cap = (size_t)1 << cfg.cap_height;
    DevBuf mem(ctx);
    std::vector<DevTable> dev(nt);
    std::vector<std::unique_ptr<BatchHolder>> commits;
    std::vector<std::vector<u64>> caps(nt, std::vector<u64>(len_cap * 4));
    OlaChallenger ch;
    challenger_init(ch, (uint32_t)ctx->hasher);
    PhaseTimer t_all(ctx, "prove_with_traces total");
    // Coset partition: a sharded table's columns are uploaded 1/world per rank and all-gathered device to device (xGMI instead of
    // `world` copies of the trace over the host's PCIe links); cpr = columns per rank, the last ranks may hold fewer or none.
    const uint32_t world = ctx->shard.world, rank = ctx->shard.rank;
    std::vector<uint32_t> cpr(nt, 0);
    for (size_t t = 0; t < nt; t++) {
        if (log_n[t] + cfg.rate_bits > 32 || log_n[t] + cfg.rate_bits < cfg.cap_height) throw OlaError(OLA_E_INVALID_ARG, "table size out of range");
        dev[t].log_n = log_n[t];
        const uint32_t w = (uint32_t)set.tables[t].ncols;
        if (table_is_sharded(ctx, cfg, set.tables[t], log_n[t])) cpr[t] = (w + world - 1) / world;
        dev[t].vals = mem.alloc((size_t)(cpr[t] ? cpr[t] * world : w) << log_n[t]);
    }
    // The traces are pageable host memory: the uploader's threads push them to the device through a pinned staging ring on
    // their own stream while this thread already interpolates / extends / hashes what has arrived (upload.h).
    const std::vector<char> lean = plan_lean_tables(ctx, cfg, set, log_n);
    TraceUploader up(ctx, nt);
    std::vector<uint32_t> own_cols(nt, 0);
    for (size_t t = 0; t < nt; t++) {
        const size_t n_t = (size_t)1 << log_n[t];
        const uint32_t w = (uint32_t)set.tables[t].ncols;
        if (cpr[t]) {
            const uint32_t c0 = std::min(w, rank * cpr[t]), c1 = std::min(w, (rank + 1) * cpr[t]);
            own_cols[t] = c1 - c0;
            up.add(t, traces[t], c0, dev[t].vals + (size_t)c0 * n_t, own_cols[t], n_t);
        } else {
            up.add(t, traces[t], 0, dev[t].vals, w, n_t);
        }
    }
    // Order of upload and commitment (the caps enter the transcript in table order afterwards, prover.rs:139-142): the tables
    // below 64 MB first -- their commitments then run in the shadow of the large upload instead of after it --, then the large
    // ones from the largest down: what is left to do when the last byte arrives is one column group and the Merkle tree of
    // the smallest large table, and the largest tree (190 ms for the CPU table under Poseidon) overlaps the rest of the upload.
    std::vector<size_t> order(nt);
    for (size_t t = 0; t < nt; t++) order[t] = t;
    {
        auto bytes_of = [&](size_t t) { return ((size_t)set.tables[t].ncols << log_n[t]) * 8; };
        const size_t big = (size_t)64 << 20;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            const bool la = bytes_of(a) >= big, lb = bytes_of(b) >= big;
            if (la != lb) return lb;                       // small ones first
            return la && bytes_of(a) > bytes_of(b);        // large: descending; small: as numbered
        });
    }
    up.set_order(order);
    up.start();
    std::unique_ptr<PhaseTimer> t_commit(new PhaseTimer(ctx, "compute trace commitments"));
    commits.resize(nt);
    for (size_t t : order) {
        ctx->scopes.table = (int)t;
        PhaseTimer tt(ctx, "  table " + std::to_string(t) + " trace commitment (upload overlapped)");
        const size_t n_t = (size_t)1 << log_n[t];
        ctx->acct.shardable = log_n[t] >= ctx->shard.min_log_n;
        if (ctx->acct.shardable && !cpr[t]) acct_exchange(ctx, (size_t)set.tables[t].ncols * n_t * 8);   // the trace values a partitioned run gathers
        ColumnFeed feed;
        feed.chunk_cols = up.chunk_cols(t);
        feed.before_chunk = [&, t, n_t](uint32_t c0, uint32_t c1) {
            up.wait(t, c1);
            // columns that crossed the link as 32-bit words are canonical as they arrive: only runs of the others are reduced
            for (uint32_t c = c0; c < c1;) {
                if (up.column_is_narrow(t, c)) { c++; continue; }
                uint32_t e = c + 1;
                while (e < c1 && !up.column_is_narrow(t, e)) e++;
                canonicalize(ctx, dev[t].vals + (size_t)c * n_t, (size_t)(e - c) * n_t);
                c = e;
            }
        };
        commits[t].reset(new BatchHolder(ctx));
        if (cpr[t]) {
            // this rank's columns have to be on the device, then every rank receives everybody's
            up.wait(t, own_cols[t]);
            const size_t blk = (size_t)cpr[t] * n_t;
            DevBuf tmp(ctx);
            u64* send = tmp.alloc(blk);
            HIP_CHECK(hipMemcpyAsync(send, dev[t].vals + (size_t)rank * blk, blk * 8, hipMemcpyDeviceToDevice, ctx->stream));
            shard_all_gather(ctx, send, dev[t].vals, blk * 8);
            canonicalize(ctx, dev[t].vals, (size_t)set.tables[t].ncols * n_t);
            commits[t]->b = commit_shared(ctx, tables, dev[t].vals, (uint32_t)set.tables[t].ncols, log_n[t], cfg, true, true, nullptr, caps[t], false);
            continue;
        }
        commits[t]->b = commit_shared(ctx, tables, dev[t].vals, (uint32_t)set.tables[t].ncols, log_n[t], cfg, true,
                                      table_is_sharded(ctx, cfg, set.tables[t], log_n[t]), &feed, caps[t], lean[t] != 0);
    }
    up.finish();
    ctx->scopes.table = -1;
    t_commit.reset();
    for (size_t t = 0; t < nt; t++) challenger_observe_cap(ch, caps[t].data(), caps[t].size() / 4);
    // CTL challenges and per-table job lists, in cross_table_lookup_data order
    std::vector<GpChallenge> ctl_ch;
    for (int c = 0; c < nch; c++) ctl_ch.push_back(get_gp(ch));
    const std::vector<std::vector<CtlJob>> jobs = ctl_jobs(set, ctl_ch);
    ByteWriter w{bytes, ctx->hasher == (int)OLA_HASH_BLAKE3};
    w.u32((uint32_t)nt);
    size_t poff = 0;
    std::vector<u64> zero_params(64, 0);
    // the proof-of-work searches of the tables run on the side stream and are collected at the end (fri.hip PowDefer); the
    // slots are this scope's, allocated before any inner scope exists (the pinned arena is a stack).  OLA_POW_DEFER=0: in line.
    PowDefer pow;
    static const bool pow_defer_on = [] { const char* e = getenv("OLA_POW_DEFER"); return !(e && *e == '0'); }();
    if (pow_defer_on) { pow.slots = (unsigned long long*)ctx->pinned_alloc(nt * 8); pow.nslots = pow.slots ? nt : 0; }
    struct PowGuard { DeviceCtx* c; PowDefer& p; ~PowGuard() { pow_abandon(c, p); } } pow_guard{ctx, pow};
    for (size_t t = 0; t < nt; t++) {
        const u64* pr = params ? params + poff : zero_params.data();
        if (!params && set.tables[t].n_params > 64) throw OlaError(OLA_E_INVALID_ARG, "params required");
        poff += set.tables[t].n_params;
        ctx->scopes.table = (int)t;
        PhaseTimer tt(ctx, "  table " + std::to_string(t) + " prove_single_table");
        ctx->acct.shardable = log_n[t] >= ctx->shard.min_log_n;
        prove_single_table(ctx, tables, cfg, set.tables[t], dev[t], *commits[t]->b, caps[t], jobs[t], pr, ch, bytes,
                           table_is_sharded(ctx, cfg, set.tables[t], log_n[t]), nullptr, &pow);
    }
    pow_finish(ctx, pow, bytes);
    ctx->acct.shardable = false;
    // compress_challenges (prover.rs:307-320) -- produced by trace generation, carried through
    w.u32((uint32_t)nt);
    for (size_t t = 0; t < nt; t++) w.field(compress ? compress[t] : 0);
    if (ctx->timing && ctx->adopted_blocks) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        fprintf(stderr, "[ola-timing] device allocator: %zu block(s), %.1f GB, taken over from idle contexts on this GPU so far\n", ctx->adopted_blocks, ctx->adopted_bytes / 1e9);
    }
}

// column counts of the tables of an AIR set (the whole-proof entry points validate their pointer arrays with it)
std::vector<size_t> airset_widths(const u64* airset, size_t airset_words) {
    HAirSet set = parse_airset(airset, airset_words);
    std::vector<size_t> w;
    for (const HTable& t : set.tables) w.push_back((size_t)t.ncols);
    return w;
}

// ola_air_kernels_available
void air_kernels_available(const u64* airset, size_t airset_words, uint8_t* has_kernel, size_t ntables) {
    HAirSet set = parse_airset(airset, airset_words);
    if (ntables != set.tables.size()) throw OlaError(OLA_E_INVALID_ARG, "ntables does not match the AIR set");
    std::vector<std::vector<const HTwc*>> jobs(set.tables.size());
    for (const HCtl& ctl : set.ctls)
        for (int c = 0; c < 2; c++) {
            for (const HTwc& twc : ctl.looking) jobs[twc.table].push_back(&twc);
            jobs[ctl.looked.table].push_back(&ctl.looked);
        }
    for (size_t t = 0; t < set.tables.size(); t++) has_kernel[t] = find_air_kernel(air_signature(set.tables[t], jobs[t])) ? 1 : 0;
}

}  // namespace ola
