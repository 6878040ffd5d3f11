// ORACLE -- test infrastructure only (see gl.hpp header).  Plain C entry points over the CPU restatement so
// that tests/ (ctypes + numpy) and bench.py's cpu_baseline leg can drive it.  Not part of the product.
#include <cstdlib>
#include <cstring>

#include "oracle.hpp"

using namespace ola_oracle;

namespace {
struct Reader {
    const uint8_t* p; size_t n, off = 0; bool ok = true;
    uint8_t u8() { if (off + 1 > n) { ok = false; return 0; } return p[off++]; }
    uint32_t u32() { uint32_t x = 0; for (int i = 0; i < 4; i++) x |= (uint32_t)u8() << (8 * i); return x; }
    u64 field() { u64 x = 0; for (int i = 0; i < 8; i++) x |= (u64)u8() << (8 * i); return x; }
    Ext2 ext() { u64 a = field(); u64 b = field(); return Ext2{a, b}; }
    std::vector<u64> field_vec() { uint32_t l = u32(); std::vector<u64> v; for (uint32_t i = 0; i < l && ok; i++) v.push_back(field()); return v; }
    std::vector<Ext2> ext_vec() { uint32_t l = u32(); std::vector<Ext2> v; for (uint32_t i = 0; i < l && ok; i++) v.push_back(ext()); return v; }
    HashOut hash() { HashOut h; for (int i = 0; i < 4; i++) h[i] = field(); return h; }
    std::vector<HashOut> cap() { uint32_t l = u32(); std::vector<HashOut> v; for (uint32_t i = 0; i < l && ok; i++) v.push_back(hash()); return v; }
    std::vector<HashOut> merkle_proof() { uint8_t l = u8(); std::vector<HashOut> v; for (int i = 0; i < l && ok; i++) v.push_back(hash()); return v; }
    StarkOpeningSet opening_set() {
        StarkOpeningSet s;
        s.local_values = ext_vec(); s.next_values = ext_vec(); s.permutation_ctl_zs = ext_vec();
        s.permutation_ctl_zs_next = ext_vec(); s.ctl_zs_last = field_vec(); s.quotient_polys = ext_vec();
        return s;
    }
    FriProof fri_proof() {
        FriProof p;
        uint32_t nc = u32();
        for (uint32_t i = 0; i < nc && ok; i++) p.commit_phase_merkle_caps.push_back(cap());
        uint32_t nq = u32();
        for (uint32_t q = 0; q < nq && ok; q++) {
            FriQueryRound r;
            uint32_t ni = u32();
            for (uint32_t i = 0; i < ni && ok; i++) { auto v = field_vec(); auto m = merkle_proof(); r.initial_trees_proof.evals_proofs.push_back({v, m}); }
            uint32_t ns = u32();
            for (uint32_t i = 0; i < ns && ok; i++) { FriQueryStep s; s.evals = ext_vec(); s.merkle_proof = merkle_proof(); r.steps.push_back(s); }
            p.query_round_proofs.push_back(r);
        }
        p.final_poly = ext_vec();
        p.pow_witness = field();
        return p;
    }
};
FriConfig make_cfg(const int* c) {
    FriConfig f;
    if (c) { f.rate_bits = c[0]; f.cap_height = c[1]; f.proof_of_work_bits = c[2]; f.arity_bits = c[3]; f.final_poly_bits = c[4]; f.num_query_rounds = c[5]; }
    return f;
}
}  // namespace

extern "C" {

// ---- field ----
void oracle_gl_vec_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        u64 x = gl_canon(a[i]), y = b ? gl_canon(b[i]) : 0;
        switch (op) {
            case 0: out[i] = gl_add(x, y); break;
            case 1: out[i] = gl_sub(x, y); break;
            case 2: out[i] = gl_mul(x, y); break;
            case 3: out[i] = x ? gl_inv(x) : 0; break;
            default: out[i] = 0;
        }
    }
}
u64 oracle_root_of_unity(int log_n) { return gl_root_of_unity(log_n); }
u64 oracle_gl_pow(u64 b, u64 e) { return gl_pow(gl_canon(b), e); }
void oracle_ext_mul(const u64* x, const u64* y, u64* out) { Ext2 r = ext_mul(Ext2{x[0], x[1]}, Ext2{y[0], y[1]}); out[0] = r.a; out[1] = r.b; }
void oracle_ext_inv(const u64* x, u64* out) { Ext2 r = ext_inv(Ext2{x[0], x[1]}); out[0] = r.a; out[1] = r.b; }

// ---- ntt ----
void oracle_evaluate_poly(u64* p, size_t n) { for (size_t i = 0; i < n; i++) p[i] = gl_canon(p[i]); evaluate_poly(p, n); }
void oracle_interpolate_poly(u64* p, size_t n) { for (size_t i = 0; i < n; i++) p[i] = gl_canon(p[i]); interpolate_poly(p, n); }
void oracle_evaluate_poly_with_offset(const u64* p, size_t n, u64 shift, size_t blowup, u64* out) {
    std::vector<u64> c(p, p + n);
    for (auto& x : c) x = gl_canon(x);
    std::vector<u64> r = evaluate_poly_with_offset(c.data(), n, shift, blowup);
    memcpy(out, r.data(), r.size() * 8);
}
void oracle_interpolate_poly_with_offset(u64* p, size_t n, u64 shift) { for (size_t i = 0; i < n; i++) p[i] = gl_canon(p[i]); interpolate_poly_with_offset(p, n, shift); }
void oracle_naive_eval(const u64* coeffs, size_t nc, size_t domain, u64 shift, u64* out) {
    std::vector<u64> c(coeffs, coeffs + nc);
    for (auto& x : c) x = gl_canon(x);
    std::vector<u64> r = naive_eval(c.data(), nc, domain, shift);
    memcpy(out, r.data(), r.size() * 8);
}

// ---- poseidon / sponge ----
void oracle_poseidon(u64* s) { poseidon_naive(s); }
void oracle_hash_no_pad(const u64* in, size_t n, u64* out) { HashOut h = hash_no_pad(in, n); memcpy(out, h.data(), 32); }
void oracle_two_to_one(const u64* l, const u64* r, u64* out) {
    HashOut h = two_to_one(HashOut{l[0], l[1], l[2], l[3]}, HashOut{r[0], r[1], r[2], r[3]});
    memcpy(out, h.data(), 32);
}

// ---- the configuration's hasher (blake3.cpp) ----
void oracle_set_hasher(int kind) { set_hasher(kind); }
int oracle_get_hasher() { return get_hasher(); }
void oracle_blake3(const uint8_t* in, size_t len, uint8_t* out32) { blake3_hash(in, len, out32); }
void oracle_blake3_permutation(u64* s) { blake3_permutation(s); }
void oracle_merkle_hash_leaf(const u64* in, size_t n, u64* out) { HashOut h = merkle_hash_leaf(in, n); memcpy(out, h.data(), 32); }
void oracle_merkle_two_to_one(const u64* l, const u64* r, u64* out) {
    HashOut h = merkle_two_to_one(HashOut{l[0], l[1], l[2], l[3]}, HashOut{r[0], r[1], r[2], r[3]});
    memcpy(out, h.data(), 32);
}

// ---- merkle ----
// cap_out: (1<<cap_height)*4;  leaf_hash_out (optional): num_leaves*4;  nodes_out (optional): num_leaves*4 (heap order)
void oracle_merkle(const u64* leaves, size_t num_leaves, size_t leaf_len, int cap_height, u64* cap_out,
                   u64* leaf_hash_out, u64* nodes_out) {
    MerkleTree t = merkle_new_v2(std::vector<u64>(leaves, leaves + num_leaves * leaf_len), num_leaves, leaf_len, cap_height);
    memcpy(cap_out, t.cap.data(), t.cap.size() * 32);
    if (leaf_hash_out) memcpy(leaf_hash_out, t.leaf_hash.data(), t.leaf_hash.size() * 32);
    if (nodes_out) memcpy(nodes_out, t.nodes.data(), t.nodes.size() * 32);
}
// returns 0 if for every leaf the heap-walk prove() equals the reference's digest-layout formula and verifies
int oracle_merkle_selfcheck(const u64* leaves, size_t num_leaves, size_t leaf_len, int cap_height) {
    MerkleTree t = merkle_new_v2(std::vector<u64>(leaves, leaves + num_leaves * leaf_len), num_leaves, leaf_len, cap_height);
    std::vector<HashOut> dg = merkle_reference_digests(t);
    for (size_t i = 0; i < num_leaves; i++) {
        std::vector<HashOut> a = t.prove(i);
        if (num_leaves > ((size_t)1 << cap_height)) {
            std::vector<HashOut> b = merkle_prove_via_digests(t, dg, i);
            if (a != b) return 1;
        }
        if (!verify_merkle_proof_to_cap(t.get(i), leaf_len, i, t.cap, a)) return 2;
    }
    return 0;
}

// ---- PolynomialBatch ----
void* oracle_batch_from_values(const u64* cols, int ncols, int log_n, int rate_bits, int cap_height) {
    size_t n = (size_t)1 << log_n;
    std::vector<std::vector<u64>> v(ncols);
    for (int c = 0; c < ncols; c++) { v[c].assign(cols + c * n, cols + (c + 1) * n); for (auto& x : v[c]) x = gl_canon(x); }
    return new PolynomialBatch(batch_from_values(v, rate_bits, cap_height));
}
void* oracle_batch_from_coeffs(const u64* cols, int ncols, int log_n, int rate_bits, int cap_height) {
    size_t n = (size_t)1 << log_n;
    std::vector<std::vector<u64>> v(ncols);
    for (int c = 0; c < ncols; c++) { v[c].assign(cols + c * n, cols + (c + 1) * n); for (auto& x : v[c]) x = gl_canon(x); }
    return new PolynomialBatch(batch_from_coeffs(std::move(v), rate_bits, cap_height));
}
void oracle_batch_free(void* h) { delete (PolynomialBatch*)h; }
void oracle_batch_cap(void* h, u64* out) { auto* b = (PolynomialBatch*)h; memcpy(out, b->merkle_tree.cap.data(), b->merkle_tree.cap.size() * 32); }
void oracle_batch_coeffs(void* h, u64* out) {  // column-major
    auto* b = (PolynomialBatch*)h;
    size_t n = b->polynomials[0].size();
    for (size_t c = 0; c < b->polynomials.size(); c++) memcpy(out + c * n, b->polynomials[c].data(), n * 8);
}
void oracle_batch_leaves(void* h, u64* out) { auto* b = (PolynomialBatch*)h; memcpy(out, b->merkle_tree.leaves.data(), b->merkle_tree.leaves.size() * 8); }
// one leaf (a row of the LDE in commitment order) without copying all of them
void oracle_batch_leaf(void* h, size_t leaf, u64* out) { auto* b = (PolynomialBatch*)h; memcpy(out, b->merkle_tree.get(leaf), b->merkle_tree.leaf_len * 8); }
// the same leaves under the hasher that is selected NOW (set_hasher): the full-size tests extend a batch once and check the
// Merkle tree of both hash configurations against it
void oracle_batch_rehash(void* h) {
    auto* b = (PolynomialBatch*)h;
    MerkleTree& t = b->merkle_tree;
    t = merkle_new_v2(std::move(t.leaves), t.num_leaves, t.leaf_len, t.cap_height);
}
int oracle_batch_prove(void* h, size_t leaf, u64* out) {
    auto* b = (PolynomialBatch*)h;
    std::vector<HashOut> s = b->merkle_tree.prove(leaf);
    memcpy(out, s.data(), s.size() * 32);
    return (int)s.size();
}

// ---- challenger ----
void* oracle_challenger_new() { return new Challenger(); }
void oracle_challenger_free(void* h) { delete (Challenger*)h; }
void oracle_challenger_observe(void* h, const u64* e, size_t n) { ((Challenger*)h)->observe_elements(e, n); }
u64 oracle_challenger_get(void* h) { return ((Challenger*)h)->get_challenge(); }
void oracle_challenger_observe_cap(void* h, const u64* digests, size_t n) {
    for (size_t i = 0; i < n; i++) ((Challenger*)h)->observe_hash(HashOut{digests[4 * i], digests[4 * i + 1], digests[4 * i + 2], digests[4 * i + 3]});
}
void oracle_challenger_compact(void* h) { ((Challenger*)h)->compact(); }
void oracle_challenger_state(void* h, u64* out12) { memcpy(out12, ((Challenger*)h)->sponge_state, 96); }

// ---- FRI ----
int oracle_fri_reduction_arity_bits(int degree_bits, const int* cfg6, int* out) {
    FriParams p = fri_params(make_cfg(cfg6), degree_bits);
    for (size_t i = 0; i < p.reduction_arity_bits.size(); i++) out[i] = p.reduction_arity_bits[i];
    return (int)p.reduction_arity_bits.size();
}
u64 oracle_fri_pow(const u64* h4, int bits) { FriConfig c; c.proof_of_work_bits = bits; return fri_proof_of_work(HashOut{h4[0], h4[1], h4[2], h4[3]}, c); }

// zeta + openings + FRI proof for three commitments; writes  opening_set bytes || fri_proof bytes ; returns length
// (or the required length if cap is too small).  zeta_out: 2 u64.
size_t oracle_open_and_prove(void* trace, void* zs, void* quot, int num_permutation_zs, void* challenger,
                             const int* cfg6, u64* zeta_out, uint8_t* out, size_t cap, size_t* openings_len) {
    OpeningProof p = open_and_prove(*(PolynomialBatch*)trace, *(PolynomialBatch*)zs, *(PolynomialBatch*)quot,
                                    num_permutation_zs, *(Challenger*)challenger, make_cfg(cfg6));
    ByteBuf b;
    b.opening_set(p.openings);
    if (openings_len) *openings_len = b.b.size();
    b.fri_proof(p.fri);
    if (zeta_out) { zeta_out[0] = p.zeta.a; zeta_out[1] = p.zeta.b; }
    if (b.b.size() <= cap) memcpy(out, b.b.data(), b.b.size());
    return b.b.size();
}

// Verifier for the same scenario: caps = 3 * (1<<cap_height) * 4 u64, num_polys[3].  The challenger must be in the
// state the prover's was in before drawing zeta.  Returns 0 = accept; otherwise writes the reason into msg.
int oracle_verify_opening(const u64* caps, const int* num_polys, int degree_bits, int num_permutation_zs,
                          const uint8_t* bytes, size_t len, void* challenger, const int* cfg6, char* msg, size_t msg_cap) {
    FriConfig cfg = make_cfg(cfg6);
    size_t cl = (size_t)1 << cfg.cap_height;
    std::vector<std::vector<HashOut>> cv(3, std::vector<HashOut>(cl));
    for (int t = 0; t < 3; t++) for (size_t i = 0; i < cl; i++) for (int k = 0; k < 4; k++) cv[t][i][k] = caps[(t * cl + i) * 4 + k];
    Reader r{bytes, len};
    StarkOpeningSet os = r.opening_set();
    FriProof fp = r.fri_proof();
    std::string why;
    if (!r.ok || r.off != len) why = "malformed proof bytes";
    else why = verify_opening(cv, {num_polys[0], num_polys[1], num_polys[2]}, degree_bits, num_permutation_zs, os, fp,
                              *(Challenger*)challenger, cfg);
    if (msg && msg_cap) { strncpy(msg, why.c_str(), msg_cap - 1); msg[msg_cap - 1] = 0; }
    return why.empty() ? 0 : 1;
}

}  // extern "C"

// ---- batch helpers for bench.py's cpu_baseline leg (OpenMP over independent columns) ----
#include <omp.h>
#include <sched.h>

#include <cstdio>
namespace {
// Threads the checker may really use: the OpenMP default is the number of visible CPUs, but a container's CPU quota can be far
// below it (a GPU box with 128 visible CPUs and a 16-CPU quota ran the oracle prover 8x slower, its 128 threads spinning on 16).
int usable_cpus() {
    int n = omp_get_max_threads();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a >= 1 && a < n) n = a; }
    long long quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
        fclose(f);
    } else {                                                           // cgroup v1
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
    return n < 1 ? 1 : n;
}
struct ThreadCap { ThreadCap() { omp_set_num_threads(usable_cpus()); } } g_thread_cap;
}  // namespace
extern "C" {
int oracle_num_threads() { return omp_get_max_threads(); }
// Horner evaluation of one polynomial at a few points (full-size NTT spot checks in tests/: fft.rs:218-252 evaluates
// naively at every point, which is out of reach at 2^24; a handful of points pins the same values)
void oracle_eval_at_points(const u64* coeffs, size_t n, const u64* points, size_t k, u64* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (long j = 0; j < (long)k; j++) {
        const u64 x = gl_canon(points[j]);
        u64 acc = 0;
        for (size_t i = n; i-- > 0;) acc = gl_add(gl_mul(acc, x), gl_canon(coeffs[i]));
        out[j] = acc;
    }
}
void oracle_evaluate_poly_batch(u64* data, size_t n, size_t batch) {
#pragma omp parallel for schedule(dynamic, 1)
    for (long c = 0; c < (long)batch; c++) evaluate_poly(data + (size_t)c * n, n);
}
}
