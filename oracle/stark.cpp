// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of the multi-table STARK prover and
// verifier, generic over an "AIR-set" description (tables, permutation pairs, cross-table lookups, constraint programs;
// format documented in olavm_amd/air/dsl.py -- the description is DATA shared with the product, the interpreters are not).
//
// Follows (relative to /root/reference/circuits/src/stark):
//   prover.rs:79-327      prove_with_traces      (trace commitments, caps -> challenger, CTL data, per-table proofs)
//   prover.rs:330-567     prove_single_table     (compact, permutation challenges, Zs, alphas, quotient, zeta, openings)
//   prover.rs:571-705     compute_quotient_polys (coset points, Lagrange first/last, Z_H inverse, coset_ifft)
//   constraint_consumer.rs:34-78                 ConstraintConsumer
//   vanishing_poly.rs:20-45                      eval_vanishing_poly = AIR, permutation checks, CTL checks
//   permutation.rs:103-155,268-289,302-360       Z polys, batches, eval_permutation_checks
//   cross_table_lookup.rs:224-311,380-421,551-584  CTL data, checks, verify_cross_table_lookups
//   get_challenges.rs:16-49,95-149               transcript replay
//   verifier.rs:35-206,220-324,381-388           verify_proof, verify_stark_proof_with_challenges, eval_l_0_and_l_last
//   serialization.rs:349-358,377-393             write_proof / write_all_proof
//   /root/reference/plonky2/field/src/zero_poly_coset.rs:18-52   ZeroPolyOnCoset
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "oracle.hpp"

namespace ola_oracle {

// ------------------------------------------------------------------------------------------------ AIR-set description
enum { OP_LOCAL = 0, OP_NEXT, OP_CONST, OP_PARAM, OP_ADD, OP_SUB, OP_MUL, OP_EMIT, OP_ISZERO };
enum { KIND_ALL = 0, KIND_TRANSITION, KIND_FIRST, KIND_LAST };

struct AirOp { int op, kind, dst, a, b; u64 imm; };
struct LinCol { std::vector<std::pair<int, u64>> terms; u64 constant = 0; };
struct TableWithColumns { int table = 0; std::vector<LinCol> columns; bool has_filter = false; LinCol filter; };
struct Ctl { std::vector<TableWithColumns> looking; TableWithColumns looked; };
struct AirTable {
    int ncols = 0, constraint_degree = 0, n_regs = 0, n_params = 0;
    std::vector<std::vector<std::pair<int, int>>> perm_pairs;
    std::vector<AirOp> ops;
    int quotient_degree_factor() const { return std::max(1, constraint_degree - 1); }
    int permutation_batch_size() const { return quotient_degree_factor(); }
    bool uses_permutation_args() const { return !perm_pairs.empty(); }
    int num_permutation_batches(int num_challenges) const {
        int inst = (int)perm_pairs.size() * num_challenges;
        return inst ? (inst + permutation_batch_size() - 1) / permutation_batch_size() : 0;
    }
};
struct AirSet { std::vector<AirTable> tables; std::vector<Ctl> ctls; };

static AirSet parse_airset(const u64* w, size_t n) {
    size_t p = 0;
    auto next = [&]() -> u64 { if (p >= n) throw std::runtime_error("airset blob truncated"); return w[p++]; };
    if (next() != 0x4F4C41414952ull) throw std::runtime_error("bad airset magic");
    if (next() != 1) throw std::runtime_error("bad airset version");
    size_t nt = next(), nc = next();
    AirSet s;
    for (size_t t = 0; t < nt; t++) {
        AirTable a;
        a.ncols = (int)next(); a.constraint_degree = (int)next(); a.n_regs = (int)next(); a.n_params = (int)next();
        size_t np = next();
        for (size_t i = 0; i < np; i++) {
            size_t len = next();
            std::vector<std::pair<int, int>> pr;
            for (size_t k = 0; k < len; k++) { int l = (int)next(); int r = (int)next(); pr.push_back({l, r}); }
            a.perm_pairs.push_back(pr);
        }
        size_t nops = next();
        for (size_t i = 0; i < nops; i++) {
            u64 w0 = next(), w1 = next();
            AirOp o;
            o.op = (int)(w0 & 0xff); o.kind = (int)((w0 >> 8) & 0xff); o.dst = (int)((w0 >> 16) & 0xffff);
            o.a = (int)((w0 >> 32) & 0xffff); o.b = (int)((w0 >> 48) & 0xffff); o.imm = w1;
            a.ops.push_back(o);
        }
        s.tables.push_back(a);
    }
    auto col = [&]() { LinCol c; size_t nt2 = next(); for (size_t i = 0; i < nt2; i++) { int cc = (int)next(); u64 f = next(); c.terms.push_back({cc, f}); } c.constant = next(); return c; };
    auto twc = [&]() { TableWithColumns t; t.table = (int)next(); size_t ncol = next(); for (size_t i = 0; i < ncol; i++) t.columns.push_back(col()); t.has_filter = next() != 0; if (t.has_filter) t.filter = col(); return t; };
    for (size_t c = 0; c < nc; c++) {
        Ctl ctl;
        size_t nl = next();
        for (size_t i = 0; i < nl; i++) ctl.looking.push_back(twc());
        ctl.looked = twc();
        s.ctls.push_back(ctl);
    }
    if (p != n) throw std::runtime_error("airset blob has trailing words");
    return s;
}

// ------------------------------------------------------------------------------------------------ field adaptors
struct BaseF {
    typedef u64 T;
    static T zero() { return 0; }
    static T one() { return 1; }
    static T from_base(u64 x) { return x; }
    static T add(T a, T b) { return gl_add(a, b); }
    static T sub(T a, T b) { return gl_sub(a, b); }
    static T mul(T a, T b) { return gl_mul(a, b); }
    static bool is_zero(T a) { return a == 0; }
};
struct ExtF {
    typedef Ext2 T;
    static T zero() { return EXT_ZERO; }
    static T one() { return EXT_ONE; }
    static T from_base(u64 x) { return ext_from(x); }
    static T add(T a, T b) { return ext_add(a, b); }
    static T sub(T a, T b) { return ext_sub(a, b); }
    static T mul(T a, T b) { return ext_mul(a, b); }
    static bool is_zero(T a) { return a.a == 0 && a.b == 0; }
};

template <class F>
struct Consumer {
    typedef typename F::T T;
    std::vector<T> alphas, accs;
    T z_last, lagrange_first, lagrange_last;
    void constraint(T c) { for (size_t i = 0; i < alphas.size(); i++) accs[i] = F::add(F::mul(accs[i], alphas[i]), c); }
    void emit(int kind, T c) {
        if (kind == KIND_TRANSITION) c = F::mul(c, z_last);
        else if (kind == KIND_FIRST) c = F::mul(c, lagrange_first);
        else if (kind == KIND_LAST) c = F::mul(c, lagrange_last);
        constraint(c);
    }
};

struct GrandProductChallenge { u64 beta, gamma; };
typedef std::vector<GrandProductChallenge> ChallengeSet;  // num_challenges entries

struct CtlZData {
    std::vector<u64> z;
    GrandProductChallenge challenge;
    const TableWithColumns* twc;
};

template <class F>
static typename F::T eval_lincol(const LinCol& c, const typename F::T* v) {
    typename F::T s = F::zero();
    for (auto& t : c.terms) s = F::add(s, F::mul(v[t.first], F::from_base(t.second)));
    return F::add(s, F::from_base(c.constant));
}

// eval_vanishing_poly: the table's constraint program, then permutation checks, then CTL checks
template <class F>
static void eval_vanishing_poly(const AirTable& air, int num_challenges, const typename F::T* local, const typename F::T* next,
                                const u64* params, const typename F::T* local_zs, const typename F::T* next_zs,
                                const std::vector<ChallengeSet>* perm_sets,
                                const std::vector<std::pair<GrandProductChallenge, const TableWithColumns*>>& ctl_vars,
                                Consumer<F>& consumer) {
    typedef typename F::T T;
    std::vector<T> reg(air.n_regs, F::zero());
    for (const AirOp& o : air.ops) {
        switch (o.op) {
            case OP_LOCAL: reg[o.dst] = local[o.a]; break;
            case OP_NEXT: reg[o.dst] = next[o.a]; break;
            case OP_CONST: reg[o.dst] = F::from_base(o.imm); break;
            case OP_PARAM: reg[o.dst] = F::from_base(params[o.a]); break;
            case OP_ADD: reg[o.dst] = F::add(reg[o.a], reg[o.b]); break;
            case OP_SUB: reg[o.dst] = F::sub(reg[o.a], reg[o.b]); break;
            case OP_MUL: reg[o.dst] = F::mul(reg[o.a], reg[o.b]); break;
            case OP_EMIT: consumer.emit(o.kind, reg[o.a]); break;
            case OP_ISZERO: reg[o.dst] = F::is_zero(reg[o.a]) ? F::one() : F::zero(); break;
            default: throw std::runtime_error("bad AIR op");
        }
    }
    const int nperm = air.num_permutation_batches(num_challenges);
    if (perm_sets) {
        for (int i = 0; i < nperm; i++) consumer.emit(KIND_FIRST, F::sub(local_zs[i], F::one()));
        // get_permutation_batches: cartesian product (pair, challenge) chunked by batch size
        const int bs = air.permutation_batch_size();
        int inst = 0;
        const int total = (int)air.perm_pairs.size() * num_challenges;
        for (int b = 0; b < nperm; b++) {
            T prod_l = F::one(), prod_r = F::one();
            for (int i = 0; i < bs && inst < total; i++, inst++) {
                const auto& pair = air.perm_pairs[inst / num_challenges];
                const GrandProductChallenge ch = (*perm_sets)[i][inst % num_challenges];
                T l = F::zero(), r = F::zero();
                for (size_t k = pair.size(); k-- > 0;) {  // reduce: Horner from the last term
                    l = F::add(F::mul(l, F::from_base(ch.beta)), local[pair[k].first]);
                    r = F::add(F::mul(r, F::from_base(ch.beta)), local[pair[k].second]);
                }
                prod_l = F::mul(prod_l, F::add(l, F::from_base(ch.gamma)));
                prod_r = F::mul(prod_r, F::add(r, F::from_base(ch.gamma)));
            }
            consumer.emit(KIND_ALL, F::sub(F::mul(next_zs[b], prod_r), F::mul(local_zs[b], prod_l)));
        }
    }
    for (size_t i = 0; i < ctl_vars.size(); i++) {
        const GrandProductChallenge ch = ctl_vars[i].first;
        const TableWithColumns& twc = *ctl_vars[i].second;
        auto combine = [&](const T* v) {
            T acc = F::zero();
            for (size_t k = twc.columns.size(); k-- > 0;) acc = F::add(F::mul(acc, F::from_base(ch.beta)), eval_lincol<F>(twc.columns[k], v));
            return F::add(acc, F::from_base(ch.gamma));
        };
        auto filter = [&](const T* v) { return twc.has_filter ? eval_lincol<F>(twc.filter, v) : F::one(); };
        auto select = [&](T f, T x) { return F::sub(F::add(F::mul(f, x), F::one()), f); };
        const T lz = local_zs[nperm + i], nz = next_zs[nperm + i];
        consumer.emit(KIND_FIRST, F::sub(lz, select(filter(local), combine(local))));
        consumer.emit(KIND_TRANSITION, F::sub(nz, F::mul(lz, select(filter(next), combine(next)))));
    }
}

// ------------------------------------------------------------------------------------------------ prover pieces
static u64 eval_lincol_table(const LinCol& c, const std::vector<std::vector<u64>>& trace, size_t row) {
    u64 s = 0;
    for (auto& t : c.terms) s = gl_add(s, gl_mul(trace[t.first][row], t.second));
    return gl_add(s, c.constant);
}

// cross_table_lookup.rs:284-311  partial_products (inclusive prefix product over filtered rows)
static std::vector<u64> partial_products(const std::vector<std::vector<u64>>& trace, const TableWithColumns& twc, GrandProductChallenge ch) {
    size_t degree = trace[0].size();
    std::vector<u64> res(degree);
    u64 pp = 1;
    for (size_t i = 0; i < degree; i++) {
        u64 f = twc.has_filter ? eval_lincol_table(twc.filter, trace, i) : 1;
        if (f == 1) {
            u64 acc = 0;
            for (size_t k = twc.columns.size(); k-- > 0;) acc = gl_add(gl_mul(acc, ch.beta), eval_lincol_table(twc.columns[k], trace, i));
            pp = gl_mul(pp, gl_add(acc, ch.gamma));
        } else if (f != 0) {
            throw std::runtime_error("Non-binary filter?");
        }
        res[i] = pp;
    }
    return res;
}

static GrandProductChallenge get_gp_challenge(Challenger& ch) { u64 b = ch.get_challenge(); u64 g = ch.get_challenge(); return {b, g}; }
static ChallengeSet get_gp_challenge_set(Challenger& ch, int num) { ChallengeSet s; for (int i = 0; i < num; i++) s.push_back(get_gp_challenge(ch)); return s; }

struct StarkProof {
    std::vector<HashOut> trace_cap, zs_cap, quotient_cap;
    StarkOpeningSet openings;
    FriProof fri;
};

// ORACLE_TIMING=1: where the checker's own time goes (stderr)
struct OracleClock {
    const char* what; std::chrono::steady_clock::time_point t0;
    explicit OracleClock(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {}
    ~OracleClock() { if (getenv("ORACLE_TIMING")) fprintf(stderr, "[oracle-timing] %-28s %8.2f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
};

struct ProverConfig { int num_challenges = 2; FriConfig fri; };

static std::vector<std::vector<u64>> compute_permutation_z_polys(const AirTable& air, const ProverConfig& cfg,
                                                                 const std::vector<std::vector<u64>>& trace,
                                                                 const std::vector<ChallengeSet>& sets) {
    const int nperm = air.num_permutation_batches(cfg.num_challenges), bs = air.permutation_batch_size();
    const int total = (int)air.perm_pairs.size() * cfg.num_challenges;
    size_t degree = trace[0].size();
    std::vector<std::vector<u64>> zs;
    int inst = 0;
    for (int b = 0; b < nperm; b++) {
        std::vector<u64> num(degree, 1), den(degree, 1);
        for (int i = 0; i < bs && inst < total; i++, inst++) {
            const auto& pair = air.perm_pairs[inst / cfg.num_challenges];
            const GrandProductChallenge ch = sets[i][inst % cfg.num_challenges];
            for (size_t r = 0; r < degree; r++) {
                u64 l = ch.gamma, rr = ch.gamma, w = 1;
                for (auto& pr : pair) { l = gl_add(l, gl_mul(trace[pr.first][r], w)); rr = gl_add(rr, gl_mul(trace[pr.second][r], w)); w = gl_mul(w, ch.beta); }
                num[r] = gl_mul(num[r], l);
                den[r] = gl_mul(den[r], rr);
            }
        }
        std::vector<u64> z(degree);
        u64 acc = 1;
        for (size_t r = 0; r < degree; r++) { z[r] = acc; acc = gl_mul(acc, gl_mul(num[r], gl_inv(den[r]))); }
        zs.push_back(z);
    }
    return zs;
}

static std::vector<std::vector<u64>> compute_quotient_polys(const AirTable& air, const ProverConfig& cfg, const PolynomialBatch& trace_c,
                                                            const PolynomialBatch& zs_c, const std::vector<ChallengeSet>* perm_sets,
                                                            const std::vector<CtlZData>& ctl, const std::vector<u64>& alphas,
                                                            const u64* params, int degree_bits) {
    const size_t degree = (size_t)1 << degree_bits;
    const int rate_bits = cfg.fri.rate_bits;
    int qdb = 0;
    while ((1 << qdb) < air.quotient_degree_factor()) qdb++;
    if (qdb > rate_bits) throw std::runtime_error("Having constraints of degree higher than the rate is not supported yet.");
    const size_t step = (size_t)1 << (rate_bits - qdb), next_step = (size_t)1 << qdb;
    const size_t size = degree << qdb;
    // Lagrange selectors on the coset: selector(n, idx).lde_onto_coset(qdb)
    auto lde_selector = [&](size_t idx) {
        std::vector<u64> v(degree, 0);
        v[idx] = 1;
        interpolate_poly(v.data(), degree);
        return evaluate_poly_with_offset(v.data(), degree, GL_GENERATOR, (size_t)1 << qdb);
    };
    std::vector<u64> lag_first = lde_selector(0), lag_last = lde_selector(degree - 1);
    // ZeroPolyOnCoset::new(degree_bits, qdb)
    u64 g_pow_n = GL_GENERATOR;
    for (int i = 0; i < degree_bits; i++) g_pow_n = gl_mul(g_pow_n, g_pow_n);
    std::vector<u64> zh_inv((size_t)1 << qdb);
    {
        u64 v = gl_root_of_unity(qdb), x = 1;
        for (size_t i = 0; i < zh_inv.size(); i++) { zh_inv[i] = gl_inv(gl_sub(gl_mul(g_pow_n, x), 1)); x = gl_mul(x, v); }
    }
    const u64 last = gl_inv(gl_root_of_unity(degree_bits));
    const u64 w = gl_root_of_unity(degree_bits + qdb);
    std::vector<std::pair<GrandProductChallenge, const TableWithColumns*>> ctl_vars;
    for (auto& c : ctl) ctl_vars.push_back({c.challenge, c.twc});
    const int nch = (int)alphas.size();
    std::vector<std::vector<u64>> qvals(nch, std::vector<u64>(size));
    // the reference walks the points with a running x (prover.rs:633-640, rayon over the points); here blocks of points on the host
    // cores, every block starting from its own x = g w^start -- the checker's speed is test budget, its values are the same
    const size_t block = 1024;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t b0 = 0; b0 < size; b0 += block) {
        u64 x = gl_mul(GL_GENERATOR, gl_pow(w, b0));
        for (size_t i = b0; i < std::min(size, b0 + block); i++) {
            const size_t i_next = (i + next_step) % size;
            Consumer<BaseF> consumer;
            consumer.alphas = alphas;
            consumer.accs.assign(nch, 0);
            consumer.z_last = gl_sub(x, last);
            consumer.lagrange_first = lag_first[i];
            consumer.lagrange_last = lag_last[i];
            eval_vanishing_poly<BaseF>(air, cfg.num_challenges, trace_c.get_lde_values(i, step), trace_c.get_lde_values(i_next, step), params,
                                       zs_c.get_lde_values(i, step), zs_c.get_lde_values(i_next, step), perm_sets, ctl_vars, consumer);
            for (int j = 0; j < nch; j++) qvals[j][i] = gl_mul(consumer.accs[j], zh_inv[i % zh_inv.size()]);
            x = gl_mul(x, w);
        }
    }
    for (auto& q : qvals) interpolate_poly_with_offset(q.data(), size, GL_GENERATOR);
    return qvals;
}

static StarkProof prove_single_table(const AirTable& air, const ProverConfig& cfg, const std::vector<std::vector<u64>>& trace,
                                     const PolynomialBatch& trace_c, const std::vector<CtlZData>& ctl, const u64* params,
                                     Challenger& ch) {
    const size_t degree = trace[0].size();
    const int degree_bits = log2_strict(degree);
    FriParams fp = fri_params(cfg.fri, degree_bits);
    if (fp.total_arities() > degree_bits + cfg.fri.rate_bits - cfg.fri.cap_height) throw std::runtime_error("FRI total reduction arity is too large.");
    ch.compact();
    std::vector<ChallengeSet> perm_sets;
    const bool use_perm = air.uses_permutation_args();
    if (use_perm) for (int i = 0; i < air.permutation_batch_size(); i++) perm_sets.push_back(get_gp_challenge_set(ch, cfg.num_challenges));
    std::vector<std::vector<u64>> z_polys;
    if (use_perm) { OracleClock c("  permutation Z"); z_polys = compute_permutation_z_polys(air, cfg, trace, perm_sets); }
    const int num_permutation_zs = (int)z_polys.size();
    for (auto& c : ctl) z_polys.push_back(c.z);
    if (z_polys.empty()) throw std::runtime_error("No CTL?");
    OracleClock* c_zs = new OracleClock("  Zs commitment");
    PolynomialBatch zs_c = batch_from_values(z_polys, cfg.fri.rate_bits, cfg.fri.cap_height);
    delete c_zs;
    ch.observe_cap(zs_c.merkle_tree.cap);
    std::vector<u64> alphas;
    for (int i = 0; i < cfg.num_challenges; i++) alphas.push_back(ch.get_challenge());
    OracleClock* c_q = new OracleClock("  quotient polys");
    std::vector<std::vector<u64>> qpolys = compute_quotient_polys(air, cfg, trace_c, zs_c, use_perm ? &perm_sets : nullptr, ctl, alphas, params, degree_bits);
    delete c_q;
    std::vector<std::vector<u64>> chunks;
    const size_t keep = degree * (size_t)air.quotient_degree_factor();
    for (auto& q : qpolys) {
        for (size_t k = keep; k < q.size(); k++)
            if (q[k] != 0) throw std::runtime_error("Quotient has failed, the vanishing polynomial is not divisible by Z_H");
        for (size_t c0 = 0; c0 < keep; c0 += degree) chunks.push_back(std::vector<u64>(q.begin() + c0, q.begin() + c0 + degree));
    }
    OracleClock* c_qc = new OracleClock("  quotient commitment");
    PolynomialBatch q_c = batch_from_coeffs(chunks, cfg.fri.rate_bits, cfg.fri.cap_height);
    delete c_qc;
    ch.observe_cap(q_c.merkle_tree.cap);
    OracleClock c_open("  openings + FRI");
    OpeningProof op = open_and_prove(trace_c, zs_c, q_c, num_permutation_zs, ch, cfg.fri);
    if (ext_pow(op.zeta, (u64)1 << degree_bits) == EXT_ONE) throw std::runtime_error("Opening point is in the subgroup.");
    StarkProof p;
    p.trace_cap = trace_c.merkle_tree.cap;
    p.zs_cap = zs_c.merkle_tree.cap;
    p.quotient_cap = q_c.merkle_tree.cap;
    p.openings = op.openings;
    p.fri = op.fri;
    return p;
}

struct AllProof { std::vector<StarkProof> proofs; std::vector<u64> compress_challenges; };

static AllProof prove_with_traces(const AirSet& set, const ProverConfig& cfg, const std::vector<std::vector<std::vector<u64>>>& traces,
                                  const std::vector<std::vector<u64>>& params, const std::vector<u64>& compress_challenges) {
    const size_t nt = set.tables.size();
    std::vector<PolynomialBatch> commits;
    { OracleClock c("trace commitments"); for (size_t t = 0; t < nt; t++) commits.push_back(batch_from_values(traces[t], cfg.fri.rate_bits, cfg.fri.cap_height)); }
    Challenger ch;
    for (size_t t = 0; t < nt; t++) ch.observe_cap(commits[t].merkle_tree.cap);
    // cross_table_lookup_data
    ChallengeSet ctl_ch = get_gp_challenge_set(ch, cfg.num_challenges);
    std::vector<std::vector<CtlZData>> ctl_data(nt);
    OracleClock* c_ctl = new OracleClock("cross_table_lookup_data");
    for (const Ctl& ctl : set.ctls) {
        for (const GrandProductChallenge& c : ctl_ch) {
            for (const TableWithColumns& twc : ctl.looking) ctl_data[twc.table].push_back({partial_products(traces[twc.table], twc, c), c, &twc});
            ctl_data[ctl.looked.table].push_back({partial_products(traces[ctl.looked.table], ctl.looked, c), c, &ctl.looked});
        }
    }
    delete c_ctl;
    AllProof all;
    for (size_t t = 0; t < nt; t++) {
        OracleClock c("prove_single_table");
        all.proofs.push_back(prove_single_table(set.tables[t], cfg, traces[t], commits[t], ctl_data[t], params[t].data(), ch));
    }
    all.compress_challenges = compress_challenges;
    return all;
}

static void write_all_proof(ByteBuf& b, const AllProof& all) {
    b.u32((uint32_t)all.proofs.size());
    for (auto& p : all.proofs) {
        b.cap(p.trace_cap); b.cap(p.zs_cap); b.cap(p.quotient_cap);
        b.opening_set(p.openings);
        b.fri_proof(p.fri);
    }
    b.field_vec(all.compress_challenges);
}

// ------------------------------------------------------------------------------------------------ verifier
struct ProofReader {
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
    StarkProof proof() {
        StarkProof s;
        s.trace_cap = cap(); s.zs_cap = cap(); s.quotient_cap = cap();
        s.openings.local_values = ext_vec(); s.openings.next_values = ext_vec(); s.openings.permutation_ctl_zs = ext_vec();
        s.openings.permutation_ctl_zs_next = ext_vec(); s.openings.ctl_zs_last = field_vec(); s.openings.quotient_polys = ext_vec();
        uint32_t nc = u32();
        for (uint32_t i = 0; i < nc && ok; i++) s.fri.commit_phase_merkle_caps.push_back(cap());
        uint32_t nq = u32();
        for (uint32_t q = 0; q < nq && ok; q++) {
            FriQueryRound r;
            uint32_t ni = u32();
            for (uint32_t i = 0; i < ni && ok; i++) { auto v = field_vec(); auto m = merkle_proof(); r.initial_trees_proof.evals_proofs.push_back({v, m}); }
            uint32_t ns = u32();
            for (uint32_t i = 0; i < ns && ok; i++) { FriQueryStep st; st.evals = ext_vec(); st.merkle_proof = merkle_proof(); r.steps.push_back(st); }
            s.fri.query_round_proofs.push_back(r);
        }
        s.fri.final_poly = ext_vec();
        s.fri.pow_witness = field();
        return s;
    }
};

static std::string verify_all_proof(const AirSet& set, const ProverConfig& cfg, const AllProof& all, const std::vector<std::vector<u64>>& params) {
    const size_t nt = set.tables.size();
    if (all.proofs.size() != nt) return "wrong number of table proofs";
    Challenger ch;
    for (auto& p : all.proofs) ch.observe_cap(p.trace_cap);
    ChallengeSet ctl_ch = get_gp_challenge_set(ch, cfg.num_challenges);
    // CtlCheckVars::from_proofs: distribute the opened ctl zs in cross_table_lookup order
    std::vector<std::vector<std::pair<GrandProductChallenge, const TableWithColumns*>>> ctl_vars(nt);
    for (const Ctl& ctl : set.ctls)
        for (const GrandProductChallenge& c : ctl_ch) {
            for (const TableWithColumns& twc : ctl.looking) ctl_vars[twc.table].push_back({c, &twc});
            ctl_vars[ctl.looked.table].push_back({c, &ctl.looked});
        }
    for (size_t t = 0; t < nt; t++) {
        const AirTable& air = set.tables[t];
        const StarkProof& p = all.proofs[t];
        const std::string tag = "table " + std::to_string(t) + ": ";
        ch.compact();
        if (p.fri.query_round_proofs.empty() || p.fri.query_round_proofs[0].initial_trees_proof.evals_proofs.empty()) return tag + "empty FRI proof";
        const int lde_bits = cfg.fri.cap_height + (int)p.fri.query_round_proofs[0].initial_trees_proof.evals_proofs[0].second.size();
        const int degree_bits = lde_bits - cfg.fri.rate_bits;
        std::vector<ChallengeSet> perm_sets;
        const bool use_perm = air.num_permutation_batches(cfg.num_challenges) > 0;
        if (use_perm) for (int i = 0; i < air.permutation_batch_size(); i++) perm_sets.push_back(get_gp_challenge_set(ch, cfg.num_challenges));
        ch.observe_cap(p.zs_cap);
        std::vector<u64> alphas;
        for (int i = 0; i < cfg.num_challenges; i++) alphas.push_back(ch.get_challenge());
        ch.observe_cap(p.quotient_cap);
        const Ext2 zeta = ch.get_extension_challenge();
        FriOpenings fo = p.openings.to_fri_openings();
        for (auto& b : fo.batches) for (auto& e : b) ch.observe_ext(e);
        FriChallenges fc = fri_challenges(ch, p.fri.commit_phase_merkle_caps, p.fri.final_poly, p.fri.pow_witness, degree_bits, cfg.fri);
        // validate_proof_shape
        const int nperm = air.num_permutation_batches(cfg.num_challenges);
        const int num_ctl = (int)ctl_vars[t].size();
        if ((int)p.openings.local_values.size() != air.ncols || (int)p.openings.next_values.size() != air.ncols) return tag + "opening width";
        if ((int)p.openings.permutation_ctl_zs.size() != nperm + num_ctl || (int)p.openings.permutation_ctl_zs_next.size() != nperm + num_ctl) return tag + "zs width";
        if ((int)p.openings.ctl_zs_last.size() != num_ctl) return tag + "ctl_zs_last width";
        if ((int)p.openings.quotient_polys.size() != air.quotient_degree_factor() * cfg.num_challenges) return tag + "quotient width";
        // constraint check at zeta
        const u64 g = gl_root_of_unity(degree_bits);
        const Ext2 z_x = ext_sub(ext_pow(zeta, (u64)1 << degree_bits), EXT_ONE);
        const u64 nn = ((u64)1 << degree_bits) % GL_P;
        Consumer<ExtF> consumer;
        for (u64 a : alphas) consumer.alphas.push_back(ext_from(a));
        consumer.accs.assign(alphas.size(), EXT_ZERO);
        consumer.z_last = ext_sub(zeta, ext_from(gl_inv(g)));
        consumer.lagrange_first = ext_mul(z_x, ext_inv(ext_scalar_mul(ext_sub(zeta, EXT_ONE), nn)));
        consumer.lagrange_last = ext_mul(z_x, ext_inv(ext_scalar_mul(ext_sub(ext_scalar_mul(zeta, g), EXT_ONE), nn)));
        eval_vanishing_poly<ExtF>(air, cfg.num_challenges, p.openings.local_values.data(), p.openings.next_values.data(), params[t].data(),
                                  p.openings.permutation_ctl_zs.data(), p.openings.permutation_ctl_zs_next.data(),
                                  use_perm ? &perm_sets : nullptr, ctl_vars[t], consumer);
        const Ext2 zeta_pow_deg = ext_pow(zeta, (u64)1 << degree_bits);
        const int q = air.quotient_degree_factor();
        for (size_t i = 0; i < alphas.size(); i++) {
            Ext2 acc = EXT_ZERO;
            for (int k = q; k-- > 0;) acc = ext_add(ext_mul(acc, zeta_pow_deg), p.openings.quotient_polys[i * q + k]);
            if (consumer.accs[i] != ext_mul(z_x, acc)) return tag + "Mismatch between evaluation and opening of quotient polynomial";
        }
        FriInstanceInfo inst = stark_fri_instance(zeta, g, degree_bits, air.ncols, nperm, num_ctl, q * cfg.num_challenges);
        FriParams fp = fri_params(cfg.fri, degree_bits);
        std::string why = verify_fri_proof(inst, fo, fc, {p.trace_cap, p.zs_cap, p.quotient_cap}, p.fri, fp);
        if (!why.empty()) return tag + why;
    }
    // verify_cross_table_lookups (extra looking products are all one, verifier.rs:152)
    std::vector<size_t> pos(nt, 0);
    for (const Ctl& ctl : set.ctls)
        for (int c = 0; c < cfg.num_challenges; c++) {
            u64 prod = 1;
            for (const TableWithColumns& twc : ctl.looking) prod = gl_mul(prod, all.proofs[twc.table].openings.ctl_zs_last[pos[twc.table]++]);
            const u64 looked = all.proofs[ctl.looked.table].openings.ctl_zs_last[pos[ctl.looked.table]++];
            if (prod != looked) return "Cross-table lookup verification failed.";
        }
    return "";
}

}  // namespace ola_oracle

// ------------------------------------------------------------------------------------------------ C entry points
using namespace ola_oracle;

static ProverConfig make_pcfg(const int* c) {
    ProverConfig p;
    if (c) { p.fri.rate_bits = c[0]; p.fri.cap_height = c[1]; p.fri.proof_of_work_bits = c[2]; p.fri.arity_bits = c[3]; p.fri.final_poly_bits = c[4]; p.fri.num_query_rounds = c[5]; }
    return p;
}
static void load_inputs(const AirSet& set, const u64* const* traces, const uint32_t* log_n, const u64* params,
                        std::vector<std::vector<std::vector<u64>>>& tr, std::vector<std::vector<u64>>& pr) {
    size_t poff = 0;
    for (size_t t = 0; t < set.tables.size(); t++) {
        size_t n = (size_t)1 << log_n[t];
        std::vector<std::vector<u64>> cols(set.tables[t].ncols);
        for (int c = 0; c < set.tables[t].ncols; c++) { cols[c].assign(traces[t] + (size_t)c * n, traces[t] + (size_t)(c + 1) * n); for (auto& x : cols[c]) x = gl_canon(x); }
        tr.push_back(cols);
        std::vector<u64> p(params ? params + poff : nullptr, params ? params + poff + set.tables[t].n_params : nullptr);
        if (!params) p.assign(set.tables[t].n_params, 0);
        poff += set.tables[t].n_params;
        pr.push_back(p);
    }
}

extern "C" {

// traces[t]: column-major ncols x 2^log_n[t]; params: concatenated per-table parameters; compress: one per table.
// Writes the all-proof wire bytes; returns the length, 0 on failure (message in err).
size_t oracle_prove_with_traces(const u64* airset, size_t airset_words, const u64* const* traces, const uint32_t* log_n,
                                const u64* params, const u64* compress, const int* cfg6, uint8_t* out, size_t cap, char* err, size_t err_cap) {
    try {
        AirSet set = parse_airset(airset, airset_words);
        std::vector<std::vector<std::vector<u64>>> tr;
        std::vector<std::vector<u64>> pr;
        load_inputs(set, traces, log_n, params, tr, pr);
        std::vector<u64> cc(compress ? compress : nullptr, compress ? compress + set.tables.size() : nullptr);
        if (!compress) cc.assign(set.tables.size(), 0);
        AllProof all = prove_with_traces(set, make_pcfg(cfg6), tr, pr, cc);
        ByteBuf b;
        write_all_proof(b, all);
        if (b.b.size() <= cap && out) memcpy(out, b.b.data(), b.b.size());
        return b.b.size();
    } catch (const std::exception& e) {
        if (err && err_cap) { strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
        return 0;
    }
}

int oracle_verify_all_proof(const u64* airset, size_t airset_words, const u64* params, const int* cfg6, const uint8_t* bytes, size_t len,
                            char* msg, size_t msg_cap) {
    std::string why;
    try {
        AirSet set = parse_airset(airset, airset_words);
        ProofReader r{bytes, len};
        AllProof all;
        uint32_t np = r.u32();
        for (uint32_t i = 0; i < np && r.ok; i++) all.proofs.push_back(r.proof());
        all.compress_challenges = r.field_vec();
        std::vector<std::vector<u64>> pr;
        size_t poff = 0;
        for (auto& t : set.tables) { pr.push_back(std::vector<u64>(params ? params + poff : nullptr, params ? params + poff + t.n_params : nullptr)); if (!params) pr.back().assign(t.n_params, 0); poff += t.n_params; }
        if (!r.ok || r.off != len) why = "malformed proof bytes";
        else why = verify_all_proof(set, make_pcfg(cfg6), all, pr);
    } catch (const std::exception& e) { why = e.what(); }
    if (msg && msg_cap) { strncpy(msg, why.c_str(), msg_cap - 1); msg[msg_cap - 1] = 0; }
    return why.empty() ? 0 : 1;
}

// Row-by-row constraint check on the trace domain (the reference's own AIR test recipe, test_utils.rs:152-195):
// evaluates the table's constraint program on (row i, row i+1 mod n) with first/last selectors and returns the index
// of the first row with a non-zero accumulator, or -1.
long oracle_check_constraints(const u64* airset, size_t airset_words, int table, const u64* trace, uint32_t log_n, const u64* params) {
    AirSet set = parse_airset(airset, airset_words);
    const AirTable& air = set.tables[table];
    size_t n = (size_t)1 << log_n;
    std::vector<u64> local(air.ncols), next(air.ncols), p(params, params + air.n_params);
    std::vector<std::pair<GrandProductChallenge, const TableWithColumns*>> none;
    u64 x = 1, g = gl_root_of_unity(log_n), last = gl_inv(g);
    for (size_t i = 0; i < n; i++) {
        for (int c = 0; c < air.ncols; c++) { local[c] = gl_canon(trace[(size_t)c * n + i]); next[c] = gl_canon(trace[(size_t)c * n + (i + 1) % n]); }
        Consumer<BaseF> consumer;
        consumer.alphas = {1}; consumer.accs = {0};
        // with alpha = 1 a sum could cancel; use two independent alphas
        consumer.alphas = {0x9E3779B97F4A7C15ull % GL_P, 0xC2B2AE3D27D4EB4Full % GL_P}; consumer.accs = {0, 0};
        consumer.z_last = gl_sub(x, last);
        consumer.lagrange_first = (i == 0) ? 1 : 0;
        consumer.lagrange_last = (i == n - 1) ? 1 : 0;
        eval_vanishing_poly<BaseF>(air, 2, local.data(), next.data(), p.data(), nullptr, nullptr, nullptr, none, consumer);
        if (consumer.accs[0] != 0 || consumer.accs[1] != 0) return (long)i;
        x = gl_mul(x, g);
    }
    return -1;
}

}  // extern "C"
