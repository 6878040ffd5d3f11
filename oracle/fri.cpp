// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of the reference's polynomial
// commitment (PolynomialBatch) and FRI prover / verifier.
//
// Follows (relative to /root/reference/plonky2/plonky2/src):
//   fri/oracle.rs:45-64     PolynomialBatch::from_values  (per-column ifft)
//   fri/oracle.rs:66-99     from_coeffs: coset LDE (shift 7, blowup 2^rate_bits) -> transpose -> bit-reverse rows
//                           -> MerkleTree::new_v2(leaves, cap_height)
//   fri/oracle.rs:167-241   prove_openings (alpha; per batch composition / divide_by_linear; *X; LDE; coset FFT)
//   fri/prover.rs:20-70     fri_proof;  :72-121 fri_committed_trees;  :126-148 fri_proof_of_work;
//                           :150-204 query rounds
//   fri/reduction_strategies.rs:40-52   ConstantArityBits
//   fri/challenges.rs:25-73 fri_challenges (verifier-side transcript)
//   fri/verifier.rs:19-270  compute_evaluation, PoW check, fri_combine_initial, query round, final poly check
//   fri/validate_shape.rs:11-67
//   util/reducing.rs:55-104 ReducingFactor::{reduce, shift};  plonk/plonk_common.rs:116-128 reduce_with_powers
//   /root/reference/plonky2/field/src/polynomial/division.rs:74-87  divide_by_linear
//   /root/reference/plonky2/field/src/interpolation.rs:31-66        barycentric interpolate
#include "oracle.hpp"

namespace ola_oracle {

FriParams fri_params(const FriConfig& cfg, int degree_bits) {
    FriParams p;
    p.config = cfg;
    p.degree_bits = degree_bits;
    int d = degree_bits;
    while (d > cfg.final_poly_bits && d + cfg.rate_bits - cfg.arity_bits >= cfg.cap_height) {
        p.reduction_arity_bits.push_back(cfg.arity_bits);
        d -= cfg.arity_bits;
    }
    return p;
}

PolynomialBatch batch_from_coeffs(std::vector<std::vector<u64>> coeffs, int rate_bits, int cap_height) {
    PolynomialBatch b;
    size_t degree = coeffs[0].size();
    size_t ncols = coeffs.size();
    size_t N = degree << rate_bits;
    b.degree_log = log2_strict(degree);
    b.rate_bits = rate_bits;
    int lde_bits = b.degree_log + rate_bits;
    std::vector<u64> leaves(N * ncols);
#pragma omp parallel for schedule(dynamic, 1)
    for (long c = 0; c < (long)ncols; c++) {
        std::vector<u64> lde = evaluate_poly_with_offset(coeffs[c].data(), degree, GL_GENERATOR, (size_t)1 << rate_bits);
        // transpose + reverse_index_bits_in_place(leaves): leaf j is natural LDE row bitrev(j)
        for (size_t j = 0; j < N; j++) leaves[j * ncols + c] = lde[reverse_bits(j, lde_bits)];
    }
    b.merkle_tree = merkle_new_v2(std::move(leaves), N, ncols, cap_height);
    b.polynomials = std::move(coeffs);
    return b;
}

PolynomialBatch batch_from_values(const std::vector<std::vector<u64>>& values, int rate_bits, int cap_height) {
    std::vector<std::vector<u64>> coeffs = values;
#pragma omp parallel for schedule(dynamic, 1)
    for (long c = 0; c < (long)coeffs.size(); c++) interpolate_poly(coeffs[c].data(), coeffs[c].size());
    return batch_from_coeffs(std::move(coeffs), rate_bits, cap_height);
}

std::vector<Ext2> divide_by_linear(const std::vector<Ext2>& p, Ext2 z) {
    // (p(X) - p(z)) / (X - z) by Horner from the top coefficient; the last accumulator (= p(z)) is dropped.
    size_t n = p.size();
    if (n == 0) return {};
    std::vector<Ext2> q(n - 1);
    Ext2 acc = EXT_ZERO;
    for (size_t k = n; k-- > 0;) {
        acc = ext_add(ext_mul(acc, z), p[k]);
        if (k > 0) q[k - 1] = acc;
    }
    return q;
}

u64 fri_proof_of_work(const HashOut& h, const FriConfig& cfg) {
    // Minimal satisfying nonce (the reference's find_any is schedule dependent).  Searched in blocks so that the
    // candidates of a block can be hashed in parallel; the smallest hit of the first block with a hit is the minimum.
    const u64 block = 1 << 12;
    for (u64 base = 0;; base += block) {
        u64 best = ~(u64)0;
#pragma omp parallel for reduction(min : best)
        for (long k = 0; k < (long)block; k++) {
            const u64 i = base + (u64)k;
            u64 in[5] = {h[0], h[1], h[2], h[3], i};
            const u64 r = hash_no_pad(in, 5)[0];
            // leading_zeros(r) >= pow_bits + (64 - 64)
            if ((r >> (64 - cfg.proof_of_work_bits)) == 0 && i < best) best = i;
        }
        if (best != ~(u64)0) return best;
    }
}

static std::vector<u64> flatten(const Ext2* e, size_t n) {
    std::vector<u64> f(2 * n);
    for (size_t i = 0; i < n; i++) { f[2 * i] = e[i].a; f[2 * i + 1] = e[i].b; }
    return f;
}

FriProof prove_openings(const FriInstanceInfo& instance, const std::vector<const PolynomialBatch*>& oracles,
                        Challenger& challenger, const FriParams& params) {
    Ext2 alpha = challenger.get_extension_challenge();
    std::vector<Ext2> final_poly;
    for (const FriBatchInfo& batch : instance.batches) {
        size_t poly_len = batch.polynomials.size();
        size_t n = oracles[batch.polynomials[0].oracle_index]->polynomials[0].size();
        std::vector<Ext2> comp(n, EXT_ZERO);
        Ext2 a = EXT_ONE;
        for (size_t i = 0; i < poly_len; i++) {
            const std::vector<u64>& f =
                oracles[batch.polynomials[i].oracle_index]->polynomials[batch.polynomials[i].polynomial_index];
            for (size_t k = 0; k < n; k++) comp[k] = ext_add(comp[k], ext_scalar_mul(a, f[k]));
            a = ext_mul(a, alpha);
        }
        std::vector<Ext2> quotient = divide_by_linear(comp, batch.point);
        Ext2 scale = ext_pow(alpha, poly_len);
        if (final_poly.size() < quotient.size()) final_poly.resize(quotient.size(), EXT_ZERO);
        for (size_t k = 0; k < final_poly.size(); k++) {
            Ext2 v = ext_mul(final_poly[k], scale);
            if (k < quotient.size()) v = ext_add(v, quotient[k]);
            final_poly[k] = v;
        }
    }
    final_poly.insert(final_poly.begin(), EXT_ZERO);  // * X
    size_t N = final_poly.size() << params.config.rate_bits;
    std::vector<Ext2> coeffs = final_poly;
    coeffs.resize(N, EXT_ZERO);
    std::vector<Ext2> values = ext_coset_fft(coeffs, GL_GENERATOR);

    // ---- fri_committed_trees ----
    FriProof proof;
    std::vector<MerkleTree> trees;
    u64 shift = GL_GENERATOR;
    for (int arity_bits : params.reduction_arity_bits) {
        size_t arity = (size_t)1 << arity_bits;
        size_t len = values.size();
        int bits = log2_strict(len);
        std::vector<u64> leaves(2 * len);
        for (size_t j = 0; j < len; j++) {
            Ext2 v = values[reverse_bits(j, bits)];
            leaves[2 * j] = v.a;
            leaves[2 * j + 1] = v.b;
        }
        MerkleTree tree = merkle_new_v2(std::move(leaves), len / arity, 2 * arity, params.config.cap_height);
        challenger.observe_cap(tree.cap);
        proof.commit_phase_merkle_caps.push_back(tree.cap);
        trees.push_back(std::move(tree));
        Ext2 beta = challenger.get_extension_challenge();
        std::vector<Ext2> folded(coeffs.size() / arity);
        for (size_t j = 0; j < folded.size(); j++) {
            Ext2 s = EXT_ZERO;
            for (size_t k = arity; k-- > 0;) s = ext_add(ext_mul(s, beta), coeffs[j * arity + k]);
            folded[j] = s;
        }
        coeffs = std::move(folded);
        shift = gl_pow(shift, arity);
        values = ext_coset_fft(coeffs, shift);
    }
    coeffs.resize(coeffs.size() >> params.config.rate_bits);
    for (const Ext2& c : coeffs) challenger.observe_ext(c);
    proof.final_poly = coeffs;

    // ---- PoW ----
    HashOut cur = challenger.get_hash();
    proof.pow_witness = fri_proof_of_work(cur, params.config);

    // ---- queries ----
    size_t n = N;
    for (int r = 0; r < params.config.num_query_rounds; r++) {
        size_t x_index = (size_t)(challenger.get_challenge() % (u64)n);
        FriQueryRound round;
        for (const PolynomialBatch* o : oracles) {
            const MerkleTree& t = o->merkle_tree;
            std::vector<u64> row(t.get(x_index), t.get(x_index) + t.leaf_len);
            round.initial_trees_proof.evals_proofs.push_back({row, t.prove(x_index)});
        }
        for (size_t i = 0; i < trees.size(); i++) {
            int ab = params.reduction_arity_bits[i];
            size_t idx = x_index >> ab;
            FriQueryStep step;
            const u64* leaf = trees[i].get(idx);
            for (size_t k = 0; k < ((size_t)1 << ab); k++) step.evals.push_back(Ext2{leaf[2 * k], leaf[2 * k + 1]});
            step.merkle_proof = trees[i].prove(idx);
            round.steps.push_back(std::move(step));
            x_index = idx;
        }
        proof.query_round_proofs.push_back(std::move(round));
    }
    return proof;
}

// ------------------------------------------------------------------------------------------------
// verifier side
// ------------------------------------------------------------------------------------------------
FriChallenges fri_challenges(Challenger& ch, const std::vector<std::vector<HashOut>>& commit_caps,
                             const std::vector<Ext2>& final_poly, u64 pow_witness, int degree_bits,
                             const FriConfig& cfg) {
    FriChallenges c;
    size_t lde_size = (size_t)1 << (degree_bits + cfg.rate_bits);
    c.fri_alpha = ch.get_extension_challenge();
    for (auto& cap : commit_caps) {
        ch.observe_cap(cap);
        c.fri_betas.push_back(ch.get_extension_challenge());
    }
    for (auto& e : final_poly) ch.observe_ext(e);
    HashOut h = ch.get_hash();
    u64 in[5] = {h[0], h[1], h[2], h[3], pow_witness};
    c.fri_pow_response = hash_no_pad(in, 5)[0];
    for (int i = 0; i < cfg.num_query_rounds; i++) c.fri_query_indices.push_back((size_t)(ch.get_challenge() % lde_size));
    return c;
}

static Ext2 interpolate_bary(const std::vector<std::pair<Ext2, Ext2>>& pts, Ext2 x) {
    size_t n = pts.size();
    for (auto& p : pts) if (p.first == x) return p.second;
    Ext2 lx = EXT_ONE;
    for (auto& p : pts) lx = ext_mul(lx, ext_sub(x, p.first));
    Ext2 sum = EXT_ZERO;
    for (size_t i = 0; i < n; i++) {
        Ext2 d = EXT_ONE;
        for (size_t j = 0; j < n; j++) if (j != i) d = ext_mul(d, ext_sub(pts[i].first, pts[j].first));
        Ext2 w = ext_inv(d);
        sum = ext_add(sum, ext_mul(ext_mul(w, ext_inv(ext_sub(x, pts[i].first))), pts[i].second));
    }
    return ext_mul(lx, sum);
}

static Ext2 compute_evaluation(u64 x, size_t x_index_within_coset, int arity_bits, const std::vector<Ext2>& evals_in,
                               Ext2 beta) {
    size_t arity = (size_t)1 << arity_bits;
    u64 g = gl_root_of_unity(arity_bits);
    std::vector<Ext2> evals(arity);
    for (size_t i = 0; i < arity; i++) evals[i] = evals_in[reverse_bits(i, arity_bits)];
    size_t rev = reverse_bits(x_index_within_coset, arity_bits);
    u64 coset_start = gl_mul(x, gl_pow(g, arity - rev));
    std::vector<std::pair<Ext2, Ext2>> pts(arity);
    u64 y = 1;
    for (size_t i = 0; i < arity; i++) {
        pts[i] = {ext_from(gl_mul(coset_start, y)), evals[i]};
        y = gl_mul(y, g);
    }
    return interpolate_bary(pts, beta);
}

static Ext2 reduce_ext(const std::vector<Ext2>& v, Ext2 alpha) {
    Ext2 acc = EXT_ZERO;
    for (size_t i = v.size(); i-- > 0;) acc = ext_add(ext_mul(acc, alpha), v[i]);
    return acc;
}

std::string verify_fri_proof(const FriInstanceInfo& instance, const FriOpenings& openings, const FriChallenges& ch,
                             const std::vector<std::vector<HashOut>>& initial_caps, const FriProof& proof,
                             const FriParams& params) {
    const FriConfig& cfg = params.config;
    // validate_fri_proof_shape
    for (auto& cap : proof.commit_phase_merkle_caps)
        if (cap.size() != ((size_t)1 << cfg.cap_height)) return "cap height";
    if ((int)proof.query_round_proofs.size() != cfg.num_query_rounds) return "number of query rounds";
    for (auto& qr : proof.query_round_proofs) {
        if (qr.initial_trees_proof.evals_proofs.size() != instance.oracle_num_polys.size()) return "initial proofs";
        for (size_t o = 0; o < instance.oracle_num_polys.size(); o++) {
            if ((int)qr.initial_trees_proof.evals_proofs[o].first.size() != instance.oracle_num_polys[o]) return "leaf len";
            if ((int)qr.initial_trees_proof.evals_proofs[o].second.size() + cfg.cap_height != params.lde_bits())
                return "initial merkle proof len";
        }
        if (qr.steps.size() != params.reduction_arity_bits.size()) return "steps";
        int bits = params.lde_bits();
        for (size_t i = 0; i < qr.steps.size(); i++) {
            bits -= params.reduction_arity_bits[i];
            if (qr.steps[i].evals.size() != ((size_t)1 << params.reduction_arity_bits[i])) return "evals len";
            if ((int)qr.steps[i].merkle_proof.size() + cfg.cap_height != bits) return "step merkle proof len";
        }
    }
    if (proof.final_poly.size() != ((size_t)1 << (params.degree_bits - params.total_arities()))) return "final poly len";

    if ((ch.fri_pow_response >> (64 - cfg.proof_of_work_bits)) != 0) return "Invalid proof of work witness.";

    std::vector<Ext2> reduced_openings;
    for (auto& b : openings.batches) reduced_openings.push_back(reduce_ext(b, ch.fri_alpha));

    int log_n = params.lde_bits();
    for (size_t r = 0; r < ch.fri_query_indices.size(); r++) {
        size_t x_index = ch.fri_query_indices[r];
        const FriQueryRound& round = proof.query_round_proofs[r];
        for (size_t o = 0; o < initial_caps.size(); o++) {
            auto& ep = round.initial_trees_proof.evals_proofs[o];
            if (!verify_merkle_proof_to_cap(ep.first.data(), ep.first.size(), x_index, initial_caps[o], ep.second))
                return "Invalid Merkle proof (initial tree).";
        }
        u64 subgroup_x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(log_n), reverse_bits(x_index, log_n)));
        // fri_combine_initial
        Ext2 sum = EXT_ZERO;
        Ext2 sx = ext_from(subgroup_x);
        for (size_t b = 0; b < instance.batches.size(); b++) {
            const FriBatchInfo& batch = instance.batches[b];
            std::vector<Ext2> evals;
            for (auto& p : batch.polynomials)
                evals.push_back(ext_from(round.initial_trees_proof.evals_proofs[p.oracle_index].first[p.polynomial_index]));
            Ext2 reduced = reduce_ext(evals, ch.fri_alpha);
            Ext2 num = ext_sub(reduced, reduced_openings[b]);
            Ext2 den = ext_sub(sx, batch.point);
            sum = ext_mul(sum, ext_pow(ch.fri_alpha, evals.size()));
            sum = ext_add(sum, ext_mul(num, ext_inv(den)));
        }
        Ext2 old_eval = ext_mul(sum, sx);
        for (size_t i = 0; i < params.reduction_arity_bits.size(); i++) {
            int ab = params.reduction_arity_bits[i];
            size_t arity = (size_t)1 << ab;
            const std::vector<Ext2>& evals = round.steps[i].evals;
            size_t coset_index = x_index >> ab;
            size_t within = x_index & (arity - 1);
            if (evals[within] != old_eval) return "FRI consistency check failed";
            old_eval = compute_evaluation(subgroup_x, within, ab, evals, ch.fri_betas[i]);
            std::vector<u64> flat = flatten(evals.data(), evals.size());
            if (!verify_merkle_proof_to_cap(flat.data(), flat.size(), coset_index, proof.commit_phase_merkle_caps[i],
                                            round.steps[i].merkle_proof))
                return "Invalid Merkle proof (commit phase).";
            for (int k = 0; k < ab; k++) subgroup_x = gl_mul(subgroup_x, subgroup_x);
            x_index = coset_index;
        }
        Ext2 fe = EXT_ZERO, sxe = ext_from(subgroup_x);
        for (size_t k = proof.final_poly.size(); k-- > 0;) fe = ext_add(ext_mul(fe, sxe), proof.final_poly[k]);
        if (fe != old_eval) return "Final polynomial evaluation is invalid.";
    }
    return "";
}

}  // namespace ola_oracle
