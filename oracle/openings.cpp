// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of the opening phase of
// prove_single_table and of the proof wire format.
//
// Follows (relative to /root/reference):
//   circuits/src/stark/proof.rs:198-233,235-265     StarkOpeningSet::new / to_fri_openings
//   circuits/src/stark/stark.rs:87-146              fri_instance (batches zeta, g*zeta, g^-1)
//   circuits/src/stark/prover.rs:499-553            zeta, subgroup check, observe openings, prove_openings
//   circuits/src/stark/serialization.rs:163-176,305-317,349-358   write_opening_set / write_fri_proof
//   plonky2/field/src/polynomial/mod.rs:167-172     PolynomialCoeffs::eval (Horner from the top)
//   plonky2/plonky2/src/fri/challenges.rs:16-23     observe_openings
#include "oracle.hpp"

namespace ola_oracle {

Ext2 eval_poly_ext(const std::vector<u64>& c, Ext2 z) {
    Ext2 acc = EXT_ZERO;
    for (size_t k = c.size(); k-- > 0;) acc = ext_add(ext_mul(acc, z), ext_from(c[k]));
    return acc;
}
u64 eval_poly_base(const std::vector<u64>& c, u64 z) {
    u64 acc = 0;
    for (size_t k = c.size(); k-- > 0;) acc = gl_add(gl_mul(acc, z), c[k]);
    return acc;
}

StarkOpeningSet stark_opening_set(Ext2 zeta, u64 g, const PolynomialBatch& trace, const PolynomialBatch& zs,
                                  const PolynomialBatch& quotient, int num_permutation_zs) {
    StarkOpeningSet s;
    Ext2 zeta_next = ext_scalar_mul(zeta, g);
    for (auto& p : trace.polynomials) { s.local_values.push_back(eval_poly_ext(p, zeta)); s.next_values.push_back(eval_poly_ext(p, zeta_next)); }
    for (auto& p : zs.polynomials) { s.permutation_ctl_zs.push_back(eval_poly_ext(p, zeta)); s.permutation_ctl_zs_next.push_back(eval_poly_ext(p, zeta_next)); }
    u64 g_inv = gl_inv(g);
    for (size_t i = num_permutation_zs; i < zs.polynomials.size(); i++) s.ctl_zs_last.push_back(eval_poly_base(zs.polynomials[i], g_inv));
    for (auto& p : quotient.polynomials) s.quotient_polys.push_back(eval_poly_ext(p, zeta));
    return s;
}

FriOpenings StarkOpeningSet::to_fri_openings() const {
    FriOpenings o;
    std::vector<Ext2> zb = local_values;
    zb.insert(zb.end(), permutation_ctl_zs.begin(), permutation_ctl_zs.end());
    zb.insert(zb.end(), quotient_polys.begin(), quotient_polys.end());
    std::vector<Ext2> nb = next_values;
    nb.insert(nb.end(), permutation_ctl_zs_next.begin(), permutation_ctl_zs_next.end());
    std::vector<Ext2> lb;
    for (u64 x : ctl_zs_last) lb.push_back(ext_from(x));
    o.batches = {zb, nb, lb};
    return o;
}

FriInstanceInfo stark_fri_instance(Ext2 zeta, u64 g, int degree_bits, int trace_cols, int num_permutation_batches,
                                   int num_ctl_zs, int num_quotient_polys) {
    (void)degree_bits;
    FriInstanceInfo inst;
    int nz = num_permutation_batches + num_ctl_zs;
    inst.oracle_num_polys = {trace_cols, nz, num_quotient_polys};
    FriBatchInfo zb, nb, lb;
    zb.point = zeta;
    nb.point = ext_scalar_mul(zeta, g);
    lb.point = ext_from(gl_inv(g));
    for (int i = 0; i < trace_cols; i++) { zb.polynomials.push_back({0, i}); nb.polynomials.push_back({0, i}); }
    for (int i = 0; i < nz; i++) { zb.polynomials.push_back({1, i}); nb.polynomials.push_back({1, i}); }
    for (int i = 0; i < num_quotient_polys; i++) zb.polynomials.push_back({2, i});
    for (int i = num_permutation_batches; i < nz; i++) lb.polynomials.push_back({1, i});
    inst.batches = {zb, nb, lb};
    return inst;
}

void ByteBuf::opening_set(const StarkOpeningSet& s) {
    ext_vec(s.local_values);
    ext_vec(s.next_values);
    ext_vec(s.permutation_ctl_zs);
    ext_vec(s.permutation_ctl_zs_next);
    field_vec(s.ctl_zs_last);
    ext_vec(s.quotient_polys);
}

void ByteBuf::fri_proof(const FriProof& p) {
    u32((uint32_t)p.commit_phase_merkle_caps.size());
    for (auto& c : p.commit_phase_merkle_caps) cap(c);
    u32((uint32_t)p.query_round_proofs.size());
    for (auto& qr : p.query_round_proofs) {
        u32((uint32_t)qr.initial_trees_proof.evals_proofs.size());
        for (auto& ep : qr.initial_trees_proof.evals_proofs) { field_vec(ep.first); merkle_proof(ep.second); }
        u32((uint32_t)qr.steps.size());
        for (auto& st : qr.steps) { ext_vec(st.evals); merkle_proof(st.merkle_proof); }
    }
    ext_vec(p.final_poly);
    field(p.pow_witness);
}

static void observe_openings(Challenger& ch, const FriOpenings& o) {
    for (auto& b : o.batches) for (auto& e : b) ch.observe_ext(e);
}

OpeningProof open_and_prove(const PolynomialBatch& trace, const PolynomialBatch& zs, const PolynomialBatch& quotient,
                            int num_permutation_zs, Challenger& ch, const FriConfig& cfg) {
    OpeningProof out;
    int degree_bits = trace.degree_log;
    out.zeta = ch.get_extension_challenge();
    u64 g = gl_root_of_unity(degree_bits);
    out.openings = stark_opening_set(out.zeta, g, trace, zs, quotient, num_permutation_zs);
    observe_openings(ch, out.openings.to_fri_openings());
    FriInstanceInfo inst = stark_fri_instance(out.zeta, g, degree_bits, (int)trace.polynomials.size(), num_permutation_zs,
                                              (int)zs.polynomials.size() - num_permutation_zs, (int)quotient.polynomials.size());
    FriParams params = fri_params(cfg, degree_bits);
    out.fri = prove_openings(inst, {&trace, &zs, &quotient}, ch, params);
    return out;
}

std::string verify_opening(const std::vector<std::vector<HashOut>>& caps, const std::vector<int>& num_polys,
                           int degree_bits, int num_permutation_zs, const StarkOpeningSet& openings,
                           const FriProof& fri, Challenger& ch, const FriConfig& cfg) {
    Ext2 zeta = ch.get_extension_challenge();
    u64 g = gl_root_of_unity(degree_bits);
    FriOpenings fo = openings.to_fri_openings();
    observe_openings(ch, fo);
    FriChallenges fc = fri_challenges(ch, fri.commit_phase_merkle_caps, fri.final_poly, fri.pow_witness, degree_bits, cfg);
    FriInstanceInfo inst = stark_fri_instance(zeta, g, degree_bits, num_polys[0], num_permutation_zs,
                                              num_polys[1] - num_permutation_zs, num_polys[2]);
    FriParams params = fri_params(cfg, degree_bits);
    return verify_fri_proof(inst, fo, fc, caps, fri, params);
}

}  // namespace ola_oracle
