// ORACLE -- test infrastructure only.  Declarations of the CPU restatement of the reference's
// `prove_with_traces` hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// use anything under oracle/, and only as the checker.  See each .cpp for the reference file:line it follows.
// Pinned by: the reference's Poseidon known answers and golden Poseidon-table rows; outputs of the reference's own hashing,
// transcript, FRI-parameter and AIR code run from its source by tools/rust_air_eval.py (tests/golden/ref_primitive_vectors.json,
// air_eval_vectors.json); and the reference's verify_proof, interpreted (tools/ref_verifier.py), accepting this prover's proof
// (tests/golden/ref_verified/).  NOT compared byte for byte with the Rust prover (no Rust toolchain in the image).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "gl.hpp"

namespace ola_oracle {

// ---------------- ntt.cpp ----------------
void permute(u64* v, size_t n);
std::vector<u64> get_twiddles(size_t n);
std::vector<u64> get_inv_twiddles(size_t n);
void evaluate_poly(u64* p, size_t n);
std::vector<u64> evaluate_poly_with_offset(const u64* p, size_t n, u64 domain_offset, size_t blowup);
void interpolate_poly(u64* e, size_t n);
void interpolate_poly_with_offset(u64* e, size_t n, u64 domain_offset);
std::vector<u64> naive_eval(const u64* coeffs, size_t ncoeffs, size_t domain, u64 shift);
std::vector<Ext2> ext_coset_fft(const std::vector<Ext2>& coeffs, u64 shift);

// ---------------- poseidon.cpp ----------------
typedef std::array<u64, 12> State;
typedef std::array<u64, 4> HashOut;
void poseidon_naive(u64 state[12]);
State poseidon(const State& in);
HashOut hash_no_pad(const u64* in, size_t n);
HashOut two_to_one(const HashOut& l, const HashOut& r);

// ---------------- blake3.cpp ----------------
// GenericConfig::Hasher (plonk/config.rs:112-161): Poseidon (PoseidonGoldilocksConfig, the default) or Blake3_256
// (Blake3GoldilocksConfig).  A process-wide switch of the checker; InnerHasher (proof of work) is Poseidon in both.
enum { HASH_POSEIDON = 0, HASH_BLAKE3 = 1 };
void set_hasher(int kind);
int get_hasher();
void blake3_hash(const uint8_t* in, size_t len, uint8_t out[32]);
HashOut blake3_hash_no_pad(const u64* in, size_t n);               // digest = 32 bytes as 4 little-endian words, never reduced
HashOut blake3_two_to_one(const HashOut& l, const HashOut& r);
void blake3_permutation(u64 state[12]);
void blake3_digest_elements(const HashOut& h, u64 out[5]);
HashOut merkle_hash_leaf(const u64* in, size_t n);                 // Hasher::hash_no_pad of the selected configuration
HashOut merkle_two_to_one(const HashOut& l, const HashOut& r);

// ---------------- merkle.cpp ----------------
struct MerkleTree {
    size_t num_leaves = 0, leaf_len = 0;
    int cap_height = 0;
    std::vector<u64> leaves;        // row-major, num_leaves x leaf_len
    std::vector<HashOut> leaf_hash;  // num_leaves
    std::vector<HashOut> nodes;      // heap order, root at 1, nodes[n/2 .. n) = parents of leaves (n = num_leaves)
    std::vector<HashOut> cap;        // 1 << cap_height
    const u64* get(size_t i) const { return leaves.data() + i * leaf_len; }
    std::vector<HashOut> prove(size_t leaf_index) const;
};
MerkleTree merkle_new_v2(std::vector<u64> leaves_row_major, size_t num_leaves, size_t leaf_len, int cap_height);
// reference digest layout (merkle_tree/mod.rs:213-259) -- used to check prove() against the reference's formula
std::vector<HashOut> merkle_reference_digests(const MerkleTree& t);
std::vector<HashOut> merkle_prove_via_digests(const MerkleTree& t, const std::vector<HashOut>& digests, size_t leaf_index);
bool verify_merkle_proof_to_cap(const u64* leaf, size_t leaf_len, size_t leaf_index, const std::vector<HashOut>& cap,
                                const std::vector<HashOut>& siblings);

// ---------------- challenger.cpp ----------------
struct Challenger {
    u64 sponge_state[12] = {0};
    std::vector<u64> input_buffer, output_buffer;
    int hasher = get_hasher();      // Challenger<F, C::Hasher>: the permutation and what observing a digest means
    void observe_element(u64 e);
    void observe_elements(const u64* e, size_t n);
    void observe_ext(Ext2 e) { observe_element(e.a); observe_element(e.b); }
    void observe_hash(const HashOut& h) {
        if (hasher == HASH_BLAKE3) { u64 e[5]; blake3_digest_elements(h, e); observe_elements(e, 5); }
        else observe_elements(h.data(), 4);
    }
    void observe_cap(const std::vector<HashOut>& cap) { for (auto& h : cap) observe_hash(h); }
    u64 get_challenge();
    Ext2 get_extension_challenge() { u64 a = get_challenge(); u64 b = get_challenge(); return Ext2{a, b}; }
    HashOut get_hash() { HashOut h; for (int i = 0; i < 4; i++) h[i] = get_challenge(); return h; }
    void duplexing();
    void compact();
};

// ---------------- fri.cpp ----------------
struct FriConfig {
    int rate_bits = 3, cap_height = 4, proof_of_work_bits = 16;
    int arity_bits = 4, final_poly_bits = 5;  // ConstantArityBits(4, 5)
    int num_query_rounds = 28;
};
struct FriParams {
    FriConfig config;
    int degree_bits = 0;
    std::vector<int> reduction_arity_bits;
    int total_arities() const { int s = 0; for (int a : reduction_arity_bits) s += a; return s; }
    int lde_bits() const { return degree_bits + config.rate_bits; }
};
FriParams fri_params(const FriConfig& cfg, int degree_bits);

// PolynomialBatch (fri/oracle.rs:31-139): coefficient form + Merkle tree over the bit-reversed LDE rows.
struct PolynomialBatch {
    std::vector<std::vector<u64>> polynomials;  // coefficient form, each of length 2^degree_log
    MerkleTree merkle_tree;
    int degree_log = 0, rate_bits = 0;
    // fri/oracle.rs:131-137  get_lde_values(index, step)
    const u64* get_lde_values(size_t index, size_t step) const {
        return merkle_tree.get(reverse_bits(index * step, degree_log + rate_bits));
    }
};
PolynomialBatch batch_from_values(const std::vector<std::vector<u64>>& values, int rate_bits, int cap_height);
PolynomialBatch batch_from_coeffs(std::vector<std::vector<u64>> coeffs, int rate_bits, int cap_height);

struct FriPolynomialInfo { int oracle_index, polynomial_index; };
struct FriBatchInfo { Ext2 point; std::vector<FriPolynomialInfo> polynomials; };
struct FriInstanceInfo { std::vector<int> oracle_num_polys; std::vector<FriBatchInfo> batches; };

struct FriQueryStep { std::vector<Ext2> evals; std::vector<HashOut> merkle_proof; };
struct FriInitialTreeProof { std::vector<std::pair<std::vector<u64>, std::vector<HashOut>>> evals_proofs; };
struct FriQueryRound { FriInitialTreeProof initial_trees_proof; std::vector<FriQueryStep> steps; };
struct FriProof {
    std::vector<std::vector<HashOut>> commit_phase_merkle_caps;
    std::vector<FriQueryRound> query_round_proofs;
    std::vector<Ext2> final_poly;
    u64 pow_witness = 0;
};

std::vector<Ext2> divide_by_linear(const std::vector<Ext2>& p, Ext2 z);
FriProof prove_openings(const FriInstanceInfo& instance, const std::vector<const PolynomialBatch*>& oracles,
                        Challenger& challenger, const FriParams& params);
u64 fri_proof_of_work(const HashOut& current_hash, const FriConfig& cfg);

// openings as seen by the FRI verifier (fri/structure.rs:58-80): per batch, the claimed values
struct FriOpenings { std::vector<std::vector<Ext2>> batches; };
struct FriChallenges {
    Ext2 fri_alpha;
    std::vector<Ext2> fri_betas;
    u64 fri_pow_response;
    std::vector<size_t> fri_query_indices;
};
FriChallenges fri_challenges(Challenger& challenger, const std::vector<std::vector<HashOut>>& commit_caps,
                             const std::vector<Ext2>& final_poly, u64 pow_witness, int degree_bits,
                             const FriConfig& cfg);
// returns empty string on success, otherwise the reason
std::string verify_fri_proof(const FriInstanceInfo& instance, const FriOpenings& openings,
                             const FriChallenges& challenges, const std::vector<std::vector<HashOut>>& initial_caps,
                             const FriProof& proof, const FriParams& params);

}  // namespace ola_oracle

// ---------------- openings.cpp ----------------
namespace ola_oracle {
// circuits/src/stark/proof.rs:181-233
struct StarkOpeningSet {
    std::vector<Ext2> local_values, next_values, permutation_ctl_zs, permutation_ctl_zs_next;
    std::vector<u64> ctl_zs_last;
    std::vector<Ext2> quotient_polys;
    FriOpenings to_fri_openings() const;
};
Ext2 eval_poly_ext(const std::vector<u64>& coeffs, Ext2 z);
u64 eval_poly_base(const std::vector<u64>& coeffs, u64 z);
StarkOpeningSet stark_opening_set(Ext2 zeta, u64 g, const PolynomialBatch& trace, const PolynomialBatch& zs,
                                  const PolynomialBatch& quotient, int num_permutation_zs);
FriInstanceInfo stark_fri_instance(Ext2 zeta, u64 g, int degree_bits, int trace_cols, int num_permutation_batches,
                                   int num_ctl_zs, int num_quotient_polys);

// wire format (circuits/src/stark/serialization.rs)
struct ByteBuf {
    std::vector<uint8_t> b;
    void u8(uint8_t x) { b.push_back(x); }
    void u32(uint32_t x) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void field(u64 x) { x = gl_canon(x); for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void ext(Ext2 e) { field(e.a); field(e.b); }
    void field_vec(const std::vector<u64>& v) { u32((uint32_t)v.size()); for (u64 x : v) field(x); }
    void ext_vec(const std::vector<Ext2>& v) { u32((uint32_t)v.size()); for (auto& x : v) ext(x); }
    void raw64(u64 x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    // write_hash = GenericHashOut::to_bytes: canonical words of a HashOut, the 32 bytes of a BytesHash as they are
    void hash(const HashOut& h) { for (int i = 0; i < 4; i++) { if (get_hasher() == HASH_BLAKE3) raw64(h[i]); else field(h[i]); } }
    void cap(const std::vector<HashOut>& c) { u32((uint32_t)c.size()); for (auto& h : c) hash(h); }
    void merkle_proof(const std::vector<HashOut>& p) { u8((uint8_t)p.size()); for (auto& h : p) hash(h); }
    void opening_set(const StarkOpeningSet& s);
    void fri_proof(const FriProof& p);
};

// The tail of prove_single_table (circuits/src/stark/prover.rs:499-553) for three given commitments:
// draw zeta, open, observe, FRI.  Used to pin the opening/FRI path before the full STARK exists.
struct OpeningProof { Ext2 zeta; StarkOpeningSet openings; FriProof fri; };
OpeningProof open_and_prove(const PolynomialBatch& trace, const PolynomialBatch& zs, const PolynomialBatch& quotient,
                            int num_permutation_zs, Challenger& challenger, const FriConfig& cfg);
std::string verify_opening(const std::vector<std::vector<HashOut>>& caps, const std::vector<int>& num_polys,
                           int degree_bits, int num_permutation_zs, const StarkOpeningSet& openings,
                           const FriProof& fri, Challenger& challenger, const FriConfig& cfg);
}  // namespace ola_oracle
