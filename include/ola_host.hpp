// Host-side C++ mirror of the reference interfaces this backend replaces, over the C ABI of ola_gpu.h (header only, C++17,
// no HIP / torch types).  The reference is Rust; where its toolchain is missing this is the host layer a caller writes
// against -- same names, argument meaning and error behaviour as the Rust items cited at each declaration, so that code
// and tests read like the reference's own (tests/host_api_check.cpp).  INTEGRATION.md shows the Rust binding of the same
// entry points.
//
//   reference item                                                        here
//   plonky2/field/src/cfft/mod.rs:22,65,128,180  evaluate_poly ...        ola_host::fft::*
//   plonky2/plonky2/src/fri/oracle.rs:31-150     PolynomialBatch          ola_host::PolynomialBatch
//   plonky2/plonky2/src/iop/challenger.rs:19-170 Challenger               ola_host::Challenger
//   plonky2/plonky2/src/hash/poseidon.rs:593     PoseidonPermutation      ola_host::hash::permute / hash_no_pad / merkle_cap
//   plonky2/plonky2/src/fri/prover.rs:126        fri_proof_of_work        ola_host::fri_proof_of_work
//   circuits/src/stark/prover.rs:79              prove_with_traces        ola_host::prove_with_traces
//   circuits/src/stark/prover.rs:330             prove_single_table       ola_host::prove_single_table
//   circuits/src/stark/lookup.rs:68              permuted_cols            ola_host::permuted_cols
//   circuits/src/generation/poseidon.rs:5        generate_poseidon_trace  ola_host::generate_poseidon_trace
//
// Errors: the Rust code returns anyhow::Result / panics on contract violations; here every failing call throws
// ola_host::Error carrying the C ABI's code (OLA_E_*) and message.  Field elements are plain uint64_t words
// (GoldilocksField is #[repr(transparent)] u64); any representative is accepted on input, outputs are canonical.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ola_gpu.h"

namespace ola_host {

using F = uint64_t;                       // GoldilocksField
using HashOut = std::array<F, 4>;         // hash/hash_types.rs:19-21
using MerkleCap = std::vector<HashOut>;   // hash/merkle_tree.rs:20
using PolynomialValues = std::vector<F>;  // field/src/polynomial/mod.rs:20
using PolynomialCoeffs = std::vector<F>;  // field/src/polynomial/mod.rs:128

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int32_t rc) {
    if (rc != OLA_OK) throw Error(rc, std::string("ola_gpu error ") + std::to_string(rc) + ": " + ola_gpu_last_error());
}

inline uint32_t log2_strict(size_t n) {   // util/src/lib.rs log2_strict: panics unless n is a power of two
    uint32_t k = 0;
    while (((size_t)1 << k) < n) k++;
    if (((size_t)1 << k) != n || n == 0) throw Error(OLA_E_INVALID_ARG, "length is not a power of two");
    return k;
}

// FriConfig / StarkConfig::standard_fast_config (circuits/src/stark/config.rs:20-35) with the fields the backend reads.
struct StarkConfig {
    uint32_t rate_bits = 3, cap_height = 4, proof_of_work_bits = 16, fri_arity_bits = 4, fri_final_poly_bits = 5, num_query_rounds = 28,
             num_challenges = 2;
    // GenericConfig::Hasher (plonk/config.rs:112-161): OLA_HASH_POSEIDON = PoseidonGoldilocksConfig, OLA_HASH_BLAKE3 = Blake3GoldilocksConfig
    uint32_t hasher = OLA_HASH_POSEIDON;
    static StarkConfig standard_fast_config() { return StarkConfig{}; }
};

// One device context (replaces the reference's process-wide init_gpu()/GPU_LOCK, cfft/ntt/mod.rs:21-50).
class Gpu {
public:
    explicit Gpu(int32_t device = -1, const StarkConfig& c = StarkConfig::standard_fast_config(), void* stream = nullptr) : config(c) {
        OlaGpuConfig g{};
        g.device = device; g.stream = stream; g.rate_bits = c.rate_bits; g.cap_height = c.cap_height;
        g.proof_of_work_bits = c.proof_of_work_bits; g.fri_arity_bits = c.fri_arity_bits; g.fri_final_poly_bits = c.fri_final_poly_bits;
        g.num_query_rounds = c.num_query_rounds; g.num_challenges = c.num_challenges; g.hasher = c.hasher;
        check(ola_gpu_init(&g, &ctx_));
    }
    // One context spanning several GPUs of the node (ola_gpu_init_multi): prove_with_traces then runs on all of them in one call.
    explicit Gpu(const std::vector<int32_t>& devices, const StarkConfig& c = StarkConfig::standard_fast_config()) : config(c) {
        OlaGpuConfig g{};
        g.device = -1; g.stream = nullptr; g.rate_bits = c.rate_bits; g.cap_height = c.cap_height;
        g.proof_of_work_bits = c.proof_of_work_bits; g.fri_arity_bits = c.fri_arity_bits; g.fri_final_poly_bits = c.fri_final_poly_bits;
        g.num_query_rounds = c.num_query_rounds; g.num_challenges = c.num_challenges; g.hasher = c.hasher;
        check(ola_gpu_init_multi(&g, devices.data(), (uint32_t)devices.size(), &ctx_));
    }
    uint32_t device_count() const { uint32_t n = 0; check(ola_gpu_device_count(ctx_, &n)); return n; }
    ~Gpu() { if (ctx_) ola_gpu_free(ctx_); }
    Gpu(const Gpu&) = delete;
    Gpu& operator=(const Gpu&) = delete;
    OlaCtx* ctx() const { return ctx_; }
    void sync() const { check(ola_gpu_sync(ctx_)); }
    void trim() const { check(ola_gpu_trim(ctx_)); }
    const StarkConfig config;

private:
    OlaCtx* ctx_ = nullptr;
};

// ---- cfft/mod.rs: transforms of one polynomial (natural order in, natural order out) --------------------------------------
namespace fft {
inline std::vector<F> run(const Gpu& g, int32_t op, const std::vector<F>& in, F shift, uint32_t blowup_log) {
    const uint32_t log_n = log2_strict(in.size());
    const bool grows = op == OLA_NTT_COSET_LDE || op == OLA_NTT_COSET_LDE_LEAF_ORDER;
    std::vector<F> out(grows ? in.size() << blowup_log : in.size());
    check(ola_ntt_batch(g.ctx(), op, in.data(), out.data(), log_n, 1, shift, blowup_log));
    return out;
}
inline PolynomialValues evaluate_poly(const Gpu& g, const PolynomialCoeffs& coeffs) { return run(g, OLA_NTT_EVALUATE, coeffs, 7, 0); }               // :22
inline PolynomialCoeffs interpolate_poly(const Gpu& g, const PolynomialValues& values) { return run(g, OLA_NTT_INTERPOLATE, values, 7, 0); }        // :128
// coset_fft with zero padding to n << rate_bits (polynomial/mod.rs:253-270 lde + coset_fft; shift = F::coset_shift() = 7)
inline PolynomialValues evaluate_poly_with_offset(const Gpu& g, const PolynomialCoeffs& coeffs, F shift = 7, uint32_t rate_bits = 0) {              // :65
    return run(g, OLA_NTT_COSET_LDE, coeffs, shift, rate_bits);
}
inline PolynomialCoeffs interpolate_poly_with_offset(const Gpu& g, const PolynomialValues& values, F shift = 7) {                                    // :180
    return run(g, OLA_NTT_COSET_INTERPOLATE, values, shift, 0);
}
}  // namespace fft

// ---- hash/poseidon.rs, hashing.rs, merkle_tree.rs -----------------------------------------------------------------------------
namespace hash {
inline std::array<F, 12> permute(const Gpu& g, std::array<F, 12> state) {            // PoseidonPermutation::permute, poseidon.rs:593-603
    check(ola_poseidon_permute(g.ctx(), state.data(), 1));
    return state;
}
inline HashOut hash_no_pad(const Gpu& g, const std::vector<F>& input) {              // hashing.rs:84-111
    HashOut h{};
    if (input.empty()) {            // absorbing nothing leaves the zero state; the squeeze returns its first four words
        return h;
    }
    check(ola_hash_rows(g.ctx(), input.data(), 1, input.size(), h.data()));
    return h;
}
inline HashOut two_to_one(const Gpu& g, const HashOut& l, const HashOut& r) {        // hashing.rs:67-82 compress
    std::array<F, 12> s{l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    s = permute(g, s);
    return HashOut{s[0], s[1], s[2], s[3]};
}
// MerkleTree::new(leaves, cap_height).cap (merkle_tree/mod.rs:180-233); leaves row-major, all of one length
inline MerkleCap merkle_cap(const Gpu& g, const std::vector<std::vector<F>>& leaves, uint32_t cap_height) {
    if (leaves.empty()) throw Error(OLA_E_INVALID_ARG, "no leaves");
    const size_t len = leaves[0].size();
    std::vector<F> flat;
    flat.reserve(leaves.size() * len);
    for (const auto& l : leaves) {
        if (l.size() != len) throw Error(OLA_E_INVALID_ARG, "ragged leaves");
        flat.insert(flat.end(), l.begin(), l.end());
    }
    MerkleCap cap((size_t)1 << cap_height);
    check(ola_merkle_cap(g.ctx(), flat.data(), leaves.size(), len, cap_height, cap[0].data()));
    return cap;
}
// verify_merkle_proof_to_cap (merkle_proofs.rs:51-76), hashing on the device like everything else here
inline bool verify_merkle_proof_to_cap(const Gpu& g, const std::vector<F>& leaf_data, size_t leaf_index, const MerkleCap& cap,
                                       const std::vector<HashOut>& siblings) {
    HashOut cur = hash_no_pad(g, leaf_data);
    size_t idx = leaf_index;
    for (const HashOut& s : siblings) {
        cur = (idx & 1) ? two_to_one(g, s, cur) : two_to_one(g, cur, s);
        idx >>= 1;
    }
    return idx < cap.size() && cap[idx] == cur;
}
}  // namespace hash

// ---- fri/oracle.rs:31-150 ---------------------------------------------------------------------------------------------------------
// A committed batch of polynomials: coefficients, their low-degree extension and its Merkle tree stay in HBM.
class PolynomialBatch {
public:
    // from_values (oracle.rs:45-64): one vector per polynomial, all of the same power-of-two length.  rate_bits / cap_height
    // are the context's (StarkConfig); blinding is always false in this prover.
    static PolynomialBatch from_values(const Gpu& g, const std::vector<PolynomialValues>& values) { return make(g, values, false); }
    static PolynomialBatch from_coeffs(const Gpu& g, const std::vector<PolynomialCoeffs>& coeffs) { return make(g, coeffs, true); }   // :66-100

    PolynomialBatch(PolynomialBatch&& o) noexcept : g_(o.g_), h_(o.h_), cap_(std::move(o.cap_)), ncols_(o.ncols_), degree_log(o.degree_log), rate_bits(o.rate_bits) { o.h_ = nullptr; }
    PolynomialBatch(const PolynomialBatch&) = delete;
    ~PolynomialBatch() { if (h_) ola_batch_free(g_->ctx(), h_); }

    const MerkleCap& merkle_cap() const { return cap_; }                 // merkle_tree.cap
    size_t num_polynomials() const { return ncols_; }
    const OlaBatch* handle() const { return h_; }
    std::vector<PolynomialCoeffs> polynomials() const {                  // the `polynomials` field
        const size_t n = (size_t)1 << degree_log;
        std::vector<F> flat(ncols_ * n);
        check(ola_batch_get_coeffs(g_->ctx(), h_, flat.data()));
        std::vector<PolynomialCoeffs> out(ncols_);
        for (size_t c = 0; c < ncols_; c++) out[c].assign(flat.begin() + c * n, flat.begin() + (c + 1) * n);
        return out;
    }
    // get_lde_values(index, step) (oracle.rs:131-137): row `index * step` of the LDE in natural order of the coset
    std::vector<F> get_lde_values(size_t index, size_t step) const {
        std::vector<F> row(ncols_);
        check(ola_batch_get_lde_row(g_->ctx(), h_, index, step, row.data()));
        return row;
    }
    // merkle_tree.get(leaf_index) and merkle_tree.prove(leaf_index) (merkle_tree/mod.rs:235-308)
    std::pair<std::vector<F>, std::vector<HashOut>> leaf_with_proof(size_t leaf_index) const {
        const size_t depth = degree_log + rate_bits - g_->config.cap_height;
        std::vector<F> row(ncols_);
        std::vector<HashOut> sib(depth);
        check(ola_batch_get_leaf(g_->ctx(), h_, leaf_index, row.data(), depth ? sib[0].data() : nullptr));
        return {row, sib};
    }

private:
    PolynomialBatch(const Gpu* g, OlaBatch* h, MerkleCap cap, size_t ncols, uint32_t dl, uint32_t rb)
        : g_(g), h_(h), cap_(std::move(cap)), ncols_(ncols), degree_log(dl), rate_bits(rb) {}
    static PolynomialBatch make(const Gpu& g, const std::vector<std::vector<F>>& cols, bool coeffs) {
        if (cols.empty()) throw Error(OLA_E_INVALID_ARG, "empty batch");
        const uint32_t log_n = log2_strict(cols[0].size());
        std::vector<const F*> ptrs;
        for (const auto& c : cols) {
            if (c.size() != cols[0].size()) throw Error(OLA_E_INVALID_ARG, "polynomials of different lengths");
            ptrs.push_back(c.data());
        }
        MerkleCap cap((size_t)1 << g.config.cap_height);
        OlaBatch* h = nullptr;
        check((coeffs ? ola_commit_coeffs : ola_commit_values)(g.ctx(), ptrs.data(), (uint32_t)cols.size(), log_n, &h, cap[0].data()));
        return PolynomialBatch(&g, h, std::move(cap), cols.size(), log_n, g.config.rate_bits);
    }
    const Gpu* g_;
    OlaBatch* h_;
    MerkleCap cap_;
    size_t ncols_;

public:
    const uint32_t degree_log, rate_bits;
};

// ---- iop/challenger.rs:19-170 (duplex sponge over the hasher's permutation; host arithmetic, no device needed) -----------
class Challenger {
public:
    Challenger() { check(ola_challenger_init(&ch_)); }                                                        // :37
    explicit Challenger(const Gpu& g) { check(ola_challenger_init_hasher(&ch_, g.config.hasher)); }           // Challenger::<F, C::Hasher>::new()
    void observe_element(F e) { check(ola_challenger_observe(&ch_, &e, 1)); }                                  // :46
    void observe_elements(const std::vector<F>& es) { check(ola_challenger_observe(&ch_, es.data(), es.size())); }   // :64
    void observe_hash(const HashOut& h) { check(ola_challenger_observe_cap(&ch_, h.data(), 1)); }              // :79
    void observe_cap(const MerkleCap& cap) { for (const auto& h : cap) observe_hash(h); }                      // :83
    F get_challenge() { F c; check(ola_challenger_get(&ch_, &c, 1)); return c; }                               // :89
    std::vector<F> get_n_challenges(size_t n) { std::vector<F> v(n); for (auto& c : v) c = get_challenge(); return v; }   // :102
    HashOut get_hash() { HashOut h; for (auto& x : h) x = get_challenge(); return h; }                         // :106
    std::array<F, 2> get_extension_challenge() { F a = get_challenge(); F b = get_challenge(); return {a, b}; }   // :117
    std::array<F, 12> compact() { check(ola_challenger_compact(&ch_)); std::array<F, 12> s; for (int i = 0; i < 12; i++) s[i] = ch_.sponge_state[i]; return s; }   // :155
    OlaChallenger& raw() { return ch_; }

private:
    OlaChallenger ch_;
};

// ---- fri/prover.rs:126-156: smallest nonce whose hash with the transcript digest has `bits` leading zeros -------------------
inline F fri_proof_of_work(const Gpu& g, const HashOut& current_hash, uint32_t bits) {
    F w = 0;
    check(ola_pow(g.ctx(), current_hash.data(), bits, &w));
    return w;
}

// ---- circuits/src/stark/prover.rs:499-553 + fri/oracle.rs:167-241 + fri/prover.rs:20-204: the opening proof of one table ----------
// One call with the transcript inside the library (the fast path): returns {opening-set bytes, FriProof bytes}, `challenger` advances as
// the reference's does.
inline std::pair<std::vector<uint8_t>, std::vector<uint8_t>> open_and_prove(const Gpu& g, const PolynomialBatch& trace, const PolynomialBatch& zs,
                                                                            const PolynomialBatch& quotient, uint32_t num_permutation_zs,
                                                                            Challenger& challenger) {
    std::vector<uint8_t> out((size_t)1 << 20);
    size_t len = 0, olen = 0;
    auto call = [&]() {
        return ola_open_and_prove(g.ctx(), trace.handle(), zs.handle(), quotient.handle(), num_permutation_zs, &challenger.raw(), out.data(), out.size(), &len, &olen);
    };
    int32_t rc = call();
    if (rc == OLA_E_INVALID_ARG && len > out.size()) { out.resize(len); rc = call(); }
    check(rc);
    return {std::vector<uint8_t>(out.begin(), out.begin() + (long)olen), std::vector<uint8_t>(out.begin() + (long)olen, out.begin() + (long)len)};
}
// The same one step per call, for a caller that keeps the reference's loops and its own Challenger (ABI revision 7).
class FriSteps {
public:
    // StarkOpeningSet::new (proof.rs:198-233): the opening set in wire format (Buffer::read_opening_set reads it)
    FriSteps(const Gpu& g, const PolynomialBatch& trace, const PolynomialBatch& zs, const PolynomialBatch& quotient, uint32_t num_permutation_zs,
             const std::array<F, 2>& zeta)
        : g_(&g) {
        openings.resize((size_t)1 << 16);
        size_t len = 0;
        auto call = [&]() { return ola_open(g.ctx(), trace.handle(), zs.handle(), quotient.handle(), num_permutation_zs, zeta.data(), openings.data(), openings.size(), &len, &h_); };
        int32_t rc = call();
        if (rc == OLA_E_INVALID_ARG && len > openings.size()) { openings.resize(len); rc = call(); }
        check(rc);
        openings.resize(len);
        uint32_t n = 0, fl = 0;
        check(ola_fri_plan(h_, nullptr, 0, &n, &fl));
        reduction_arity_bits.resize(n);
        check(ola_fri_plan(h_, reduction_arity_bits.data(), n, &n, &fl));
        final_poly_len = fl;
    }
    FriSteps(const FriSteps&) = delete;
    ~FriSteps() { if (h_) ola_fri_free(h_); }
    void begin(const std::array<F, 2>& alpha) { check(ola_fri_commit_begin(h_, alpha.data())); }                       // fri/oracle.rs:178-219
    MerkleCap next_layer(const std::array<F, 2>* beta) {                                                               // fri/prover.rs:72-121, one turn
        MerkleCap cap((size_t)1 << g_->config.cap_height);
        check(ola_fri_commit_next_layer(h_, beta ? beta->data() : nullptr, cap[0].data()));
        return cap;
    }
    std::vector<std::array<F, 2>> finish(const std::array<F, 2>* beta) {                                               // prover.rs:114-119
        std::vector<std::array<F, 2>> poly(final_poly_len ? final_poly_len : 1);
        size_t n = 0;
        check(ola_fri_commit_finish(h_, beta ? beta->data() : nullptr, poly[0].data(), poly.size(), &n));
        poly.resize(n);
        return poly;
    }
    std::vector<uint8_t> query_rounds(const std::vector<F>& x_index) {                                                 // prover.rs:150-204, wire format
        std::vector<uint8_t> out((size_t)1 << 20);
        size_t len = 0;
        auto call = [&]() { return ola_fri_query(h_, x_index.data(), (uint32_t)x_index.size(), out.data(), out.size(), &len); };
        int32_t rc = call();
        if (rc == OLA_E_INVALID_ARG && len > out.size()) { out.resize(len); rc = call(); }
        check(rc);
        out.resize(len);
        return out;
    }
    std::vector<uint8_t> openings;                  // write_opening_set's bytes (serialization.rs:164-175)
    std::vector<uint32_t> reduction_arity_bits;     // fri_params.reduction_arity_bits
    size_t final_poly_len = 0;

private:
    const Gpu* g_;
    OlaFri* h_ = nullptr;
};

// ---- circuits/src/stark/prover.rs:79-327 --------------------------------------------------------------------------------------------
// `airset`: the tables and cross-table lookups as data (olavm_amd/air/dsl.py blob); traces[t]: table t column-major,
// width_t columns of 2^log_n[t] words; params: per-table public parameters in blob order; compress_challenges: one per
// table (0 where unused).  Returns the AllProof wire bytes (serialization.rs:377-393).
inline std::vector<uint8_t> prove_with_traces(const Gpu& g, const std::vector<F>& airset, const std::vector<std::vector<F>>& traces,
                                              const std::vector<uint32_t>& log_n, const std::vector<F>& params,
                                              const std::vector<F>& compress_challenges) {
    if (traces.size() != log_n.size()) throw Error(OLA_E_INVALID_ARG, "one height per trace");
    std::vector<const F*> ptrs;
    for (const auto& t : traces) ptrs.push_back(t.data());
    std::vector<uint8_t> out((size_t)8 << 20);
    size_t len = 0;
    int32_t rc = ola_prove_with_traces(g.ctx(), airset.data(), airset.size(), ptrs.data(), log_n.data(), params.empty() ? nullptr : params.data(),
                                       compress_challenges.empty() ? nullptr : compress_challenges.data(), out.data(), out.size(), &len);
    if (rc == OLA_E_INVALID_ARG && len > out.size()) {      // buffer too small: the finished proof waits in the context
        out.resize(len);
        rc = ola_take_pending_proof(g.ctx(), out.data(), out.size(), &len);
    }
    check(rc);
    out.resize(len);
    return out;
}

// prove_single_table (prover.rs:330-513) for callers that keep the reference's orchestration: `trace` = the table's columns,
// `commitment` = PolynomialBatch::from_values of them, ctl_challenges = num_challenges x {beta, gamma}; the shared challenger
// advances as in the reference.  Returns the table's StarkProof bytes (serialization.rs:349-358).
inline std::vector<uint8_t> prove_single_table(const Gpu& g, const std::vector<F>& airset, uint32_t table, const std::vector<PolynomialValues>& trace,
                                               const PolynomialBatch& commitment, const std::vector<std::array<F, 2>>& ctl_challenges,
                                               const std::vector<F>& params, Challenger& challenger) {
    std::vector<const F*> cols;
    for (const auto& c : trace) cols.push_back(c.data());
    std::vector<F> cap, ctl;
    for (const auto& h : commitment.merkle_cap()) cap.insert(cap.end(), h.begin(), h.end());
    for (const auto& c : ctl_challenges) { ctl.push_back(c[0]); ctl.push_back(c[1]); }
    std::vector<uint8_t> out((size_t)4 << 20);
    size_t len = 0;
    auto call = [&]() {
        return ola_prove_single_table(g.ctx(), airset.data(), airset.size(), table, cols.data(), commitment.handle(), cap.data(), ctl.data(),
                                      params.empty() ? nullptr : params.data(), &challenger.raw(), out.data(), out.size(), &len);
    };
    int32_t rc = call();
    if (rc == OLA_E_INVALID_ARG && len > out.size()) { out.resize(len); rc = call(); }
    check(rc);
    out.resize(len);
    return out;
}

// ---- trace-generation helpers -----------------------------------------------------------------------------------------------------
inline std::pair<std::vector<F>, std::vector<F>> permuted_cols(const Gpu& g, const std::vector<F>& inputs, const std::vector<F>& table) {   // lookup.rs:68
    if (inputs.size() != table.size()) throw Error(OLA_E_INVALID_ARG, "inputs and table differ in length");
    std::vector<F> pi(inputs.size()), pt(inputs.size());
    check(ola_permuted_cols(g.ctx(), inputs.data(), table.data(), inputs.size(), pi.data(), pt.data()));
    return {pi, pt};
}
// inputs: 12 x n column-major -> the 134 x n Poseidon table (generation/poseidon.rs:5-80)
inline std::vector<F> generate_poseidon_trace(const Gpu& g, const std::vector<F>& inputs, size_t n, const std::vector<F>& filters = {}) {
    if (inputs.size() != 12 * n || (!filters.empty() && filters.size() != 4 * n)) throw Error(OLA_E_INVALID_ARG, "shape");
    std::vector<F> out(134 * n);
    check(ola_generate_poseidon_trace(g.ctx(), inputs.data(), filters.empty() ? nullptr : filters.data(), n, out.data()));
    return out;
}

}  // namespace ola_host
