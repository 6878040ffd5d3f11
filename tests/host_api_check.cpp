// GPU check of the C++ host layer (include/ola_host.hpp): the reference's own unit tests for this path, restated against
// the mirror of its interfaces.  Built with g++ and run by tests/test_gpu_host_api.py, which also supplies the fixture
// file (Poseidon known-answer vectors, an AIR-set blob with traces) and compares the proof this program writes with the
// oracle's bytes.
//
//   fft_and_ifft / coset transforms      plonky2/field/src/fft.rs:218-252, polynomial/mod.rs:494-538
//   test_merkle_trees                    plonky2/plonky2/src/hash/merkle_tree/mod.rs:397 (every opened leaf verifies against the cap)
//   poseidon test_vectors                plonky2/plonky2/src/hash/poseidon_goldilocks.rs:281-314
//   no_duplicate_challenges              plonky2/plonky2/src/iop/challenger.rs:317-338
//   proof_of_work                        plonky2/plonky2/src/fri/prover.rs:126-156 + verifier check fri/verifier.rs:57-66
//   prove_openings / fri_proof loops     plonky2/plonky2/src/fri/oracle.rs:167-241, fri/prover.rs:20-204 (steps against the fused call)
//   prove_with_traces                    circuits/src/stark/prover.rs:79 (bytes are compared by the Python driver)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <set>

#include "ola_host.hpp"

using namespace ola_host;

static const F P = 0xFFFFFFFF00000001ULL;
static F mulmod(F a, F b) { return (F)((unsigned __int128)(a % P) * (b % P) % P); }
static F addmod(F a, F b) { return (F)(((unsigned __int128)(a % P) + (b % P)) % P); }
static F powmod(F b, F e) { F r = 1; while (e) { if (e & 1) r = mulmod(r, b); b = mulmod(b, b); e >>= 1; } return r; }
static F root_of_unity(uint32_t bits) { return powmod(powmod(7, (P - 1) >> 32), (F)1 << (32 - bits)); }   // goldilocks_field.rs:104-110
static F eval_naive(const std::vector<F>& c, F x) { F acc = 0; for (size_t i = c.size(); i-- > 0;) acc = addmod(mulmod(acc, x), c[i]); return acc; }

static int failures = 0;
#define EXPECT(cond)                                                                     \
    do {                                                                                 \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static std::vector<F> rand_vec(std::mt19937_64& rng, size_t n) {
    std::vector<F> v(n);
    for (auto& x : v) x = rng() % P;
    return v;
}

static std::vector<std::vector<F>> read_sections(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::printf("cannot open %s\n", path); std::exit(2); }
    auto rd = [&]() { F w = 0; f.read(reinterpret_cast<char*>(&w), 8); return w; };
    const F magic = rd(), count = rd();
    if (magic != 0x4F4C41484F5354ULL) { std::printf("bad fixture magic\n"); std::exit(2); }
    std::vector<std::vector<F>> s(count);
    for (auto& sec : s) { sec.resize(rd()); f.read(reinterpret_cast<char*>(sec.data()), (std::streamsize)sec.size() * 8); }
    return s;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: host_api_check <fixture> <proof-out>\n"); return 2; }
    const auto sec = read_sections(argv[1]);
    std::mt19937_64 rng(20240927);
    Gpu gpu;

    // ---- fft_and_ifft; values against direct evaluation
    for (uint32_t log_n : {0u, 1u, 5u, 10u, 15u}) {
        const size_t n = (size_t)1 << log_n;
        const auto coeffs = rand_vec(rng, n);
        const auto values = fft::evaluate_poly(gpu, coeffs);
        EXPECT(fft::interpolate_poly(gpu, values) == coeffs);
        const F w = root_of_unity(log_n);
        for (size_t i : {(size_t)0, n / 3, n - 1}) EXPECT(values[i] == eval_naive(coeffs, powmod(w, i)));
        // test_coset_fft / test_coset_ifft with the field's coset shift and with an arbitrary one
        for (F shift : {(F)7, (F)0x123456789ABCDEFULL}) {
            const auto cv = fft::evaluate_poly_with_offset(gpu, coeffs, shift);
            EXPECT(fft::interpolate_poly_with_offset(gpu, cv, shift) == coeffs);
            for (size_t i : {(size_t)0, n / 2, n - 1}) EXPECT(cv[i] == eval_naive(coeffs, mulmod(shift, powmod(w, i))));
        }
        // lde(rate_bits).coset_fft(shift): the evaluations PolynomialBatch commits to
        const auto lde = fft::evaluate_poly_with_offset(gpu, coeffs, 7, 3);
        EXPECT(lde.size() == 8 * n);
        const F w8 = root_of_unity(log_n + 3);
        for (size_t i : {(size_t)1, 4 * n + 1, 8 * n - 1}) EXPECT(lde[i] == eval_naive(coeffs, mulmod(7, powmod(w8, i))));
    }

    // ---- Poseidon test_vectors
    EXPECT(sec[0].size() == 4 * 24);
    for (size_t v = 0; v + 24 <= sec[0].size(); v += 24) {
        std::array<F, 12> in, want;
        for (int i = 0; i < 12; i++) { in[i] = sec[0][v + i]; want[i] = sec[0][v + 12 + i]; }
        EXPECT(hash::permute(gpu, in) == want);
    }
    {   // hash_no_pad of up to 8 words is one permutation of the zero-extended input (hashing.rs:84-111)
        const auto x = rand_vec(rng, 8);
        std::array<F, 12> s{};
        for (int i = 0; i < 8; i++) s[i] = x[i];
        s = hash::permute(gpu, s);
        EXPECT((hash::hash_no_pad(gpu, x) == HashOut{s[0], s[1], s[2], s[3]}));
    }

    // ---- PolynomialBatch::from_values: coefficients, LDE rows, test_merkle_trees
    {
        const uint32_t log_n = 7;
        const size_t n = (size_t)1 << log_n;
        std::vector<PolynomialValues> cols;
        for (int c = 0; c < 5; c++) cols.push_back(rand_vec(rng, n));
        const auto batch = PolynomialBatch::from_values(gpu, cols);
        EXPECT(batch.merkle_cap().size() == 16 && batch.num_polynomials() == 5 && batch.degree_log == log_n);
        const auto polys = batch.polynomials();
        for (int c = 0; c < 5; c++) EXPECT(polys[c] == fft::interpolate_poly(gpu, cols[c]));
        const auto lde0 = fft::evaluate_poly_with_offset(gpu, polys[0], 7, 3);
        for (size_t i : {(size_t)0, (size_t)77, 8 * n - 1}) EXPECT(batch.get_lde_values(i, 1)[0] == lde0[i]);
        for (size_t leaf = 0; leaf < 8 * n; leaf += 37) {
            const auto [row, proof] = batch.leaf_with_proof(leaf);
            EXPECT(proof.size() == log_n + 3 - 4);
            EXPECT(hash::verify_merkle_proof_to_cap(gpu, row, leaf, batch.merkle_cap(), proof));
            auto wrong = row;
            wrong[2] ^= 1;
            EXPECT(!hash::verify_merkle_proof_to_cap(gpu, wrong, leaf, batch.merkle_cap(), proof));
        }
        // from_coeffs of the interpolated polynomials commits to the same tree
        EXPECT(PolynomialBatch::from_coeffs(gpu, polys).merkle_cap() == batch.merkle_cap());
        // contract violations surface as errors, not as wrong answers
        bool threw = false;
        try { cols[1].pop_back(); (void)PolynomialBatch::from_values(gpu, cols); } catch (const Error& e) { threw = e.code == OLA_E_INVALID_ARG; }
        EXPECT(threw);
    }

    // ---- Challenger: no_duplicate_challenges, and the transcript is a function of what was observed
    {
        Challenger a, b;
        a.observe_elements(rand_vec(rng, 5));
        std::set<F> seen;
        for (int i = 0; i < 50; i++) { for (F c : a.get_n_challenges(7)) seen.insert(c); a.observe_element((F)i); }
        EXPECT(seen.size() == 50 * 7);
        Challenger c1, c2;
        c1.observe_hash(HashOut{1, 2, 3, 4});
        c2.observe_elements({1, 2, 3, 4});
        EXPECT(c1.get_extension_challenge() == c2.get_extension_challenge());
        c2.observe_element(5);
        EXPECT(c1.get_challenge() != c2.get_challenge());
        (void)b;
    }

    // ---- fri_proof_of_work: the witness makes the leading zeros appear (fri/verifier.rs:57-66)
    {
        const HashOut h{11, 22, 33, 44};
        const uint32_t bits = 12;
        const F wit = fri_proof_of_work(gpu, h, bits);
        const auto out = hash::permute(gpu, {h[0], h[1], h[2], h[3], wit, 0, 0, 0, 0, 0, 0, 0});   // hash_no_pad([h.., witness])
        EXPECT((out[0] >> (64 - bits)) == 0);                                                       // leading_zeros >= bits
    }

    // ---- the opening proof one step per call with this side's Challenger (the reference's prove_openings / fri_proof loops,
    //      fri/oracle.rs:167-241, fri/prover.rs:20-204) against the fused call: same bytes, same transcript
    for (uint32_t log_n : {6u, 13u}) {
        const size_t n = (size_t)1 << log_n;
        std::vector<PolynomialValues> tv(5), zv(3);
        std::vector<PolynomialCoeffs> qc(2);
        for (auto& c : tv) c = rand_vec(rng, n);
        for (auto& c : zv) c = rand_vec(rng, n);
        for (auto& c : qc) c = rand_vec(rng, n);
        const auto bt = PolynomialBatch::from_values(gpu, tv), bz = PolynomialBatch::from_values(gpu, zv);
        const auto bq = PolynomialBatch::from_coeffs(gpu, qc);
        Challenger fused(gpu);
        for (const auto* b : {&bt, &bz, &bq}) fused.observe_cap(b->merkle_cap());
        Challenger mine = fused;
        const auto want = open_and_prove(gpu, bt, bz, bq, 1, fused);
        auto words = [](const std::vector<uint8_t>& b, size_t& at) {      // read_field_ext_vec / read_field_vec: u32 count, then little-endian words
            uint32_t k; std::memcpy(&k, b.data() + at, 4); at += 4;
            return k;
        };
        FriSteps st(gpu, bt, bz, bq, 1, mine.get_extension_challenge());
        EXPECT(st.openings == want.first);
        {   // observe_openings in to_fri_openings order (proof.rs:235-265): local, zs, quotient, next, zs next, ctl_zs_last as extension elements
            std::vector<std::vector<F>> vec;
            size_t at = 0;
            for (int part = 0; part < 6; part++) {
                const uint32_t k = words(st.openings, at);
                const size_t nw = part == 4 ? k : 2 * (size_t)k;
                std::vector<F> v(nw);
                std::memcpy(v.data(), st.openings.data() + at, nw * 8);
                at += nw * 8;
                vec.push_back(v);
            }
            for (int part : {0, 2, 5, 1, 3}) mine.observe_elements(vec[part]);
            for (F x : vec[4]) { mine.observe_element(x); mine.observe_element(0); }
        }
        st.begin(mine.get_extension_challenge());
        std::vector<MerkleCap> caps;
        std::array<F, 2> beta{};
        for (size_t i = 0; i < st.reduction_arity_bits.size(); i++) {
            caps.push_back(st.next_layer(i ? &beta : nullptr));
            mine.observe_cap(caps.back());
            beta = mine.get_extension_challenge();
        }
        const auto final_poly = st.finish(caps.empty() ? nullptr : &beta);
        for (const auto& e : final_poly) { mine.observe_element(e[0]); mine.observe_element(e[1]); }
        const F witness = fri_proof_of_work(gpu, mine.get_hash(), gpu.config.proof_of_work_bits);
        std::vector<F> xs(gpu.config.num_query_rounds);
        for (auto& x : xs) x = mine.get_challenge() % ((F)n << gpu.config.rate_bits);
        const auto rounds = st.query_rounds(xs);
        std::vector<uint8_t> got;
        auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) got.push_back((uint8_t)(v >> (8 * i))); };
        auto put64 = [&](F v) { for (int i = 0; i < 8; i++) got.push_back((uint8_t)(v >> (8 * i))); };
        put32((uint32_t)caps.size());
        for (const auto& cap : caps) { put32((uint32_t)cap.size()); for (const auto& h : cap) for (F x : h) put64(x); }
        got.insert(got.end(), rounds.begin(), rounds.end());
        put32((uint32_t)final_poly.size());
        for (const auto& e : final_poly) { put64(e[0]); put64(e[1]); }
        put64(witness);
        EXPECT(got == want.second);
        EXPECT(mine.get_challenge() == fused.get_challenge());
    }

    // ---- permuted_cols: a permutation of both columns, and Halo2's rule on a valid lookup
    {
        const size_t n = 1000;
        std::vector<F> table(n), inputs(n);
        for (size_t i = 0; i < n; i++) { table[i] = i; inputs[i] = rng() % 300; }
        const auto [pi, pt] = permuted_cols(gpu, inputs, table);
        auto si = inputs, st = pt;
        std::sort(si.begin(), si.end());
        std::sort(st.begin(), st.end());
        EXPECT(pi == si && st == table);
        for (size_t i = 0; i < n; i++) EXPECT(pi[i] == pt[i] || (i > 0 && pi[i] == pi[i - 1]));
    }

    // ---- prove_with_traces on the fixture's AIR set; the Python driver compares the bytes with the oracle's
    {
        const auto& airset = sec[1];
        std::vector<uint32_t> log_n(sec[2].begin(), sec[2].end());
        const auto& n_params = sec[5];
        std::vector<std::vector<F>> traces(sec.begin() + 6, sec.end());
        EXPECT(traces.size() == log_n.size() && n_params.size() == log_n.size());
        const auto proof = prove_with_traces(gpu, airset, traces, log_n, sec[3], sec[4]);
        EXPECT(proof.size() > 1000 && proof == prove_with_traces(gpu, airset, traces, log_n, sec[3], sec[4]));
        std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char*>(proof.data()), (std::streamsize)proof.size());
        bool threw = false;
        try { auto bad = airset; bad[0] ^= 1; (void)prove_with_traces(gpu, bad, traces, log_n, sec[3], sec[4]); } catch (const Error&) { threw = true; }
        EXPECT(threw);
        {   // the same call on ONE context that spans two (logical) GPUs: same bytes (ola_gpu_init_multi, SURVEY 8(b) Threading)
            Gpu two(std::vector<int32_t>{0, 0});
            EXPECT(two.device_count() == 2);
            EXPECT(prove_with_traces(two, airset, traces, log_n, sec[3], sec[4]) == proof);
        }

        // the same proof with the reference's own orchestration (prover.rs:79-327): commit every trace, observe the caps, draw
        // the CTL challenges, prove table by table on the shared challenger, concatenate
        std::vector<PolynomialBatch> commitments;
        std::vector<std::vector<PolynomialValues>> columns;
        for (size_t t = 0; t < traces.size(); t++) {
            const size_t n = (size_t)1 << log_n[t], ncols = traces[t].size() / n;
            std::vector<PolynomialValues> cols(ncols);
            for (size_t c = 0; c < ncols; c++) cols[c].assign(traces[t].begin() + c * n, traces[t].begin() + (c + 1) * n);
            commitments.push_back(PolynomialBatch::from_values(gpu, cols));
            columns.push_back(std::move(cols));
        }
        Challenger challenger;
        for (const auto& c : commitments) challenger.observe_cap(c.merkle_cap());
        std::vector<std::array<F, 2>> ctl_challenges;
        for (int c = 0; c < 2; c++) { const F beta = challenger.get_challenge(), gamma = challenger.get_challenge(); ctl_challenges.push_back({beta, gamma}); }
        auto put_u32 = [](std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); };
        std::vector<uint8_t> assembled;
        put_u32(assembled, (uint32_t)traces.size());
        size_t poff = 0;
        for (size_t t = 0; t < traces.size(); t++) {
            const std::vector<F> params(sec[3].begin() + poff, sec[3].begin() + poff + n_params[t]);
            poff += n_params[t];
            const auto part = prove_single_table(gpu, airset, (uint32_t)t, columns[t], commitments[t], ctl_challenges, params, challenger);
            assembled.insert(assembled.end(), part.begin(), part.end());
        }
        put_u32(assembled, (uint32_t)traces.size());
        for (size_t t = 0; t < traces.size(); t++) for (int i = 0; i < 8; i++) assembled.push_back((uint8_t)(sec[4][t] >> (8 * i)));
        EXPECT(assembled == proof);
    }

    std::printf(failures ? "host_api_check: %d FAILURES\n" : "host_api_check: all checks passed\n", failures);
    return failures ? 1 : 0;
}
