// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of the Fiat-Shamir challenger.
//
// Follows /root/reference/plonky2/plonky2/src/iop/challenger.rs:
//   :47-58    observe_element (clears buffered outputs; auto-duplex when 8 inputs are pending)
//   :86-100   get_challenge   (duplex if inputs pending or outputs exhausted; pops from the END of state[0..8])
//   :134-153  duplexing       (overwrite-mode absorb, permute, refill outputs with state[0..8])
//   :155-161  compact
#include "oracle.hpp"

namespace ola_oracle {

void Challenger::observe_element(u64 e) {
    output_buffer.clear();
    input_buffer.push_back(gl_canon(e));
    if (input_buffer.size() == 8) duplexing();
}
void Challenger::observe_elements(const u64* e, size_t n) {
    for (size_t i = 0; i < n; i++) observe_element(e[i]);
}
u64 Challenger::get_challenge() {
    if (!input_buffer.empty() || output_buffer.empty()) duplexing();
    u64 r = output_buffer.back();
    output_buffer.pop_back();
    return r;
}
void Challenger::duplexing() {
    for (size_t i = 0; i < input_buffer.size(); i++) sponge_state[i] = input_buffer[i];
    input_buffer.clear();
    if (hasher == HASH_BLAKE3) blake3_permutation(sponge_state);   // H::Permutation (challenger.rs:134-153 is generic over it)
    else poseidon_naive(sponge_state);
    output_buffer.assign(sponge_state, sponge_state + 8);
}
void Challenger::compact() {
    if (!input_buffer.empty()) duplexing();
    output_buffer.clear();
}

}  // namespace ola_oracle
