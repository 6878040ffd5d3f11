// Host-side Fiat-Shamir challenger + device-side opening / FRI pipeline.
//
// Challenger: plonky2/plonky2/src/iop/challenger.rs:36-162 (overwrite duplex, rate 8, pops from the back).
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/ola_gpu.h"
#include "device_ctx.h"
#include "gl.cuh"
#include "poseidon_host.h"

namespace ola {

static void challenger_duplex(OlaChallenger& ch) {
    for (uint32_t i = 0; i < ch.input_len; i++) ch.sponge_state[i] = ch.input_buffer[i];
    ch.input_len = 0;
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = ch.sponge_state[i];
    poseidon_permute_host(s);
    for (int i = 0; i < 12; i++) ch.sponge_state[i] = s[i];
    for (int i = 0; i < 8; i++) ch.output_buffer[i] = s[i];
    ch.output_len = 8;
}
void challenger_observe(OlaChallenger& ch, const u64* e, size_t n) {
    for (size_t i = 0; i < n; i++) {
        ch.output_len = 0;
        ch.input_buffer[ch.input_len++] = gl_canon(e[i]);
        if (ch.input_len == 8) challenger_duplex(ch);
    }
}
u64 challenger_get(OlaChallenger& ch) {
    if (ch.input_len != 0 || ch.output_len == 0) challenger_duplex(ch);
    return ch.output_buffer[--ch.output_len];
}
void challenger_compact(OlaChallenger& ch) {
    if (ch.input_len != 0) challenger_duplex(ch);
    ch.output_len = 0;
}

void open_and_prove(DeviceCtx*, NttTables&, const OlaGpuConfig&, const OlaBatch&, const OlaBatch&, const OlaBatch&,
                    uint32_t, OlaChallenger&, std::vector<uint8_t>&, size_t&) {
    throw OlaError(OLA_E_INTERNAL, "ola_open_and_prove: not implemented yet");
}

}  // namespace ola
