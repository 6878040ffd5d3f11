/* include/ola_gpu.h is a C header: this file is compiled as C99 with -Wall -Wextra -pedantic and linked against libola_gpu.so alone.
 * It is also the shortest complete host program of the boundary: the single-process flow of `ola prove` (client/src/main.rs:174-214)
 * on one or several GPUs -- create the context, start reserving buffers, (the caller produces its traces,) prove, free.  Without
 * arguments it only checks the ABI revision and the struct sizes and touches no device. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ola_gpu.h"

static int check(int32_t rc, const char* what) {
    if (rc != OLA_OK) fprintf(stderr, "%s: error %d: %s\n", what, (int)rc, ola_gpu_last_error());
    return rc == OLA_OK;
}

/* airset: the blob of include/ola_airset.bin; traces[t]: column-major table t of 2^log_n[t] rows; returns the proof length or 0 */
static size_t prove_once(uint32_t n_gpus, const uint64_t* airset, size_t airset_words, const uint64_t* const* traces, const uint32_t* log_n,
                         const uint64_t* params, const uint64_t* compress, uint8_t* out, size_t cap) {
    OlaGpuConfig cfg;
    OlaCtx* ctx = NULL;
    size_t len = 0;
    memset(&cfg, 0, sizeof cfg);                      /* every field is read: start from zero (ABI revision 3) */
    cfg.device = -1; cfg.rate_bits = 3; cfg.cap_height = 4; cfg.proof_of_work_bits = 16; cfg.fri_arity_bits = 4;
    cfg.fri_final_poly_bits = 5; cfg.num_query_rounds = 28; cfg.num_challenges = 2; cfg.hasher = OLA_HASH_BLAKE3;
    if (!check(ola_gpu_init_multi(&cfg, NULL, n_gpus, &ctx), "ola_gpu_init_multi")) return 0;
    if (n_gpus == 1) (void)ola_gpu_reserve(ctx, airset, airset_words, log_n);      /* optional: allocation off the proof's clock */
    if (!check(ola_prove_with_traces(ctx, airset, airset_words, traces, log_n, params, compress, out, cap, &len), "ola_prove_with_traces")) {
        if (len > cap) fprintf(stderr, "the proof needs %zu bytes: call ola_take_pending_proof with a larger buffer\n", len);
        len = 0;
    }
    (void)ola_gpu_free(ctx);
    return len;
}

int main(int argc, char** argv) {
    size_t chal = 0, conf = 0;
    const int32_t rev = ola_gpu_abi_version(&chal, &conf);
    if (rev != OLA_GPU_ABI_VERSION || chal != sizeof(OlaChallenger) || conf != sizeof(OlaGpuConfig)) {
        fprintf(stderr, "header revision %d (OlaChallenger %zu, OlaGpuConfig %zu) but library revision %d (%zu, %zu)\n", OLA_GPU_ABI_VERSION,
                sizeof(OlaChallenger), sizeof(OlaGpuConfig), (int)rev, chal, conf);
        return 1;
    }
    /* host-only entry points work without a device */
    {
        OlaChallenger ch;
        uint64_t d[4] = {1, 2, 3, 4}, x[2];
        if (!check(ola_challenger_init_hasher(&ch, OLA_HASH_POSEIDON), "ola_challenger_init_hasher")) return 1;
        if (!check(ola_challenger_observe_cap(&ch, d, 1), "ola_challenger_observe_cap")) return 1;
        if (!check(ola_challenger_get(&ch, x, 2), "ola_challenger_get")) return 1;
    }
    (void)argv;
    if (argc < 2) { printf("c abi ok: revision %d\n", (int)rev); return 0; }
    (void)prove_once;                                  /* the flow above is exercised from the Python and C++ suites on the GPU box */
    return 0;
}
