/* include/ola_gpu.h is a C header: this file is compiled as C99 with -Wall -Wextra -pedantic and linked against libola_gpu.so alone.
 * It is also the shortest complete host program of the boundary: the single-process flow of `ola prove` (client/src/main.rs:174-214)
 * on one or several GPUs -- create the context, start reserving buffers, (the caller produces its traces,) prove, free.  Without
 * arguments it only checks the ABI revision and the struct sizes and touches no device; with a fixture it proves it from
 * separately allocated columns (ola_prove_with_traces_cols) and from contiguous tables and compares. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ola_gpu.h"

static int check(int32_t rc, const char* what) {
    if (rc != OLA_OK) fprintf(stderr, "%s: error %d: %s\n", what, (int)rc, ola_gpu_last_error());
    return rc == OLA_OK;
}

/* The fixture of tests/test_gpu_host_api.py: u64 magic, u64 count, then `count` sections of (u64 length, words). */
typedef struct { uint64_t* w; size_t n; } Section;
static int read_fixture(const char* path, Section* sec, size_t max_sec, size_t* n_sec) {
    FILE* f = fopen(path, "rb");
    uint64_t head[2], len;
    size_t i;
    if (!f) return 0;
    if (fread(head, 8, 2, f) != 2 || head[0] != 0x4F4C41484F5354ull || head[1] > max_sec) { fclose(f); return 0; }
    for (i = 0; i < head[1]; i++) {
        if (fread(&len, 8, 1, f) != 1) { fclose(f); return 0; }
        sec[i].n = (size_t)len;
        sec[i].w = (uint64_t*)malloc(len ? len * 8 : 8);
        if (!sec[i].w || fread(sec[i].w, 8, (size_t)len, f) != (size_t)len) { fclose(f); return 0; }
    }
    *n_sec = (size_t)head[1];
    fclose(f);
    return 1;
}

/* The single-process flow of `ola prove` (client/src/main.rs:174-214) on one or several GPUs -- create the context, start reserving
 * buffers, (the caller produces its traces,) prove, free -- with the traces the way the reference holds them
 * (circuits/src/stark/prover.rs:79-83: per table a Vec of columns, EVERY COLUMN ITS OWN ALLOCATION): cols[t][c] are separately
 * malloc'ed copies of the fixture's columns, handed to ola_prove_with_traces_cols; the same proof is then asked for through
 * ola_prove_with_traces from the contiguous tables and the two byte strings must agree.  Returns the proof length or 0. */
static size_t prove_both_ways(uint32_t n_gpus, const uint64_t* airset, size_t airset_words, size_t n_tables, const uint64_t* const* traces,
                              const size_t* widths, const uint32_t* log_n, const uint64_t* params, const uint64_t* compress, uint8_t* out, size_t cap) {
    OlaGpuConfig cfg;
    OlaCtx* ctx = NULL;
    size_t len = 0, len2 = 0, t, c, ok = 1;
    uint8_t* out2 = (uint8_t*)malloc(cap);
    uint64_t*** cols = (uint64_t***)calloc(n_tables, sizeof *cols);
    double up[8];
    if (!out2 || !cols) return 0;
    for (t = 0; t < n_tables && ok; t++) {
        const size_t n = (size_t)1 << log_n[t];
        cols[t] = (uint64_t**)calloc(widths[t], sizeof **cols);
        for (c = 0; cols[t] && c < widths[t]; c++) {
            /* odd sizes in between so that neighbouring columns do not end up back to back */
            void* spacer = malloc(64 + 24 * ((t * 131 + c) % 17));
            cols[t][c] = (uint64_t*)malloc(n * 8);
            free(spacer);
            if (!cols[t][c]) { ok = 0; break; }
            memcpy(cols[t][c], traces[t] + c * n, n * 8);
        }
        if (!cols[t]) ok = 0;
    }
    if (!ok) { fprintf(stderr, "out of host memory\n"); return 0; }
    memset(&cfg, 0, sizeof cfg);                      /* every field is read: start from zero (ABI revision 3) */
    cfg.device = -1; cfg.rate_bits = 3; cfg.cap_height = 4; cfg.proof_of_work_bits = 16; cfg.fri_arity_bits = 4;
    cfg.fri_final_poly_bits = 5; cfg.num_query_rounds = 28; cfg.num_challenges = 2; cfg.hasher = OLA_HASH_POSEIDON;
    if (!check(ola_gpu_init_multi(&cfg, NULL, n_gpus, &ctx), "ola_gpu_init_multi")) return 0;
    if (n_gpus == 1) (void)ola_gpu_reserve(ctx, airset, airset_words, log_n);      /* optional: allocation off the proof's clock */
    if (!check(ola_prove_with_traces_cols(ctx, airset, airset_words, (const uint64_t* const* const*)cols, log_n, params, compress, out, cap, &len),
               "ola_prove_with_traces_cols")) {
        if (len > cap) fprintf(stderr, "the proof needs %zu bytes: call ola_take_pending_proof with a larger buffer\n", len);
        len = 0;
    }
    if (len && check(ola_gpu_upload_stats(ctx, up), "ola_gpu_upload_stats"))
        printf("upload: %.0f bytes of separately allocated columns, path %d, %d copier thread(s)\n", up[3], (int)up[4], (int)up[5]);
    if (len && !check(ola_prove_with_traces(ctx, airset, airset_words, traces, log_n, params, compress, out2, cap, &len2), "ola_prove_with_traces")) len = 0;
    if (len && (len != len2 || memcmp(out, out2, len) != 0)) { fprintf(stderr, "per-column and contiguous entry points produced different proofs\n"); len = 0; }
    /* a NULL column is refused, not dereferenced */
    if (len) {
        uint64_t* keep = cols[0][0];
        size_t l3 = 0;
        cols[0][0] = NULL;
        if (ola_prove_with_traces_cols(ctx, airset, airset_words, (const uint64_t* const* const*)cols, log_n, params, compress, out2, cap, &l3) != OLA_E_INVALID_ARG) {
            fprintf(stderr, "a NULL column pointer was accepted\n"); len = 0;
        }
        cols[0][0] = keep;
    }
    (void)ola_gpu_free(ctx);
    for (t = 0; t < n_tables; t++) { for (c = 0; c < widths[t]; c++) free(cols[t][c]); free(cols[t]); }
    free(cols); free(out2);
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
    if (argc < 3) { printf("c abi ok: revision %d\n", (int)rev); return 0; }
    /* host_c_abi_check <fixture> <proof out>: sections = Poseidon vectors (unused here), AIR set, log_n, params, compress challenges,
     * params per table (unused), then one column-major table per section (tests/test_gpu_host_api.py writes it) */
    {
        Section sec[64];
        size_t n_sec = 0, n_tables, t, len;
        const uint64_t* traces[58];
        size_t widths[58];
        uint32_t log_n[58];
        const size_t cap = (size_t)8 << 20;
        uint8_t* out = (uint8_t*)malloc(cap);
        FILE* f;
        if (!out || !read_fixture(argv[1], sec, 64, &n_sec) || n_sec < 7) { fprintf(stderr, "cannot read fixture %s\n", argv[1]); return 1; }
        n_tables = n_sec - 6;
        if (sec[2].n != n_tables) { fprintf(stderr, "fixture: %zu tables but %zu heights\n", n_tables, sec[2].n); return 1; }
        for (t = 0; t < n_tables; t++) {
            log_n[t] = (uint32_t)sec[2].w[t];
            traces[t] = sec[6 + t].w;
            widths[t] = sec[6 + t].n >> log_n[t];
        }
        len = prove_both_ways(1, sec[1].w, sec[1].n, n_tables, traces, widths, log_n, sec[3].w, sec[4].w, out, cap);
        if (!len) return 1;
        f = fopen(argv[2], "wb");
        if (!f || fwrite(out, 1, len, f) != len) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
        fclose(f);
        printf("c abi proof ok: %zu bytes, per-column == contiguous\n", len);
    }
    return 0;
}
