/* ola_gpu.h -- C ABI of the MI355X-native Goldilocks STARK proving backend for OlaVM.
 *
 * This is the drop-in boundary behind circuits::stark::prover::prove_with_traces (reference
 * circuits/src/stark/prover.rs:79).  It widens the reference's existing GPU FFI precedent
 *     extern "C" { gpu_init; gpu_method; gpu_free }      plonky2/field/src/cfft/ntt/mod.rs:21-45
 * (one column per call, host<->device round trip per column, global mutex) to whole-phase granularity:
 * a PolynomialBatch is committed in one call and stays resident in HBM.  INTEGRATION.md shows the Rust-side
 * bindings (`extern "C"` block + build.rs link lines) a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int32_t: 0 = OLA_OK, negative = error; ola_gpu_last_error() gives the message of
 *     the last failure on the calling thread.  No exceptions, no abort() cross this boundary
 *     (the reference panics / returns anyhow::Error: prover.rs:352-355,471-473,508-511).
 *   - field elements are uint64_t Goldilocks values (p = 2^64 - 2^32 + 1), GoldilocksField is
 *     #[repr(transparent)] u64 (plonky2/field/src/goldilocks_field.rs:24-26).  Inputs may be non-canonical
 *     (>= p); ALL outputs are canonical (< p), i.e. exactly what the reference serialises
 *     (circuits/src/stark/serialization.rs:50-52).
 *   - "host" pointers are ordinary process memory owned by the caller and are not retained after return.
 *     "_dev" entry points take device (HBM) pointers valid on the context's device; work is enqueued on the
 *     context's stream and the call returns after the stream has been synchronised unless stated otherwise.
 *   - column-major: a table of `ncols` columns of n = 2^log_n rows is ncols contiguous runs of n elements
 *     (the reference's Vec<PolynomialValues<F>>, plonky2/field/src/polynomial/mod.rs:24-26).
 *   - thread-compatible per OlaCtx: one orchestrating thread per context (prove_with_traces is sequential
 *     across tables, prover.rs:160-287).
 */
#ifndef OLA_GPU_H
#define OLA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision of this header.  3: OlaChallenger carries `hasher` + `reserved` (sizeof 240, was 232 in revision 1) and
 * OlaGpuConfig.hasher occupies what was tail padding (sizeof unchanged, but the field must be set: zero-initialise the struct);
 * multi-device contexts and the accounting entry points were added.  A host built against an older header must be rebuilt:
 * ola_gpu_abi_version() lets it check at start-up that library and header agree (also on sizeof(OlaChallenger)).
 * 5: ola_prove_with_traces_cols (one pointer per column: the reference's [Vec<PolynomialValues<F>>; NUM_TABLES] as it is),
 * ola_gpu_scope_times (the `timed!` scopes with device times, for the caller's TimingTree), ola_gpu_upload_stats; no struct of
 * revision 4 changed.
 * 6: ola_gpu_warmup / ola_gpu_warmup_wait (start-up work ahead of the first context, where the reference calls init_gpu()),
 * ola_gpu_ntt_pass_times (the transform passes one by one, for the dominant kernel's roofline); no struct changed. */
#define OLA_GPU_ABI_VERSION 7
#define OLA_OK 0
#define OLA_E_INVALID_ARG (-1)
#define OLA_E_NO_DEVICE (-2)
#define OLA_E_OOM (-3)
#define OLA_E_QUOTIENT_DEGREE (-4) /* prover.rs:469-473 "vanishing polynomial is not divisible by Z_H" */
#define OLA_E_HIP (-5)
#define OLA_E_ZETA_IN_SUBGROUP (-6) /* prover.rs:508-511 */
#define OLA_E_INTERNAL (-7)

typedef struct OlaCtx OlaCtx;     /* device context: streams, twiddle tables, Poseidon constants            */
typedef struct OlaBatch OlaBatch; /* a committed PolynomialBatch resident in HBM (fri/oracle.rs:31-39)      */

/* StarkConfig::standard_fast_config (circuits/src/stark/config.rs:18-30) is the default when NULL is passed. */
typedef struct OlaGpuConfig {
    int32_t device;             /* HIP device ordinal; -1 = current device                                   */
    void* stream;               /* hipStream_t to run on, NULL = the library creates its own                 */
    uint32_t rate_bits;         /* 3                                                                         */
    uint32_t cap_height;        /* 4                                                                         */
    uint32_t proof_of_work_bits;/* 16                                                                        */
    uint32_t fri_arity_bits;    /* ConstantArityBits(4, 5)                                                   */
    uint32_t fri_final_poly_bits;
    uint32_t num_query_rounds;  /* 28                                                                        */
    uint32_t num_challenges;    /* 2                                                                         */
    uint32_t hasher;            /* GenericConfig::Hasher of the Merkle trees and the challenger (plonk/config.rs:112-161):
                                 * OLA_HASH_POSEIDON = PoseidonGoldilocksConfig (0, the default),
                                 * OLA_HASH_BLAKE3 = Blake3GoldilocksConfig.  InnerHasher (proof of work) is Poseidon in both.
                                 * The field takes the place of the struct's tail padding: sizeof is unchanged.             */
} OlaGpuConfig;
#define OLA_HASH_POSEIDON 0u
#define OLA_HASH_BLAKE3 1u

/* ---- lifetime (replaces gpu_init / gpu_free, cfft/ntt/mod.rs:89-121 and core/src/storage/db.rs:248) ---- */
int32_t ola_gpu_init(const OlaGpuConfig* cfg, OlaCtx** out_ctx);
/* OLA_GPU_ABI_VERSION of the library that was loaded; *challenger_size / *config_size (may be NULL) receive its sizeof(OlaChallenger)
 * and sizeof(OlaGpuConfig). */
int32_t ola_gpu_abi_version(size_t* challenger_size, size_t* config_size);
/* One context that spans n_devices GPUs of the node (1, 2, 4 or 8; devices = HIP ordinals, NULL = 0..n-1): the caller stays the
 * single process the reference's prover is (client/src/main.rs:174-214, one `prove` per process; the reference's own GPU state is
 * process-wide, cfft/ntt/mod.rs:14-17,48-50) and calls ola_prove_with_traces ONCE; inside, rank r is a worker thread on
 * devices[r] with its own stream and buffer pool, the proof runs on the coset partition described under ola_set_shard below, and
 * the exchanges are the library's own all-gather over xGMI (peer-to-peer pulls ordered by events on the ranks' streams, no host
 * synchronisation; olavm_amd/csrc/peer_group.h).  Every other entry point of such a context works on devices[0] as on a
 * single-device context.  Entries of devices[] may repeat (logical ranks sharing a GPU: how the one-GPU test box exercises the
 * path).  cfg->device is ignored, cfg->stream must be NULL when n_devices > 1.  Needs peer access between the devices
 * (OLA_E_HIP otherwise). */
int32_t ola_gpu_init_multi(const OlaGpuConfig* cfg, const int32_t* devices, uint32_t n_devices, OlaCtx** out_ctx);
int32_t ola_gpu_device_count(OlaCtx* ctx, uint32_t* n_devices);
/* Who carries the exchanges of a multi-device context (SURVEY 8(b),(e): "RCCL-over-xGMI all-gather of Merkle caps and FRI
 * commitments").  OLA_COLLECTIVE = peer (default) | rccl, read by ola_gpu_init_multi: with rccl the library dlopen()s librccl.so,
 * creates one communicator per device (ncclCommInitAll over the context's devices) and every exchange is an ncclAllGather on the
 * rank's stream; with peer it is the library's own event-ordered pulls (hipMemcpyPeerAsync).  RCCL is never a link dependency.
 * When RCCL cannot carry the context -- library absent, or logical ranks aliased onto one physical GPU -- the context keeps the
 * peer carrier and `note` says why.  *carrier: OLA_COLLECTIVE_*, *ranks: ranks the carrier spans; note (may be NULL) receives a
 * NUL-terminated explanation (RCCL version when active, the refusal otherwise). */
#define OLA_COLLECTIVE_NONE 0u /* single-device context: no exchanges */
#define OLA_COLLECTIVE_PEER 1u
#define OLA_COLLECTIVE_RCCL 2u
int32_t ola_gpu_collective(OlaCtx* ctx, uint32_t* carrier, uint32_t* ranks, char* note, size_t note_cap);
/* The context's all-gather by itself (what bench.py times for the 512-byte cap exchange and the trace exchange, and what the
 * tests check): every rank contributes bytes_per_rank bytes of its own pattern, `reps` gathers run back to back on the ranks'
 * streams through `carrier` (OLA_COLLECTIVE_PEER / _RCCL; the latter needs a context created under OLA_COLLECTIVE=rccl, also a
 * single-device one: a 1-rank communicator).  *ms_per_gather: slowest rank's mean; *mismatches: wrong bytes over all ranks (0). */
int32_t ola_gpu_all_gather_check(OlaCtx* ctx, uint32_t carrier, size_t bytes_per_rank, uint32_t reps, double* ms_per_gather, uint64_t* mismatches);
int32_t ola_gpu_free(OlaCtx* ctx);
/* Start-up ahead of the first context: replaces the reference's early hook -- OlaStark::default() calls
 * plonky2::field::cfft::ntt::init_gpu() (circuits/src/stark/ola_stark.rs:47, plonky2/field/src/cfft/ntt/mod.rs:53-99) before
 * prove() generates the traces (client/src/main.rs:193-195).  Returns at once; a helper thread starts the HIP runtime, opens
 * `device` (-1: the current one), loads the library's code objects (the main one and one per generated quotient kernel) and, with
 * OLA_WARMUP_PINNED_RING, pins the 128 MB staging ring of the trace upload.  With an AIR set (may be NULL; copied) it also creates
 * a context with the default configuration and PRIMES it: one throw-away proof per hash configuration of an all-zero instance with
 * small tables (divisibility check off, bytes discarded), so that every kernel of the proof path has been launched once, the
 * transform tables exist and the upload path has carried data -- what otherwise makes the first proof of a process 1.2 - 1.4 x a
 * warm one.  None of this depends on the StarkConfig or the hasher, which the caller does not know yet at that point:
 * ola_gpu_init / ola_gpu_init_multi wait for a warm-up that is under way and, when cfg->stream is NULL and the device is the
 * warmed one, take the primed context over (configuration and hasher are set then); otherwise they find at least the runtime up
 * (5 ms instead of 160 ms: the split is printed under OLA_TIMING=1).  Calling it again is a no-op; a failure inside the thread
 * (no device) is reported by the ola_gpu_init that follows, as it would have been without the warm-up. */
#define OLA_WARMUP_PINNED_RING 1u
int32_t ola_gpu_warmup(int32_t device, uint32_t flags, const uint64_t* airset, size_t airset_words);
/* Waits for the warm-up thread; *ms_out (may be NULL) = how long it ran.  OLA_E_INVALID_ARG when ola_gpu_warmup was never called,
 * OLA_E_HIP when the thread failed (message in ola_gpu_last_error). */
int32_t ola_gpu_warmup_wait(double* ms_out);
const char* ola_gpu_last_error(void);
int32_t ola_gpu_sync(OlaCtx* ctx);
/* Scratch and commitment buffers are recycled through a per-context cache (tens of GB after a 2^22-row proof); this returns
 * the cached blocks to the driver.  Live OlaBatch objects are not affected.
 * Single-device contexts of one process that sit on the same GPU hand cached blocks to each other: a context that needs a
 * block takes a fitting one from a sibling that is idle (no call running on it, its stream drained) before it asks the
 * driver -- a host that keeps a Poseidon and a Blake3 context pays for one pool, and the second context's first proof does
 * not wait for the driver to scrub recycled VRAM.  OLA_POOL_SHARE=0 (read at ola_gpu_init) keeps a context out of it. */
int32_t ola_gpu_trim(OlaCtx* ctx);
/* Device memory of the context's buffer pool, in bytes: out[0] handed out now, out[1] the most ever handed out at once,
 * out[2] handed out + cached now, out[3] the most ever held (the high-water mark of a proof; reset = 1 restarts the marks).
 * On a multi-device context every rank has its own pool: each figure is the largest over the ranks (what one GPU must hold). */
int32_t ola_gpu_memory_stats(OlaCtx* ctx, uint64_t out[4], int32_t reset);
/* Where the device time of a proof goes with respect to the coset partition.  enable: 1 / 0 switches the accounting on / off for
 * the proofs that follow (a few hundred event records per proof, no synchronisation), -1 leaves it as it is.  out (may be NULL)
 * describes the LAST ola_prove_with_traces of this context:
 *   out[0]  wall-clock milliseconds of the call
 *   out[1..3]  milliseconds of kernel time in work that the partition divides among the ranks, by the largest world that still
 *           divides it: [1] at most 2 ranks, [2] at most 4, [3] 8 (commitment LDEs, leaf hashing, Merkle sub-trees, quotient
 *           evaluation on 2^qdb cosets, opening evaluations, the first FRI layer).  On a multi-rank run this is rank 0's SHARE.
 *   out[4], out[5]  bytes gathered by the partition's exchanges (all ranks' payload together; a rank receives (G-1)/G of it)
 *           and their number -- counted on a single-GPU run as well, which is what makes a projection from one GPU possible:
 *           T(G) ~ out[0] - sum_k out[k] * (1 - 1/min(G, 2^k)) + out[4] * (G-1)/G / (xGMI rate) + out[5] * latency
 *   out[6], out[7]  multi-device context only: exchanges performed by the library's own all-gather and the bytes they moved between
 *           the ranks, both SINCE THE CONTEXT WAS CREATED (divide by the number of proofs). */
int32_t ola_gpu_proof_stats(OlaCtx* ctx, int32_t enable, double out[8]);
/* With the accounting on, the kernel families of the last proof one by one (Merkle leaves/s,
 * permutations/s, FRI-fold bytes/s of a REAL proof): out[3*i .. 3*i+2] = device milliseconds and two unit counters of phase i.
 *   OLA_PHASE_LEAF_HASH      hash invocations (sponge permutations / Blake3 compressions), bytes read + written
 *   OLA_PHASE_MERKLE_LEVELS  inner nodes hashed (one permutation each under Poseidon), bytes read + written
 *   OLA_PHASE_FRI_FOLD       bytes read + written, extension elements folded (fri/prover.rs:98-112)
 *   OLA_PHASE_LDE            bytes read + written by the coset LDEs (8*n*(1 + cosets) per column), column-cosets
 *   OLA_PHASE_INTT           bytes (16*n per column), columns
 *   OLA_PHASE_QUOTIENT       LDE points evaluated, bytes a point streams (local + next row of the trace and Z batches, 16 written)
 *   OLA_PHASE_OPEN_EVAL      coefficient x point products, bytes read
 * (hash/merkle_tree/mod.rs:180-266, fri/prover.rs:72-121, fri/oracle.rs:66-99).  On a multi-device context: rank 0's share. */
#define OLA_PHASE_LEAF_HASH 0
#define OLA_PHASE_MERKLE_LEVELS 1
#define OLA_PHASE_FRI_FOLD 2
#define OLA_PHASE_LDE 3
#define OLA_PHASE_INTT 4
#define OLA_PHASE_QUOTIENT 5
#define OLA_PHASE_OPEN_EVAL 6
#define OLA_PHASE_COUNT 7
/* Asked for 2 * OLA_PHASE_COUNT rows, rows OLA_PHASE_COUNT + p hold phase p's DOMINANT scope -- the one launch (or group of launches) that
 * moved the most bytes: {device ms, its bytes, 0}.  A phase's totals hide it behind the launch-bound small ones (a proof has about thirty
 * FRI folds, two of them large). */
int32_t ola_gpu_phase_stats(OlaCtx* ctx, double* out /* 3 * n_phases */, uint32_t n_phases);
/* The reference's `timed!` scopes of the LAST whole proof with device times, for the caller's TimingTree (prover.rs:84
 * `timing: &mut TimingTree`; plonky2/plonky2/src/util/timing.rs:7-194; scope names prover.rs:111-553, fri/oracle.rs:56-90,221-225,
 * fri/prover.rs:41-58).  enable: 1 / 0 switches the recording on / off for the proofs that follow (two event records per scope on
 * the context's stream, no synchronisation inside the proof), -1 leaves it.  out (may be NULL; cap entries): the scopes in the
 * order they were entered; *n_out (may be NULL) their number (returns OLA_E_INVALID_ARG when cap is too small; *n_out is set).
 *   name                the reference's scope name ("compute Zs commitment", "IFFT", "perform final FFT 33554432", ...) or one of
 *                       this library's grouping scopes ("table 3 prove_single_table", "prove_with_traces total")
 *   is_reference_scope  1 when the reference has a `timed!` of that name at that place
 *   depth / ref_depth   open scopes around it: all of them / only the reference's (the reference's tree is flat across tables)
 *   table               index of the table being proven, -1 outside
 *   start_ms, ms        when the GPU entered the scope (since the proof's first enqueue) and how long it stayed
 *   sharded_ms          part of `ms` that the coset partition divides among ranks (needs ola_gpu_proof_stats switched on)
 * "transpose LDEs" (fri/oracle.rs:84) never appears: the LDE is produced in leaf order, there is no transpose. */
typedef struct OlaScopeTime {
    char name[64];
    uint32_t depth;
    uint32_t ref_depth;
    int32_t table;
    uint32_t is_reference_scope;
    double start_ms;
    double ms;
    double sharded_ms;
} OlaScopeTime;
int32_t ola_gpu_scope_times(OlaCtx* ctx, int32_t enable, OlaScopeTime* out, uint32_t cap, uint32_t* n_out);
/* The trace upload of the last whole proof: out[0] milliseconds the proving thread was blocked waiting for column groups,
 * out[1] milliseconds from the first byte asked for to the last byte on the device, out[2] until the first column group was
 * complete, out[3] bytes, out[4] path (0 pinned staging ring, 1 pageable hipMemcpyAsync: env OLA_UPLOAD),
 * out[5] copier threads, out[6] bytes that crossed the link (columns whose words are all below 2^32 travel as 32-bit words and are
 * widened on the device: OLA_UPLOAD_PACK=0 switches that off), out[7] reserved.  On a multi-device context: rank 0's share. */
int32_t ola_gpu_upload_stats(OlaCtx* ctx, double out[8]);
/* The transform passes one by one (the roofline of the DOMINANT kernel needs that kernel's own launch duration, not
 * the mean over a transform's passes).  enable: 1 / 0 switches on / off, for the transforms that follow, two event records around
 * every launch of ntt2t_pass_kernel on the context's stream (log_n >= 14; no synchronisation inside a transform), -1 leaves it.
 * The call drains the stream and returns what was recorded since the last call, summed per kernel instantiation
 * (template arguments as rocprofv3 prints them: <R, MODE, INV, CB, LM>), and forgets it.  out may be NULL; *n_out = entries. */
typedef struct OlaPassTime {
    char kernel[48];
    uint32_t launches;
    uint32_t reserved;
    double total_ms;
    double elements;            /* field elements the launches transformed (each read once and written once) */
} OlaPassTime;
int32_t ola_gpu_ntt_pass_times(OlaCtx* ctx, int32_t enable, OlaPassTime* out, uint32_t cap, uint32_t* n_out);
/* Self-test of the device field arithmetic: the kernels' modular reduction is written with explicit carry chains in inline
 * assembly (olavm_amd/csrc/gl.cuh); this compares it with the plain C++ reduction on a table of edge values and on `pairs`
 * pseudo-random operand pairs and returns the number of disagreements (0 expected; about 10^9 pairs per 50 ms).  It also runs
 * the limb arithmetic of the transform passes (olavm_amd/csrc/ntt2t.cuh: fold to u64 with its carry fixes, carry step, product
 * cut into limbs, table multiplication, the sixteen shifts of a radix-16 block) against canonical arithmetic on pairs / 16 limb
 * vectors that sit on the magnitude bounds.  Meant to be run once after ola_gpu_init on a new driver or compiler. */
int32_t ola_gpu_selftest(OlaCtx* ctx, uint64_t pairs, uint64_t* mismatches);
/* Start allocating, on a helper thread, the large device buffers that ola_prove_with_traces will need for this AIR set and these
 * table heights (log2 rows per table); returns at once.  The driver scrubs previously used VRAM inside hipMalloc (about 30 ms per
 * GB here), which is what makes the first proof of a process slow; called right after ola_gpu_init -- before the host reads or
 * generates the traces -- it moves that cost off the proof.  Optional: proving without it is correct, only colder. */
int32_t ola_gpu_reserve(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, const uint32_t* log_n);

/* ---- per-phase entry points: one `timed!` scope of prove_single_table at a time ---------------------------------------
 * The Rust host keeps its transcript and draws the challenges exactly where the reference does; each call replaces the body of
 * one scope (circuits/src/stark/prover.rs:374-480) and hands host arrays back, so a port can move to the GPU one scope at a
 * time and compare with the CPU prover after every step.  ola_prove_single_table is the same sequence in one call. */
/* out[6] = columns, public-parameter words, permutation Z columns, CTL Z columns, quotient_degree_factor,
 * permutation_batch_size of table `table` of the AIR set (for config.num_challenges of the context). */
int32_t ola_table_shape(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, uint32_t out[6]);
/* compute_permutation_z_polys (permutation.rs:103-187).  trace_cols: host columns of 2^log_n rows; perm_challenges:
 * [permutation_batch_size][num_challenges][2] = (beta, gamma) in get_n_grand_product_challenge_sets order (prover.rs:360-367);
 * z_out: [permutation Z columns][n], column-major values (the head of the reference's `z_polys`). */
int32_t ola_perm_z(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, uint32_t log_n,
                   const uint64_t* const* trace_cols, const uint64_t* perm_challenges, uint64_t* z_out);
/* The table's columns of cross_table_lookup_data (cross_table_lookup.rs:224-311), looking sides before looked sides, lookups in
 * declaration order, challenge-minor.  ctl_challenges: [num_challenges][2] = (beta, gamma).  z_out: [CTL Z columns][n].
 * "Non-binary filter?" is reported as OLA_E_INVALID_ARG. */
int32_t ola_ctl_z(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, uint32_t log_n,
                  const uint64_t* const* trace_cols, const uint64_t* ctl_challenges, uint64_t* z_out);
/* compute_quotient_polys and the split into chunks (prover.rs:441-480, 571-705).  trace / zs: the table's trace and Z commitments
 * (ola_commit_values of the trace columns and of perm Z columns followed by CTL Z columns); alphas: [num_challenges];
 * chunks_out: [num_challenges * quotient_degree_factor][n] coefficients, the input of from_coeffs (ola_commit_coeffs).
 * OLA_E_QUOTIENT_DEGREE when the trace does not satisfy the constraints. */
int32_t ola_quotient(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, const OlaBatch* trace,
                     const OlaBatch* zs, const uint64_t* perm_challenges, const uint64_t* ctl_challenges, const uint64_t* alphas,
                     const uint64_t* params, uint64_t* chunks_out);

/* ---- NTT family: replaces gpu_method and the cfft CPU paths --------------------------------------------
 * op selects the reference function (plonky2/field/src/cfft/mod.rs):
 *   OLA_NTT_EVALUATE              evaluate_poly               :22   coeffs -> values on <w>, natural order
 *   OLA_NTT_INTERPOLATE           interpolate_poly            :128  values -> coeffs (x n^-1), natural order
 *   OLA_NTT_COSET_LDE             evaluate_poly_with_offset   :65   n coeffs -> n*2^blowup_log values at
 *                                                                   shift*g^m, m natural (shift = `shift`)
 *   OLA_NTT_COSET_INTERPOLATE     interpolate_poly_with_offset:180  values on shift*<w> -> coeffs
 *   OLA_NTT_COSET_LDE_LEAF_ORDER  the same LDE (shift must be 7) but rows in commitment-leaf order, i.e.
 *                                 bit-reversed (what PolynomialBatch::from_coeffs feeds the Merkle tree,
 *                                 fri/oracle.rs:84-85) -- the order the device keeps internally.
 * `in`/`out` are column-major batches; out may equal in when the op preserves length.
 */
#define OLA_NTT_EVALUATE 0
#define OLA_NTT_INTERPOLATE 1
#define OLA_NTT_COSET_LDE 2
#define OLA_NTT_COSET_INTERPOLATE 3
#define OLA_NTT_COSET_LDE_LEAF_ORDER 4

int32_t ola_ntt_batch(OlaCtx* ctx, int32_t op, const uint64_t* in, uint64_t* out, uint32_t log_n, uint32_t batch,
                      uint64_t shift, uint32_t blowup_log);
/* same, operands resident in HBM; `scratch_dev` must hold batch * 2^log_n elements when log_n > 13 and the op
 * produces natural order (NULL lets the library allocate and free it). */
int32_t ola_ntt_batch_dev(OlaCtx* ctx, int32_t op, const uint64_t* in_dev, uint64_t* out_dev, uint64_t* scratch_dev,
                          uint32_t log_n, uint32_t batch, uint64_t shift, uint32_t blowup_log);

/* ---- Poseidon / Merkle: replaces hash/poseidon.rs:593-603, hashing.rs:84-111, merkle_tree/mod.rs:180-337 --
 * ola_hash_rows, ola_merkle_cap and every commitment hash with the context's Hasher: the Poseidon sponge, or under
 * OLA_HASH_BLAKE3 Blake3_256 (hash/blake3.rs:203-233) of the canonical little-endian words; a digest is 4 words (32 bytes)
 * either way, and a Blake3 digest is bytes -- its words may be >= p and are never reduced. */
/* n states of 12 elements each, permuted in place (host memory). */
int32_t ola_poseidon_permute(OlaCtx* ctx, uint64_t* states, size_t n);
/* hash_no_pad over each row of a row-major num_rows x row_len host matrix -> digests (num_rows x 4). */
int32_t ola_hash_rows(OlaCtx* ctx, const uint64_t* rows, size_t num_rows, size_t row_len, uint64_t* digests);
/* MerkleTree::new_v2 over row-major host leaves: writes the cap (2^cap_height x 4). */
int32_t ola_merkle_cap(OlaCtx* ctx, const uint64_t* leaves, size_t num_leaves, size_t leaf_len, uint32_t cap_height,
                       uint64_t* cap_out);

/* ---- PolynomialBatch commitment: replaces PolynomialBatch::from_values / from_coeffs (fri/oracle.rs:45,66) ---
 * iNTT -> coset LDE (shift 7, blowup 2^rate_bits) -> Poseidon leaf hashes -> Merkle cap, all on the device.
 * cols: ncols host pointers (ola_commit_*) or one device buffer, column-major (ola_commit_*_dev).
 * cap_out: 2^cap_height x 4 canonical elements (host).  *out_batch stays resident until ola_batch_free. */
int32_t ola_commit_values(OlaCtx* ctx, const uint64_t* const* cols, uint32_t ncols, uint32_t log_n,
                          OlaBatch** out_batch, uint64_t* cap_out);
int32_t ola_commit_coeffs(OlaCtx* ctx, const uint64_t* const* cols, uint32_t ncols, uint32_t log_n,
                          OlaBatch** out_batch, uint64_t* cap_out);
int32_t ola_commit_values_dev(OlaCtx* ctx, const uint64_t* cols_dev, uint32_t ncols, uint32_t log_n,
                              OlaBatch** out_batch, uint64_t* cap_out);
int32_t ola_commit_coeffs_dev(OlaCtx* ctx, const uint64_t* cols_dev, uint32_t ncols, uint32_t log_n,
                              OlaBatch** out_batch, uint64_t* cap_out);
/* One GPU's share of PolynomialBatch::from_values under the coset partition: with rate_bits = 3 the LDE is 8
 * independent cosets that are also contiguous blocks of n commitment leaves.  Shard `rank` of `world`
 * (1, 2, 4 or 8) interpolates all columns (replicated), extends them onto cosets [rank*8/world, (rank+1)*8/world), hashes
 * those leaves and builds their sub-trees; cap_slice_out receives entries [rank*16/world, (rank+1)*16/world) of the
 * commitment's Merkle cap (2^cap_height/world digests).  Concatenating the slices of all ranks in rank order -- one
 * all-gather of 512/world bytes -- gives exactly the cap ola_commit_values returns.  The batch answers
 * ola_batch_get_leaf for local leaf indices (global index - rank*N/world) with paths up to its own cap slice. */
int32_t ola_commit_values_shard(OlaCtx* ctx, const uint64_t* const* cols, uint32_t ncols, uint32_t log_n, uint32_t rank,
                                uint32_t world, OlaBatch** out_batch, uint64_t* cap_slice_out);
int32_t ola_commit_values_shard_dev(OlaCtx* ctx, const uint64_t* cols_dev, uint32_t ncols, uint32_t log_n, uint32_t rank,
                                    uint32_t world, OlaBatch** out_batch, uint64_t* cap_slice_out);
int32_t ola_batch_free(OlaCtx* ctx, OlaBatch* batch);
/* accessors (PolynomialBatch.polynomials, MerkleTree::get / prove, merkle_tree/mod.rs:268-308) */
int32_t ola_batch_shape(const OlaBatch* batch, uint32_t* ncols, uint32_t* log_n, uint32_t* rate_bits);
int32_t ola_batch_get_coeffs(OlaCtx* ctx, const OlaBatch* batch, uint64_t* out /* ncols x n, column-major */);
int32_t ola_batch_get_leaf(OlaCtx* ctx, const OlaBatch* batch, size_t leaf_index, uint64_t* row_out /* ncols */,
                           uint64_t* siblings_out /* (log_n+rate_bits-cap_height) x 4 */);
/* fri/oracle.rs:131-137 get_lde_values(index, step): natural-order LDE row index*step */
int32_t ola_batch_get_lde_row(OlaCtx* ctx, const OlaBatch* batch, size_t index, size_t step, uint64_t* row_out);

/* ---- openings + FRI: replaces StarkOpeningSet::new (circuits/src/stark/proof.rs:198-233),
 * PolynomialBatch::prove_openings (fri/oracle.rs:167-241) and fri_proof (fri/prover.rs:20-204).
 * The Fiat-Shamir transcript stays on the host: the caller passes the 12-element challenger state and the
 * buffered inputs, and receives them back (iop/challenger.rs:36-162); see OlaChallenger. */
typedef struct OlaChallenger {
    uint64_t sponge_state[12];
    uint64_t input_buffer[8];
    uint64_t output_buffer[8];
    uint32_t input_len;
    uint32_t output_len;
    uint32_t hasher;            /* OLA_HASH_*: the permutation (Poseidon, or Blake3Permutation, hash/blake3.rs:166-201) and  */
    uint32_t reserved;          /* how a digest is observed; set by the init calls, do not change afterwards                 */
} OlaChallenger;

int32_t ola_challenger_init(OlaChallenger* ch);                          /* Challenger::<F, PoseidonHash>::new()          */
int32_t ola_challenger_init_hasher(OlaChallenger* ch, uint32_t hasher);  /* Challenger::<F, C::Hasher>::new()             */
/* observe_cap (iop/challenger.rs:75-84) of n digests of 4 words: a Poseidon digest is 4 elements, a Blake3 digest is observed
 * as the 5 elements BytesHash::to_vec cuts it into (7 bytes each, hash/hash_types.rs:142-152). */
int32_t ola_challenger_observe_cap(OlaChallenger* ch, const uint64_t* digests, size_t n);
/* Blake3_256::hash_no_pad of n field elements as this backend computes it on the device (canonical little-endian words,
 * hash/blake3.rs:203-213), on the host: for digests the integrator needs outside a proof, and the CPU-side test of the kernel code. */
int32_t ola_blake3_hash_elements(const uint64_t* elems, size_t n, uint64_t out[4]);
int32_t ola_challenger_observe(OlaChallenger* ch, const uint64_t* elems, size_t n);
int32_t ola_challenger_get(OlaChallenger* ch, uint64_t* out, size_t n);
int32_t ola_challenger_compact(OlaChallenger* ch);

/* The tail of prove_single_table from `zeta` on (prover.rs:499-553) for three resident commitments
 * (trace, permutation/CTL Zs, quotient chunks): draws zeta, evaluates the opening set, observes it, runs FRI
 * (commit phase, minimal proof-of-work nonce, 28 query rounds).  Output is the reference wire format
 * (circuits/src/stark/serialization.rs:163-176 write_opening_set || :305-317 write_fri_proof).
 * Returns OLA_E_INVALID_ARG with *out_len = required size when `cap` is too small (challenger is then unchanged). */
int32_t ola_open_and_prove(OlaCtx* ctx, const OlaBatch* trace, const OlaBatch* zs, const OlaBatch* quotient,
                           uint32_t num_permutation_zs, OlaChallenger* challenger, uint8_t* out, size_t cap,
                           size_t* out_len, size_t* openings_len);

/* FRI proof of work (fri/prover.rs:126-148): the MINIMAL witness i such that
 * Poseidon.hash_no_pad([h0..h3, i])[0] has >= bits leading zeros. */
int32_t ola_pow(OlaCtx* ctx, const uint64_t h[4], uint32_t bits, uint64_t* witness);

/* ---- the same, one step per call (ABI revision 7): for a host that keeps the reference's own loops and its own Challenger and
 * hands over the device work only.  Each beta depends on the previous layer's cap through the host challenger
 * (fri/prover.rs:98-101), so the commit phase is layer-stepped.  The bytes the steps return reassemble to ola_open_and_prove's
 * (tests/test_gpu_fri_steps.py does exactly that with ola_challenger_*).  Single-device contexts; the three batches must outlive
 * the OlaFri.
 *   ola_open                   StarkOpeningSet::new (circuits/src/stark/proof.rs:198-233) at the caller's zeta: the opening set in
 *                              wire format (serialization.rs:164-175 write_opening_set; read it back with read_opening_set, :176-193) -- decode, then
 *                              challenger.observe_openings(&openings.to_fri_openings()) (fri/challenges.rs:16-23)
 *   ola_fri_plan               fri_params.reduction_arity_bits (fri/reduction_strategies.rs:40-52) and the final polynomial's length
 *   ola_fri_commit_begin       PolynomialBatch::prove_openings up to the final polynomial (fri/oracle.rs:178-219), alpha =
 *                              challenger.get_extension_challenge()
 *   ola_fri_commit_next_layer  one turn of fri_committed_trees (fri/prover.rs:72-121): folds by the PREVIOUS layer's beta (NULL for the
 *                              first layer), commits the current polynomial's values on its coset, cap_out = 2^cap_height x 4 words
 *   ola_fri_commit_finish      folds by the last beta (NULL when the plan has no layer): the final polynomial, (a, b) pairs
 *                              (prover.rs:114-119); *n_out = its length
 *   ola_pow (above)            fri_proof_of_work (prover.rs:126-148), minimal witness
 *   ola_fri_query              fri_prover_query_rounds (prover.rs:150-204) for the caller's indices (challenger output mod the LDE
 *                              size): the query round proofs in wire format (write_fri_query_rounds, serialization.rs:275-292: count, per query the three
 *                              oracles' rows and paths and every layer's leaf and path) */
typedef struct OlaFri OlaFri;
int32_t ola_open(OlaCtx* ctx, const OlaBatch* trace, const OlaBatch* zs, const OlaBatch* quotient, uint32_t num_permutation_zs,
                 const uint64_t zeta[2], uint8_t* out, size_t cap, size_t* out_len, OlaFri** fri_out);
int32_t ola_fri_plan(const OlaFri* fri, uint32_t* arity_bits, uint32_t cap, uint32_t* n_layers, uint32_t* final_poly_len);
int32_t ola_fri_commit_begin(OlaFri* fri, const uint64_t alpha[2]);
int32_t ola_fri_commit_next_layer(OlaFri* fri, const uint64_t* beta, uint64_t* cap_out);
int32_t ola_fri_commit_finish(OlaFri* fri, const uint64_t* beta, uint64_t* final_poly_out, size_t cap_elems, size_t* n_out);
int32_t ola_fri_query(OlaFri* fri, const uint64_t* x_index, uint32_t n, uint8_t* out, size_t cap, size_t* out_len);
int32_t ola_fri_free(OlaFri* fri);

/* ---- the whole multi-table proof: replaces prove_with_traces (circuits/src/stark/prover.rs:79-327) -------------
 * airset: the AIR-set description (tables, constraint programs, permutation pairs, cross-table lookups) as a u64
 *   array -- the data form of the reference's OlaStark (stark/ola_stark.rs:29-64, 122-560); format and generator in
 *   olavm_amd/air/dsl.py.
 * traces[t]: pointer to table t, column-major ncols x 2^log_n[t] (the [Vec<PolynomialValues<F>>; NUM_TABLES]
 *   that generate_traces returns, generation/mod.rs:77-213): host memory (pageable is fine; the upload overlaps the first
 *   commitments), or memory of this GPU for a table that is already resident -- table by table, the library looks at the
 *   pointer.  A resident table must be complete on the context's stream (or the device idle) when the call is made; the
 *   tables are not modified.
 * params: concatenated per-table constraint parameters (e.g. the bitwise/program compress challenge read inside
 *   the AIR), may be NULL when no table has any; compress_challenges: one per table as carried in AllProof
 *   (prover.rs:307-320), may be NULL (zeros).
 * out: AllProof in the reference wire format (serialization.rs:377-393 write_all_proof).  Returns
 * OLA_E_INVALID_ARG with *out_len = required size when `cap` is too small -- the finished proof then stays in the context
 * and ola_take_pending_proof copies it out without proving again -- and OLA_E_QUOTIENT_DEGREE when the trace does not
 * satisfy the constraints (prover.rs:469-473).  A 12-table proof is 0.8 - 1.2 MB. */
int32_t ola_prove_with_traces(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, const uint64_t* const* traces,
                              const uint32_t* log_n, const uint64_t* params, const uint64_t* compress_challenges,
                              uint8_t* out, size_t cap, size_t* out_len);

/* The same proof from the reference's own trace type: prove_with_traces takes `&[Vec<PolynomialValues<F>>; NUM_TABLES]`
 * (circuits/src/stark/prover.rs:79-83) -- per table a Vec of columns, every column its own Vec<F>
 * (plonky2/field/src/polynomial/mod.rs:24-26), F = GoldilocksField = repr(transparent) u64 (goldilocks_field.rs:24-26).
 * cols[t][c]: pointer to the 2^log_n[t] values of column c of table t, wherever the caller's allocator put it; the Rust shim
 * passes `c.values.as_ptr() as *const u64` and copies nothing (precedent for per-column pointers: cfft/ntt/mod.rs:123-147).
 * ola_prove_with_traces is the special case cols[t][c] = traces[t] + c 2^log_n[t].  Everything else as above; host columns are
 * staged through the context's pinned ring by a few copier threads (OLA_UPLOAD_THREADS, default min(4, host threads per rank - 1); olavm_amd/csrc/upload.h),
 * a table whose FIRST column is device memory is taken as resident on this GPU column by column. */
int32_t ola_prove_with_traces_cols(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, const uint64_t* const* const* cols,
                                   const uint32_t* log_n, const uint64_t* params, const uint64_t* compress_challenges,
                                   uint8_t* out, size_t cap, size_t* out_len);

/* Copies out (and forgets) the proof a preceding ola_prove_with_traces could not return because its buffer was too small. */
int32_t ola_take_pending_proof(OlaCtx* ctx, uint8_t* out, size_t cap, size_t* out_len);

/* One table of that proof with the orchestration left to the caller: replaces prove_single_table
 * (circuits/src/stark/prover.rs:330-513) together with this table's share of cross_table_lookup_data
 * (cross_table_lookup.rs:224-311), compute_permutation_z_polys (permutation.rs:103-155) and compute_quotient_polys
 * (prover.rs:571-705).  The caller has committed the traces (ola_commit_values -> trace_commitment, trace_cap), observed
 * all trace caps and drawn the CTL challenges (get_grand_product_challenge_set, permutation.rs:190-216:
 * ctl_challenges = num_challenges x (beta, gamma)); `challenger` is the shared transcript at the point where the
 * reference calls prove_single_table for table `table` and is advanced exactly as the reference advances it.
 * trace_cols: host pointers to the table's columns (values, 2^log_n each).  out: this table's StarkProof bytes
 * (serialization.rs:349-358 write_proof); an AllProof is u32 count, the proofs in table order, u32 count, the compress
 * challenges.  Same error codes as ola_prove_with_traces; on error the challenger is left untouched. */
int32_t ola_prove_single_table(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table,
                               const uint64_t* const* trace_cols, const OlaBatch* trace_commitment, const uint64_t* trace_cap,
                               const uint64_t* ctl_challenges, const uint64_t* params, OlaChallenger* challenger, uint8_t* out,
                               size_t cap, size_t* out_len);

/* ---- trace generation helper -------------------------------------------------------------------------
 * The 134-column Poseidon STARK table (circuits/src/builtins/poseidon/columns.rs) from the permutation inputs: what the
 * reference's executor records per hash (core/src/util/poseidon_utils.rs) and generate_poseidon_trace lays out
 * (circuits/src/generation/poseidon.rs:5-80).  inputs: 12 x n column-major; filters: 4 x n column-major
 * (FILTER_LOOKED_NORMAL, _TREEKEY, _STORAGE_LEAF, _STORAGE_BRANCH) or NULL for zeros; out: 134 x n column-major.  Padding
 * rows are rows with all-zero inputs (the reference's POSEIDON_ZERO_HASH_* constants). */
int32_t ola_generate_poseidon_trace(OlaCtx* ctx, const uint64_t* inputs, const uint64_t* filters, size_t n, uint64_t* out);
/* The permuted columns of one Halo2-style lookup (circuits/src/stark/lookup.rs:68-132 permuted_cols, called for every
 * looking column of the range-check, bitwise and program tables: generation/builtin.rs:121-200, generation/prog.rs):
 * permuted_inputs = the n inputs in canonical form, ascending; permuted_table = the n table values arranged as the
 * reference's merge loop arranges them (a value next to the first of its inputs, the unused ones in the other slots, in
 * the reference's stack order).  Inputs need not occur in the table.  Host buffers / buffers resident in HBM. */
int32_t ola_permuted_cols(OlaCtx* ctx, const uint64_t* inputs, const uint64_t* table, size_t n, uint64_t* permuted_inputs,
                          uint64_t* permuted_table);
int32_t ola_permuted_cols_dev(OlaCtx* ctx, const uint64_t* inputs_dev, const uint64_t* table_dev, size_t n,
                              uint64_t* permuted_inputs_dev, uint64_t* permuted_table_dev);

/* ---- coset-partitioned proving over several GPUs ---------------------------------------------------------
 * One process per GPU; every process calls ola_prove_with_traces with the SAME traces.  Because the transcript is a
 * function of the (identical) commitments, all ranks draw the same challenges without talking to each other; only the
 * heavy work is divided.  For every table with at least 2^12 rows, rank r uploads 1/world of the columns (the values are
 * all-gathered device to device: xGMI instead of `world` copies over PCIe), extends and hashes cosets
 * [r*8/world, (r+1)*8/world) of all three commitments, and evaluates the quotient on those of its cosets that belong to the
 * quotient domain (the first 2^qdb cosets; all of them for the CPU, memory and Poseidon tables); it evaluates its 1/world of the
 * COLUMNS at the opening points, and extends / hashes its cosets of the first FRI layer.  The exchanges go through `all_gather`,
 * twelve per table -- the trace values (8n bytes per column, once), Merkle cap slices of the three commitments and of the first
 * FRI layer (512 B per tree), the two planes of quotient values (16 * 8n bytes per table), the opening values (a few KB), the
 * opened rows and paths of the 28 queries in the three commitments and in the first FRI layer -- and every rank ends up with the
 * complete, identical AllProof bytes.
 * all_gather(user, send_dev, recv_dev, bytes): gather `bytes` bytes of DEVICE memory from every rank into recv_dev in rank
 * order (world * bytes); return 0 when recv_dev is complete (or, with OLA_SHARD_STREAM_ORDERED, when the collective has been
 * enqueued on the context's stream).  world must be 1, 2, 4 or 8; world = 1 (or a NULL callback) restores single-GPU proving. */
typedef int32_t (*ola_all_gather_fn)(void* user, const void* send_dev, void* recv_dev, size_t bytes);
int32_t ola_set_shard(OlaCtx* ctx, uint32_t rank, uint32_t world, ola_all_gather_fn all_gather, void* user);
/* OLA_SHARD_STREAM_ORDERED: the callback enqueues the collective on the context's stream (ola_gpu_get_stream) -- RCCL launched
 * on that stream -- so the library neither synchronises before the call nor needs the result on return; without the flag the
 * library drains its stream before the call and expects recv_dev complete on return (host-staged collectives). */
#define OLA_SHARD_STREAM_ORDERED 1u
int32_t ola_set_shard_options(OlaCtx* ctx, uint32_t flags);
/* The hipStream_t all of the context's device work is issued on (the one given to ola_gpu_init, or the context's own). */
int32_t ola_gpu_get_stream(OlaCtx* ctx, void** stream_out);

/* Which tables of `airset` have an ahead-of-time specialised constraint-quotient kernel in this build (the counterpart of
 * the reference compiling each table's eval_packed_generic, e.g. cpu/cpu_stark.rs:325): has_kernel[t] = 1 or 0 for every
 * table.  Tables without one are proven with the generic interpreter kernel -- same bytes, fewer points per second.
 * The environment variable OLA_AIR_KERNELS=interpreter disables the specialised kernels, =crosscheck runs both and fails
 * with OLA_E_INTERNAL if they disagree (test hooks). */
int32_t ola_air_kernels_available(const uint64_t* airset, size_t airset_words, uint8_t* has_kernel, size_t ntables);

#ifdef __cplusplus
}
#endif
#endif /* OLA_GPU_H */
