//! Raw bindings of `libola_gpu.so` (`include/ola_gpu.h`, ABI revision 6) for the reference's `circuits` crate.
//!
//! Drop into `circuits/src/stark/ola_gpu_sys.rs` (integration/patches/0001-feature-hip.patch adds the `mod` line and the
//! feature).  Replaces the reference's dead CUDA FFI -- `gpu_init` / `gpu_method` / `gpu_free`,
//! `plonky2/field/src/cfft/ntt/mod.rs:21-45` -- with the wider boundary: one call per proof, per table or per `timed!` scope.
//! `GoldilocksField` is `#[repr(transparent)] u64` (`plonky2/field/src/goldilocks_field.rs:24-26`) and `HashOut` is `[F; 4]`
//! (`hash/hash_types.rs:19-21`), so `Vec<F>` / `MerkleCap` cross as plain `*const u64`.
//!
//! This file is checked mechanically against the header (tests/test_rust_shim.py: every export, its arity and the width /
//! constness of every argument); no Rust toolchain exists in the build image, so it has not been compiled there.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_void};

pub const OLA_GPU_ABI_VERSION: i32 = 7;
pub const OLA_OK: i32 = 0;
pub const OLA_E_INVALID_ARG: i32 = -1;
pub const OLA_E_NO_DEVICE: i32 = -2;
pub const OLA_E_OOM: i32 = -3;
/// prover.rs:469-473 "vanishing polynomial is not divisible by Z_H"
pub const OLA_E_QUOTIENT_DEGREE: i32 = -4;
pub const OLA_E_HIP: i32 = -5;
/// prover.rs:508-511
pub const OLA_E_ZETA_IN_SUBGROUP: i32 = -6;
pub const OLA_E_INTERNAL: i32 = -7;
/// PoseidonGoldilocksConfig (plonk/config.rs:112-121)
pub const OLA_HASH_POSEIDON: u32 = 0;
/// Blake3GoldilocksConfig (plonk/config.rs:153-161)
pub const OLA_HASH_BLAKE3: u32 = 1;
pub const OLA_NTT_EVALUATE: i32 = 0;
pub const OLA_NTT_INTERPOLATE: i32 = 1;
pub const OLA_NTT_COSET_LDE: i32 = 2;
pub const OLA_NTT_COSET_INTERPOLATE: i32 = 3;
pub const OLA_NTT_COSET_LDE_LEAF_ORDER: i32 = 4;
pub const OLA_SHARD_STREAM_ORDERED: u32 = 1;
/// ola_gpu_collective: who carries a multi-device context's exchanges
pub const OLA_COLLECTIVE_NONE: u32 = 0;
pub const OLA_COLLECTIVE_PEER: u32 = 1;
pub const OLA_COLLECTIVE_RCCL: u32 = 2;
pub const OLA_PHASE_COUNT: u32 = 7;
/// ola_gpu_warmup: also pin the trace upload's staging ring
pub const OLA_WARMUP_PINNED_RING: u32 = 1;

#[repr(C)]
pub struct OlaCtx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct OlaBatch {
    _p: [u8; 0],
}
/// one table's opening proof between the steps of `ola_open` / `ola_fri_*`
#[repr(C)]
pub struct OlaFri {
    _p: [u8; 0],
}
/// `StarkConfig::standard_fast_config()` (circuits/src/stark/config.rs:18-30) + device selection + the hash configuration.
/// Zero-initialise, then fill: `hasher` sits where the struct's tail padding used to be.
#[repr(C)]
pub struct OlaGpuConfig {
    pub device: i32,
    pub stream: *mut c_void,
    pub rate_bits: u32,
    pub cap_height: u32,
    pub proof_of_work_bits: u32,
    pub fri_arity_bits: u32,
    pub fri_final_poly_bits: u32,
    pub num_query_rounds: u32,
    pub num_challenges: u32,
    pub hasher: u32,
}
/// The fields of `plonky2::iop::challenger::Challenger` (iop/challenger.rs:19-24), lengths <= SPONGE_RATE = 8.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct OlaChallenger {
    pub sponge_state: [u64; 12],
    pub input_buffer: [u64; 8],
    pub output_buffer: [u64; 8],
    pub input_len: u32,
    pub output_len: u32,
    pub hasher: u32,
    pub reserved: u32,
}
/// One `timed!` scope of the last proof with device times (`ola_gpu_scope_times`), for the caller's `TimingTree`
/// (plonky2/plonky2/src/util/timing.rs:7-194).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct OlaScopeTime {
    pub name: [c_char; 64],
    pub depth: u32,
    pub ref_depth: u32,
    pub table: i32,
    pub is_reference_scope: u32,
    pub start_ms: f64,
    pub ms: f64,
    pub sharded_ms: f64,
}
/// The launches of one transform-pass kernel instantiation, summed (`ola_gpu_ntt_pass_times`).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct OlaPassTime {
    pub kernel: [c_char; 48],
    pub launches: u32,
    pub reserved: u32,
    pub total_ms: f64,
    pub elements: f64,
}
/// `ola_all_gather_fn`: gather `bytes` bytes of device memory from every rank into `recv_dev` (rank order), 0 = done
pub type OlaAllGatherFn = Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes: usize) -> i32>;

#[link(name = "ola_gpu")]
extern "C" {
    pub fn ola_gpu_init(cfg: *const OlaGpuConfig, out_ctx: *mut *mut OlaCtx) -> i32;
    pub fn ola_gpu_abi_version(challenger_size: *mut usize, config_size: *mut usize) -> i32;
    pub fn ola_gpu_init_multi(cfg: *const OlaGpuConfig, devices: *const i32, n_devices: u32, out_ctx: *mut *mut OlaCtx) -> i32;
    pub fn ola_gpu_device_count(ctx: *mut OlaCtx, n_devices: *mut u32) -> i32;
    pub fn ola_gpu_collective(ctx: *mut OlaCtx, carrier: *mut u32, ranks: *mut u32, note: *mut c_char, note_cap: usize) -> i32;
    pub fn ola_gpu_all_gather_check(ctx: *mut OlaCtx, carrier: u32, bytes_per_rank: usize, reps: u32, ms_per_gather: *mut f64,
        mismatches: *mut u64) -> i32;
    pub fn ola_gpu_free(ctx: *mut OlaCtx) -> i32;
    pub fn ola_gpu_warmup(device: i32, flags: u32, airset: *const u64, airset_words: usize) -> i32;
    pub fn ola_gpu_warmup_wait(ms_out: *mut f64) -> i32;
    pub fn ola_gpu_last_error() -> *const c_char;
    pub fn ola_gpu_sync(ctx: *mut OlaCtx) -> i32;
    pub fn ola_gpu_trim(ctx: *mut OlaCtx) -> i32;
    pub fn ola_gpu_memory_stats(ctx: *mut OlaCtx, out: *mut u64, reset: i32) -> i32;
    pub fn ola_gpu_proof_stats(ctx: *mut OlaCtx, enable: i32, out: *mut f64) -> i32;
    pub fn ola_gpu_phase_stats(ctx: *mut OlaCtx, out: *mut f64, n_phases: u32) -> i32;
    pub fn ola_gpu_scope_times(ctx: *mut OlaCtx, enable: i32, out: *mut OlaScopeTime, cap: u32, n_out: *mut u32) -> i32;
    pub fn ola_gpu_upload_stats(ctx: *mut OlaCtx, out: *mut f64) -> i32;
    pub fn ola_gpu_ntt_pass_times(ctx: *mut OlaCtx, enable: i32, out: *mut OlaPassTime, cap: u32, n_out: *mut u32) -> i32;
    pub fn ola_gpu_selftest(ctx: *mut OlaCtx, pairs: u64, mismatches: *mut u64) -> i32;
    pub fn ola_gpu_reserve(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, log_n: *const u32) -> i32;
    pub fn ola_table_shape(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, table: u32, out: *mut u32) -> i32;
    pub fn ola_perm_z(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, table: u32, log_n: u32,
        trace_cols: *const *const u64, perm_challenges: *const u64, z_out: *mut u64) -> i32;
    pub fn ola_ctl_z(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, table: u32, log_n: u32,
        trace_cols: *const *const u64, ctl_challenges: *const u64, z_out: *mut u64) -> i32;
    pub fn ola_quotient(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, table: u32, trace: *const OlaBatch,
        zs: *const OlaBatch, perm_challenges: *const u64, ctl_challenges: *const u64, alphas: *const u64,
        params: *const u64, chunks_out: *mut u64) -> i32;
    pub fn ola_ntt_batch(ctx: *mut OlaCtx, op: i32, input: *const u64, out: *mut u64, log_n: u32, batch: u32, shift: u64,
        blowup_log: u32) -> i32;
    pub fn ola_ntt_batch_dev(ctx: *mut OlaCtx, op: i32, in_dev: *const u64, out_dev: *mut u64, scratch_dev: *mut u64,
        log_n: u32, batch: u32, shift: u64, blowup_log: u32) -> i32;
    pub fn ola_poseidon_permute(ctx: *mut OlaCtx, states: *mut u64, n: usize) -> i32;
    pub fn ola_hash_rows(ctx: *mut OlaCtx, rows: *const u64, num_rows: usize, row_len: usize, digests: *mut u64) -> i32;
    pub fn ola_merkle_cap(ctx: *mut OlaCtx, leaves: *const u64, num_leaves: usize, leaf_len: usize, cap_height: u32,
        cap_out: *mut u64) -> i32;
    pub fn ola_commit_values(ctx: *mut OlaCtx, cols: *const *const u64, ncols: u32, log_n: u32,
        out_batch: *mut *mut OlaBatch, cap_out: *mut u64) -> i32;
    pub fn ola_commit_coeffs(ctx: *mut OlaCtx, cols: *const *const u64, ncols: u32, log_n: u32,
        out_batch: *mut *mut OlaBatch, cap_out: *mut u64) -> i32;
    pub fn ola_commit_values_dev(ctx: *mut OlaCtx, cols_dev: *const u64, ncols: u32, log_n: u32,
        out_batch: *mut *mut OlaBatch, cap_out: *mut u64) -> i32;
    pub fn ola_commit_coeffs_dev(ctx: *mut OlaCtx, cols_dev: *const u64, ncols: u32, log_n: u32,
        out_batch: *mut *mut OlaBatch, cap_out: *mut u64) -> i32;
    pub fn ola_commit_values_shard(ctx: *mut OlaCtx, cols: *const *const u64, ncols: u32, log_n: u32, rank: u32,
        world: u32, out_batch: *mut *mut OlaBatch, cap_slice_out: *mut u64) -> i32;
    pub fn ola_commit_values_shard_dev(ctx: *mut OlaCtx, cols_dev: *const u64, ncols: u32, log_n: u32, rank: u32,
        world: u32, out_batch: *mut *mut OlaBatch, cap_slice_out: *mut u64) -> i32;
    pub fn ola_batch_free(ctx: *mut OlaCtx, batch: *mut OlaBatch) -> i32;
    pub fn ola_batch_shape(batch: *const OlaBatch, ncols: *mut u32, log_n: *mut u32, rate_bits: *mut u32) -> i32;
    pub fn ola_batch_get_coeffs(ctx: *mut OlaCtx, batch: *const OlaBatch, out: *mut u64) -> i32;
    pub fn ola_batch_get_leaf(ctx: *mut OlaCtx, batch: *const OlaBatch, leaf_index: usize, row_out: *mut u64,
        siblings_out: *mut u64) -> i32;
    pub fn ola_batch_get_lde_row(ctx: *mut OlaCtx, batch: *const OlaBatch, index: usize, step: usize, row_out: *mut u64) -> i32;
    pub fn ola_challenger_init(ch: *mut OlaChallenger) -> i32;
    pub fn ola_challenger_init_hasher(ch: *mut OlaChallenger, hasher: u32) -> i32;
    pub fn ola_challenger_observe_cap(ch: *mut OlaChallenger, digests: *const u64, n: usize) -> i32;
    pub fn ola_blake3_hash_elements(elems: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn ola_challenger_observe(ch: *mut OlaChallenger, elems: *const u64, n: usize) -> i32;
    pub fn ola_challenger_get(ch: *mut OlaChallenger, out: *mut u64, n: usize) -> i32;
    pub fn ola_challenger_compact(ch: *mut OlaChallenger) -> i32;
    pub fn ola_open_and_prove(ctx: *mut OlaCtx, trace: *const OlaBatch, zs: *const OlaBatch, quotient: *const OlaBatch,
        num_permutation_zs: u32, challenger: *mut OlaChallenger, out: *mut u8, cap: usize, out_len: *mut usize,
        openings_len: *mut usize) -> i32;
    pub fn ola_pow(ctx: *mut OlaCtx, h: *const u64, bits: u32, witness: *mut u64) -> i32;
    pub fn ola_open(ctx: *mut OlaCtx, trace: *const OlaBatch, zs: *const OlaBatch, quotient: *const OlaBatch, num_permutation_zs: u32,
        zeta: *const u64, out: *mut u8, cap: usize, out_len: *mut usize, fri_out: *mut *mut OlaFri) -> i32;
    pub fn ola_fri_plan(fri: *const OlaFri, arity_bits: *mut u32, cap: u32, n_layers: *mut u32, final_poly_len: *mut u32) -> i32;
    pub fn ola_fri_commit_begin(fri: *mut OlaFri, alpha: *const u64) -> i32;
    pub fn ola_fri_commit_next_layer(fri: *mut OlaFri, beta: *const u64, cap_out: *mut u64) -> i32;
    pub fn ola_fri_commit_finish(fri: *mut OlaFri, beta: *const u64, final_poly_out: *mut u64, cap_elems: usize, n_out: *mut usize) -> i32;
    pub fn ola_fri_query(fri: *mut OlaFri, x_index: *const u64, n: u32, out: *mut u8, cap: usize, out_len: *mut usize) -> i32;
    pub fn ola_fri_free(fri: *mut OlaFri) -> i32;
    pub fn ola_prove_with_traces(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, traces: *const *const u64,
        log_n: *const u32, params: *const u64, compress_challenges: *const u64, out: *mut u8, cap: usize,
        out_len: *mut usize) -> i32;
    pub fn ola_prove_with_traces_cols(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, cols: *const *const *const u64,
        log_n: *const u32, params: *const u64, compress_challenges: *const u64, out: *mut u8, cap: usize,
        out_len: *mut usize) -> i32;
    pub fn ola_take_pending_proof(ctx: *mut OlaCtx, out: *mut u8, cap: usize, out_len: *mut usize) -> i32;
    pub fn ola_prove_single_table(ctx: *mut OlaCtx, airset: *const u64, airset_words: usize, table: u32,
        trace_cols: *const *const u64, trace_commitment: *const OlaBatch, trace_cap: *const u64,
        ctl_challenges: *const u64, params: *const u64, challenger: *mut OlaChallenger, out: *mut u8, cap: usize,
        out_len: *mut usize) -> i32;
    pub fn ola_generate_poseidon_trace(ctx: *mut OlaCtx, inputs: *const u64, filters: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn ola_permuted_cols(ctx: *mut OlaCtx, inputs: *const u64, table: *const u64, n: usize, permuted_inputs: *mut u64,
        permuted_table: *mut u64) -> i32;
    pub fn ola_permuted_cols_dev(ctx: *mut OlaCtx, inputs_dev: *const u64, table_dev: *const u64, n: usize,
        permuted_inputs_dev: *mut u64, permuted_table_dev: *mut u64) -> i32;
    pub fn ola_set_shard(ctx: *mut OlaCtx, rank: u32, world: u32, all_gather: OlaAllGatherFn, user: *mut c_void) -> i32;
    pub fn ola_set_shard_options(ctx: *mut OlaCtx, flags: u32) -> i32;
    pub fn ola_gpu_get_stream(ctx: *mut OlaCtx, stream_out: *mut *mut c_void) -> i32;
    pub fn ola_air_kernels_available(airset: *const u64, airset_words: usize, has_kernel: *mut u8, ntables: usize) -> i32;
}

/// Non-zero status -> `anyhow::Error` carrying the library's message (the reference's `Result` / `ensure!` sites).
pub fn check(rc: i32) -> anyhow::Result<()> {
    if rc == OLA_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(ola_gpu_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow::anyhow!("ola_gpu error {rc}: {msg}"))
}

/// Refuse a library whose ABI revision or struct sizes differ from this file's (call once at start-up).
pub fn check_abi() -> anyhow::Result<()> {
    let (mut chal, mut cfg) = (0usize, 0usize);
    let rev = unsafe { ola_gpu_abi_version(&mut chal, &mut cfg) };
    anyhow::ensure!(
        rev == OLA_GPU_ABI_VERSION && chal == std::mem::size_of::<OlaChallenger>() && cfg == std::mem::size_of::<OlaGpuConfig>(),
        "libola_gpu.so is ABI revision {rev} (OlaChallenger {chal} bytes, OlaGpuConfig {cfg} bytes); this binding is revision {OLA_GPU_ABI_VERSION}"
    );
    Ok(())
}
