//! `prove_with_traces` on the MI355X backend: the body the `hip` feature substitutes for the CPU prover's
//! (circuits/src/stark/prover.rs:79-327).  Drop into `circuits/src/stark/hip_prover.rs`.
//!
//! One call crosses the boundary.  The twelve traces go in as they are -- `[Vec<PolynomialValues<F>>; NUM_TABLES]`, every column
//! its own `Vec<F>` (plonky2/field/src/polynomial/mod.rs:24-26), `F = GoldilocksField` = `repr(transparent) u64`
//! (goldilocks_field.rs:24-26): the shim hands the library one pointer per column (`ola_prove_with_traces_cols`) and copies
//! nothing.  The `AllProof` comes back in the reference's own wire format (`Buffer::write_all_proof`, serialization.rs:377-393)
//! and is decoded with `Buffer::read_all_proof` (:394-412).  The `timed!` scopes of the CPU prover come back with the GPU's
//! times and are replayed into the caller's `TimingTree`.  The executor / client path above (`client/src/main.rs:174-214`,
//! `circuits/benches/fibo_loop.rs:72-91`) does not change.  The hash configuration is the caller's `C`:
//! `PoseidonGoldilocksConfig` or `Blake3GoldilocksConfig`.
use std::any::type_name;
use std::ffi::CStr;
use std::sync::{Mutex, OnceLock};
use std::time::{Duration, Instant};

use anyhow::{bail, ensure, Result};
use plonky2::field::extension::Extendable;
use plonky2::field::polynomial::PolynomialValues;
use plonky2::field::types::PrimeField64;
use plonky2::fri::reduction_strategies::FriReductionStrategy;
use plonky2::hash::hash_types::RichField;
use plonky2::plonk::config::GenericConfig;
use plonky2::util::timing::TimingTree;

use super::config::StarkConfig;
use super::ola_gpu_sys::*;
use super::ola_stark::{OlaStark, Table, NUM_TABLES};
use super::proof::{AllProof, PublicValues};
use super::serialization::Buffer;

/// The data form of `OlaStark`'s twelve `Stark` impls and `all_cross_table_lookups()` (stark/ola_stark.rs:29-64,122-560):
/// include/ola_airset.bin, staged by build.rs; little-endian u64 words.
static OLA_AIRSET_BYTES: &[u8] = include_bytes!(concat!(env!("OUT_DIR"), "/ola_airset.bin"));

fn airset() -> &'static [u64] {
    static WORDS: OnceLock<Vec<u64>> = OnceLock::new();
    WORDS.get_or_init(|| OLA_AIRSET_BYTES.chunks_exact(8).map(|c| u64::from_le_bytes(c.try_into().unwrap())).collect())
}

struct SendPtr(*mut OlaCtx);
unsafe impl Send for SendPtr {}

/// Start-up where the reference has it: `OlaStark::default()` calls `plonky2::field::cfft::ntt::init_gpu()`
/// (circuits/src/stark/ola_stark.rs:47, plonky2/field/src/cfft/ntt/mod.rs:53-99) before `prove()` generates the traces
/// (client/src/main.rs:191-200); the patch adds this call on the next line.  Returns at once: a helper thread inside the library
/// starts the HIP runtime, opens the device, loads the code objects, pins the upload ring and primes a context with a throw-away
/// proof of a small all-zero instance while the host generates traces; the first `with_ctx` waits for it and takes that context
/// over.  The hasher and the `StarkConfig` are not
/// known here (`OlaStark` is generic over `F, D` only) and are not needed.  Errors are left to the `ola_gpu_init` that follows.
pub fn init_early() {
    static ONCE: std::sync::Once = std::sync::Once::new();
    ONCE.call_once(|| {
        if check_abi().is_ok() {
            let words = airset();
            let _ = unsafe { ola_gpu_warmup(-1, OLA_WARMUP_PINNED_RING, words.as_ptr(), words.len()) };
        }
    });
}

/// What a context was created with: a later call with another `StarkConfig` gets a new context, not the stale one.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
struct CtxKey {
    rate_bits: u32,
    cap_height: u32,
    proof_of_work_bits: u32,
    arity_bits: u32,
    final_poly_bits: u32,
    num_query_rounds: u32,
    num_challenges: u32,
    gpus: u32,
}

/// One context per hash configuration for the life of the process, and ONE proof at a time on the GPU: a context is
/// thread-compatible, not re-entrant (the reference's own GPU state is process-wide behind a mutex as well:
/// cfft/ntt/mod.rs:14-17,48-50).  The slot is filled under its lock, so two threads that arrive together create one context,
/// not two (a second context would hold tens of GB of pooled HBM for nothing).  `OLA_GPUS` = 1, 2, 4 or 8: the context spans
/// that many devices and the partition and its exchanges happen inside the library (`OLA_COLLECTIVE` = peer | rccl selects who
/// moves the bytes).  The guard is held for the whole proof.
fn with_ctx<R>(hasher: u32, config: &StarkConfig, body: impl FnOnce(*mut OlaCtx) -> Result<R>) -> Result<R> {
    static CTX: [OnceLock<Mutex<Option<(SendPtr, CtxKey)>>>; 2] = [OnceLock::new(), OnceLock::new()];
    let mut slot = CTX[hasher as usize].get_or_init(|| Mutex::new(None)).lock().unwrap_or_else(|e| e.into_inner());
    let fri = &config.fri_config;
    // the library folds with one arity; StarkConfig::standard_fast_config is ConstantArityBits(4, 5) (config.rs:18-30)
    let (arity_bits, final_poly_bits) = match fri.reduction_strategy {
        FriReductionStrategy::ConstantArityBits(a, f) => (a as u32, f as u32),
        ref other => bail!("the hip backend folds with FriReductionStrategy::ConstantArityBits only, the config asks for {other:?}"),
    };
    let key = CtxKey {
        rate_bits: fri.rate_bits as u32,
        cap_height: fri.cap_height as u32,
        proof_of_work_bits: fri.proof_of_work_bits,
        arity_bits,
        final_poly_bits,
        num_query_rounds: fri.num_query_rounds as u32,
        num_challenges: config.num_challenges as u32,
        gpus: std::env::var("OLA_GPUS").ok().and_then(|v| v.parse().ok()).unwrap_or(1),
    };
    if let Some((old, old_key)) = slot.as_ref() {
        if *old_key != key {
            // proving under the first call's FRI parameters would produce a proof the caller's verifier rejects
            check(unsafe { ola_gpu_free(old.0) })?;
            *slot = None;
        }
    }
    if slot.is_none() {
        init_early(); // a caller that never built an OlaStark::default(): the same start-up, then waited for below
        check_abi()?;
        let mut cfg: OlaGpuConfig = unsafe { std::mem::zeroed() };
        cfg.device = -1;
        cfg.rate_bits = key.rate_bits;
        cfg.cap_height = key.cap_height;
        cfg.proof_of_work_bits = key.proof_of_work_bits;
        cfg.fri_arity_bits = key.arity_bits;
        cfg.fri_final_poly_bits = key.final_poly_bits;
        cfg.num_query_rounds = key.num_query_rounds;
        cfg.num_challenges = key.num_challenges;
        cfg.hasher = hasher;
        let mut c = std::ptr::null_mut();
        check(unsafe { ola_gpu_init_multi(&cfg, std::ptr::null(), key.gpus, &mut c) })?;
        *slot = Some((SendPtr(c), key));
    }
    body(slot.as_ref().unwrap().0 .0)
}

/// The `timed!` scopes of the proof that has just run, with the GPU's times (`ola_gpu_scope_times`), replayed into the
/// caller's tree under the names the CPU prover uses -- "compute trace commitments", "compute Zs commitment", "IFFT",
/// "FFT + blinding", "build Merkle tree", "compute quotient polys", "compute openings proof", "perform final FFT {n}",
/// "fold codewords in the commitment phase", "find proof-of-work witness" (prover.rs:111-553, fri/oracle.rs:56-90,221-225,
/// fri/prover.rs:41-58) -- so that `timing.print()` reads as it does for the CPU prover, one line per scope.  Only the
/// reference's scopes are replayed (the library's own grouping scopes, "table 3 prove_single_table", are skipped), at the depth
/// they have among themselves; `t0` is the host's clock when the call was made.  `TimingTree::record` is the one method the
/// patch adds to plonky2/plonky2/src/util/timing.rs (the tree's fields are private).
fn replay_scopes(c: *mut OlaCtx, timing: &mut TimingTree, t0: Instant) -> Result<()> {
    let mut n = 0u32;
    check(unsafe { ola_gpu_scope_times(c, -1, std::ptr::null_mut(), 0, &mut n) })?;
    if n == 0 {
        return Ok(());
    }
    let mut scopes: Vec<OlaScopeTime> = Vec::with_capacity(n as usize);
    check(unsafe { ola_gpu_scope_times(c, -1, scopes.as_mut_ptr(), n, &mut n) })?;
    unsafe { scopes.set_len(n as usize) };
    for s in scopes.iter().filter(|s| s.is_reference_scope != 0) {
        let name = unsafe { CStr::from_ptr(s.name.as_ptr()) }.to_string_lossy();
        timing.record(
            s.ref_depth as usize,
            &name,
            log::Level::Debug,
            t0 + Duration::from_secs_f64(s.start_ms * 1e-3),
            Duration::from_secs_f64(s.ms * 1e-3),
        );
    }
    Ok(())
}

/// `C::Hasher` -> OLA_HASH_*: the two configurations the reference instantiates the prover with
/// (client/src/main.rs:21,31: PoseidonGoldilocksConfig; circuits/benches/fibo_loop.rs:26: Blake3GoldilocksConfig).
fn hasher_of<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>() -> Result<u32> {
    let name = type_name::<C::Hasher>();
    if name.contains("PoseidonHash") {
        Ok(OLA_HASH_POSEIDON)
    } else if name.contains("Blake3") {
        Ok(OLA_HASH_BLAKE3)
    } else {
        Err(anyhow::anyhow!("the hip backend has no Merkle hasher for {name}"))
    }
}

pub fn prove_with_traces_hip<F, C, const D: usize>(
    ola_stark: &OlaStark<F, D>,
    config: &StarkConfig,
    trace_poly_values: &[Vec<PolynomialValues<F>>; NUM_TABLES],
    public_values: PublicValues,
    timing: &mut TimingTree,
) -> Result<AllProof<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
{
    ensure!(D == 2, "the backend proves over the quadratic extension");
    ensure!(std::mem::size_of::<F>() == 8, "the backend reads columns as u64 words (GoldilocksField is repr(transparent) u64)");
    // one pointer per column, no copy: values may be any representative, the library canonicalises on the device
    let cols: Vec<Vec<*const u64>> = trace_poly_values.iter().map(|t| t.iter().map(|c| c.values.as_ptr() as *const u64).collect()).collect();
    let tables: Vec<*const *const u64> = cols.iter().map(|t| t.as_ptr()).collect();
    for t in trace_poly_values.iter() {
        ensure!(!t.is_empty() && t.iter().all(|c| c.len() == t[0].len() && c.len().is_power_of_two()), "ragged trace table");
    }
    let log_n: Vec<u32> = trace_poly_values.iter().map(|t| t[0].len().trailing_zeros()).collect();
    // prover.rs:307-320: the two compress challenges come from trace generation, not from this transcript
    let bitwise = ola_stark.bitwise_stark.get_compress_challenge().unwrap().to_canonical_u64();
    let program = ola_stark.program_stark.get_compress_challenge().unwrap().to_canonical_u64();
    let params = [bitwise, program];
    let mut compress = [0u64; NUM_TABLES];
    compress[Table::Bitwise as usize] = bitwise;
    compress[Table::Program as usize] = program;

    let words = airset();
    let out = with_ctx(hasher_of::<F, C, D>()?, config, |c| {
        // the library reads cols[t][0 .. width of table t in the AIR set): a shorter table would be read past its pointer Vec
        for (t, table) in trace_poly_values.iter().enumerate() {
            let mut shape = [0u32; 6];
            check(unsafe { ola_table_shape(c, words.as_ptr(), words.len(), t as u32, shape.as_mut_ptr()) })?;
            ensure!(table.len() == shape[0] as usize, "table {t} has {} columns, its AIR has {}", table.len(), shape[0]);
        }
        // scope times cost two event records per scope; ask for them when somebody will read the tree (TimingTree logs at Debug)
        let want_scopes = log::log_enabled!(log::Level::Debug);
        check(unsafe { ola_gpu_scope_times(c, want_scopes as i32, std::ptr::null_mut(), 0, std::ptr::null_mut()) })?;
        let mut out = vec![0u8; 8 << 20];
        let mut len = 0usize;
        let t0 = Instant::now();
        let rc = unsafe {
            ola_prove_with_traces_cols(c, words.as_ptr(), words.len(), tables.as_ptr(), log_n.as_ptr(), params.as_ptr(), compress.as_ptr(),
                                       out.as_mut_ptr(), out.len(), &mut len)
        };
        if rc == OLA_E_INVALID_ARG && len > out.len() {
            // the proof is larger than the buffer: it was kept, fetch it without proving again
            out.resize(len, 0);
            check(unsafe { ola_take_pending_proof(c, out.as_mut_ptr(), out.len(), &mut len) })?;
        } else {
            check(rc)?;
        }
        out.truncate(len);
        if want_scopes {
            replay_scopes(c, timing, t0)?;
        }
        Ok(out)
    })?;
    let mut proof: AllProof<F, C, D> = Buffer::new(out).read_all_proof()?;
    proof.public_values = public_values; // not part of the wire format (serialization.rs:391)
    Ok(proof)
}
