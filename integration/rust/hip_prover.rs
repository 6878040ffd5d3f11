//! `prove_with_traces` on the MI355X backend: the body the `hip` feature substitutes for the CPU prover's
//! (circuits/src/stark/prover.rs:79-327).  Drop into `circuits/src/stark/hip_prover.rs`.
//!
//! One call crosses the boundary: the twelve column-major traces go in as host pointers, the `AllProof` comes back in the
//! reference's own wire format (`Buffer::write_all_proof`, serialization.rs:377-393) and is decoded with `Buffer::read_all_proof`
//! (:394-412).  The executor / client path above (`client/src/main.rs:174-214`, `circuits/benches/fibo_loop.rs:72-91`) does not
//! change.  The hash configuration is the caller's `C`: `PoseidonGoldilocksConfig` or `Blake3GoldilocksConfig`.
use std::any::type_name;
use std::sync::OnceLock;

use anyhow::{ensure, Result};
use plonky2::field::extension::Extendable;
use plonky2::field::polynomial::PolynomialValues;
use plonky2::field::types::PrimeField64;
use plonky2::hash::hash_types::RichField;
use plonky2::plonk::config::GenericConfig;

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
unsafe impl Sync for SendPtr {}

/// One context per hash configuration for the life of the process (the reference's own GPU state is process-wide:
/// cfft/ntt/mod.rs:14-17,48-50).  `OLA_GPUS` = 1, 2, 4 or 8: the context spans that many devices and the partition and its
/// exchanges happen inside the library (`OLA_COLLECTIVE` = peer | rccl selects who moves the bytes).
fn ctx(hasher: u32, fri: &plonky2::fri::FriConfig) -> Result<*mut OlaCtx> {
    static CTX: [OnceLock<SendPtr>; 2] = [OnceLock::new(), OnceLock::new()];
    let slot = &CTX[hasher as usize];
    if let Some(c) = slot.get() {
        return Ok(c.0);
    }
    check_abi()?;
    let n: u32 = std::env::var("OLA_GPUS").ok().and_then(|v| v.parse().ok()).unwrap_or(1);
    let mut cfg: OlaGpuConfig = unsafe { std::mem::zeroed() };
    cfg.device = -1;
    cfg.rate_bits = fri.rate_bits as u32;
    cfg.cap_height = fri.cap_height as u32;
    cfg.proof_of_work_bits = fri.proof_of_work_bits;
    cfg.fri_arity_bits = 4; // FriReductionStrategy::ConstantArityBits(4, 5), config.rs:18-30
    cfg.fri_final_poly_bits = 5;
    cfg.num_query_rounds = fri.num_query_rounds as u32;
    cfg.num_challenges = 2;
    cfg.hasher = hasher;
    let mut c = std::ptr::null_mut();
    check(unsafe { ola_gpu_init_multi(&cfg, std::ptr::null(), n, &mut c) })?;
    Ok(slot.get_or_init(|| SendPtr(c)).0)
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
) -> Result<AllProof<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
{
    ensure!(D == 2, "the backend proves over the quadratic extension");
    // column-major u64 copies of the twelve tables (values may be any representative; the library canonicalises)
    let flat: Vec<Vec<u64>> = trace_poly_values
        .iter()
        .map(|t| t.iter().flat_map(|c| c.values.iter().map(|x| x.to_noncanonical_u64())).collect())
        .collect();
    let ptrs: Vec<*const u64> = flat.iter().map(|t| t.as_ptr()).collect();
    let log_n: Vec<u32> = trace_poly_values.iter().map(|t| t[0].len().trailing_zeros()).collect();
    // prover.rs:307-320: the two compress challenges come from trace generation, not from this transcript
    let bitwise = ola_stark.bitwise_stark.get_compress_challenge().unwrap().to_canonical_u64();
    let program = ola_stark.program_stark.get_compress_challenge().unwrap().to_canonical_u64();
    let params = [bitwise, program];
    let mut compress = [0u64; NUM_TABLES];
    compress[Table::Bitwise as usize] = bitwise;
    compress[Table::Program as usize] = program;

    let c = ctx(hasher_of::<F, C, D>()?, &config.fri_config)?;
    let words = airset();
    let mut out = vec![0u8; 8 << 20];
    let mut len = 0usize;
    let rc = unsafe {
        ola_prove_with_traces(c, words.as_ptr(), words.len(), ptrs.as_ptr(), log_n.as_ptr(), params.as_ptr(), compress.as_ptr(),
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
    let mut proof: AllProof<F, C, D> = Buffer::new(out).read_all_proof()?;
    proof.public_values = public_values; // not part of the wire format (serialization.rs:391)
    Ok(proof)
}
