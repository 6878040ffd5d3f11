//! Pin-readiness harness for the MI355X backend (olavm_amd): NOT compiled in that repository (no Rust toolchain there).
//! Drop into the reference as circuits/src/stark/pin_dump.rs (`#[cfg(test)] mod pin_dump;` in circuits/src/stark/mod.rs) and run
//!     OLA_PIN_DIR=/tmp/ola_pin cargo test --release -p circuits pin_dump -- --nocapture
//! For each program below it writes <name>.traces and <name>.proof; integration/pin/compare_with_dump.py compares the GPU prover's
//! bytes with them.  The body follows ola_stark.rs::test_by_asm_json (execute -> generate_traces -> prove_with_traces -> verify).
#![allow(unused)]
use std::collections::HashMap;
use std::fs::File;
use std::io::Write;
use std::path::PathBuf;

use assembler::encoder::encode_asm_from_json_file;
use core::crypto::ZkHasher;
use core::merkle_tree::tree::AccountTree;
use core::program::Program;
use core::types::account::Address;
use core::types::merkle_tree::{encode_addr, tree_key_default};
use core::types::{Field, GoldilocksField};
use core::vm::transaction::init_tx_context_mock;
use executor::load_tx::init_tape;
use executor::trace::{gen_storage_hash_table, gen_storage_table};
use executor::{Process, TxScopeCacheManager};
use core::types::storage::{StorageLog, WitnessStorageLog};
use plonky2::field::polynomial::PolynomialValues;
use plonky2::field::types::PrimeField64;
use plonky2::plonk::config::{Blake3GoldilocksConfig, GenericConfig, PoseidonGoldilocksConfig};
use plonky2::util::timing::TimingTree;

use crate::generation::{generate_traces, GenerationInputs};
use crate::stark::config::StarkConfig;
use crate::stark::ola_stark::{OlaStark, NUM_TABLES};
use crate::stark::prover::prove_with_traces;
use crate::stark::serialization::Buffer;
use crate::stark::verifier::verify_proof;

const D: usize = 2;
type C = PoseidonGoldilocksConfig;
type CB = Blake3GoldilocksConfig; // the configuration of the reference's own tests and benches (ola_stark.rs:684, fibo_loop.rs:26)
type F = <C as GenericConfig<D>>::F;

fn write_traces(path: &PathBuf, traces: &[Vec<PolynomialValues<F>>; NUM_TABLES], compress: &[F]) {
    let mut f = File::create(path).unwrap();
    f.write_all(b"OLAPIN01").unwrap();
    f.write_all(&(NUM_TABLES as u32).to_le_bytes()).unwrap();
    for t in traces.iter() {
        let rows = t[0].values.len();
        assert!(rows.is_power_of_two());
        f.write_all(&(t.len() as u32).to_le_bytes()).unwrap();
        f.write_all(&(rows.trailing_zeros() as u32).to_le_bytes()).unwrap();
        for col in t.iter() {
            assert_eq!(col.values.len(), rows);
            for v in col.values.iter() {
                f.write_all(&v.to_canonical_u64().to_le_bytes()).unwrap();
            }
        }
    }
    f.write_all(&(compress.len() as u32).to_le_bytes()).unwrap();
    for c in compress {
        f.write_all(&c.to_canonical_u64().to_le_bytes()).unwrap();
    }
}

/// <name>.diag: what a first mismatch is diagnosed with in ONE run (compare_with_dump.py reads it): the reference's own
/// cross-table-lookup challenges (get_challenges.rs:23-33) and the three Merkle caps of every table, hex of write_merkle_cap.
fn write_diag(path: &PathBuf, proof: &crate::stark::proof::AllProof<F, C, D>, ola_stark: &OlaStark<F, D>, config: &StarkConfig) {
    let ch = proof.get_challenges(ola_stark, config);
    let mut s = String::from("OLADIAG01\n");
    for c in ch.ctl_challenges.challenges.iter() {
        s += &format!("ctl_challenge {} {}\n", c.beta.to_canonical_u64(), c.gamma.to_canonical_u64());
    }
    for (i, p) in proof.stark_proofs.iter().enumerate() {
        for (name, cap) in [("trace_cap", &p.trace_cap), ("zs_cap", &p.permutation_ctl_zs_cap), ("quotient_cap", &p.quotient_polys_cap)] {
            let mut b = Buffer::new(Vec::new());
            b.write_merkle_cap(cap).unwrap();
            let hex: String = b.bytes().iter().map(|x| format!("{:02x}", x)).collect();
            s += &format!("table {} {} {}\n", i, name, hex);
        }
    }
    File::create(path).unwrap().write_all(s.as_bytes()).unwrap();
}

fn dump_one(file_name: &str, call_data: Option<Vec<GoldilocksField>>) {
    dump_named(file_name.trim_end_matches(".json"), file_name, call_data);
}

/// Runs `file_name` and returns the height (log2 rows) of the CPU table its execution fills, without proving.
fn cpu_table_bits(file_name: &str, call_data: Option<Vec<GoldilocksField>>) -> usize {
    std::env::set_var("OLA_PIN_HEIGHT_ONLY", "1");
    let bits = dump_named("_probe", file_name, call_data);
    std::env::remove_var("OLA_PIN_HEIGHT_ONLY");
    bits
}

fn dump_named(stem: &str, file_name: &str, call_data: Option<Vec<GoldilocksField>>) -> usize {
    let out_dir = PathBuf::from(std::env::var("OLA_PIN_DIR").unwrap_or_else(|_| "/tmp/ola_pin".to_string()));
    std::fs::create_dir_all(&out_dir).unwrap();
    // ---- exactly ola_stark.rs::test_by_asm_json up to generate_traces ----
    let mut path = PathBuf::from(env!("CARGO_MANIFEST_DIR"));
    path.push("../assembler/test_data/asm/");
    path.push(file_name);
    let mut db = AccountTree::new_test();
    let program = encode_asm_from_json_file(path.display().to_string()).unwrap();
    let hash = ZkHasher::default();
    let instructions = program.bytecode.split("\n");
    let code: Vec<_> = instructions
        .clone()
        .map(|e| GoldilocksField::from_canonical_u64(u64::from_str_radix(&e[2..], 16).unwrap()))
        .collect();
    let code_hash = hash.hash_bytes(&code);
    let mut prophets = HashMap::new();
    for item in program.prophets {
        prophets.insert(item.host as u64, item);
    }
    let mut program: Program = Program::default();
    for inst in instructions {
        program.instructions.push(inst.to_string());
    }
    let mut process = Process::new();
    let callee: Address = [9u64, 10, 11, 12].map(GoldilocksField::from_canonical_u64);
    let caller_addr = [17u64, 18, 19, 20].map(GoldilocksField::from_canonical_u64);
    let callee_exe_addr = [13u64, 14, 15, 16].map(GoldilocksField::from_canonical_u64);
    if let Some(calldata) = call_data {
        process.tp = GoldilocksField::from_canonical_u64(0);
        init_tape(&mut process, calldata, caller_addr, callee, callee_exe_addr, &init_tx_context_mock());
    }
    process.addr_code = callee_exe_addr;
    process.addr_storage = callee;
    program.trace.addr_program_hash.insert(encode_addr(&callee_exe_addr), code);
    db.process_block(vec![WitnessStorageLog {
        storage_log: StorageLog::new_write_log(callee_exe_addr, code_hash),
        previous_value: tree_key_default(),
    }]);
    let _ = db.save();
    let start = db.root_hash();
    process.program_log.push(WitnessStorageLog {
        storage_log: StorageLog::new_read_log(callee_exe_addr, code_hash),
        previous_value: tree_key_default(),
    });
    program.prophets = prophets;
    process.execute(&mut program, &mut db, &mut TxScopeCacheManager::default()).expect("execute");
    let hash_roots = gen_storage_hash_table(&mut process, &mut program, &mut db);
    gen_storage_table(&mut process, &mut program, hash_roots).unwrap();
    program.trace.start_end_roots = (start, db.root_hash());

    let mut ola_stark = OlaStark::default();
    let (traces, public_values) = generate_traces(program, &mut ola_stark, GenerationInputs::default());
    let cpu_bits = traces[0][0].values.len().trailing_zeros() as usize;
    if std::env::var("OLA_PIN_HEIGHT_ONLY").is_ok() {
        return cpu_bits;
    }
    let config = StarkConfig::standard_fast_config();
    // the traces as the prover receives them; the compress challenges are read back from the proof below
    let traces_copy: [Vec<PolynomialValues<F>>; NUM_TABLES] = traces.clone();
    let proof_b3 = prove_with_traces::<F, CB, D>(&ola_stark, &config, traces.clone(), public_values.clone(), &mut TimingTree::default())
        .expect("prove (Blake3GoldilocksConfig)");
    let proof = prove_with_traces::<F, C, D>(&ola_stark, &config, traces, public_values, &mut TimingTree::default()).expect("prove");

    write_diag(&out_dir.join(format!("{stem}.diag")), &proof, &OlaStark::default(), &config);
    write_traces(&out_dir.join(format!("{stem}.traces")), &traces_copy, &proof.compress_challenges);
    let mut buf = Buffer::new(Vec::new());
    buf.write_all_proof(&proof).unwrap();
    File::create(out_dir.join(format!("{stem}.proof"))).unwrap().write_all(&buf.bytes()).unwrap();
    // The same traces under Blake3GoldilocksConfig -> <name>.blake3.proof.  The GPU backend hashes CANONICAL words, the CPU prover
    // the words as they lie in memory (hash/blake3.rs:204-207); at these sizes (< 10^7 hashed words, a non-canonical word every
    // ~2^32 field operations) the two agree, and the reference verifier -- which hashes deserialised, canonical words -- says so.
    let mut buf_b3 = Buffer::new(Vec::new());
    buf_b3.write_all_proof(&proof_b3).unwrap();
    File::create(out_dir.join(format!("{stem}.blake3.proof"))).unwrap().write_all(&buf_b3.bytes()).unwrap();
    verify_proof(OlaStark::default(), proof_b3, &config).expect("the reference verifier must accept its own Blake3 proof");
    let degree_bits = proof.degree_bits(&config);
    verify_proof(OlaStark::default(), proof, &config).expect("the reference verifier must accept its own proof");
    println!("pin_dump: {stem}: table heights 2^{:?}, {} proof bytes", degree_bits, buf.len());
    cpu_bits
}

#[test]
fn pin_dump() {
    // programs and calldata exactly as the reference's own tests run them (ola_stark.rs:691-760); fib_asm's first calldata word is
    // the loop count -- raise it for larger CPU tables
    let words = |v: &[u64]| v.iter().map(|x| GoldilocksField::from_canonical_u64(*x)).collect::<Vec<_>>();
    dump_one("fibo_recursive.json", None);
    dump_one("memory.json", None);
    dump_one("fib_asm.json", Some(words(&[10, 1, 2, 4185064725])));
    dump_one("sqrt_prophet_asm.json", Some(words(&[144, 10, 2, 3509365327])));
    // BASELINE config 1: the Fibonacci program at a 2^12-row CPU table.  The loop count that fills 2^12 rows is found by running
    // the executor (no proving) with doubling counts; the dump is then made at that count.
    let mut loops = 16u64;
    while cpu_table_bits("fib_asm.json", Some(words(&[loops, 1, 2, 4185064725]))) < 12 && loops < (1 << 20) {
        loops *= 2;
    }
    assert_eq!(cpu_table_bits("fib_asm.json", Some(words(&[loops, 1, 2, 4185064725]))), 12, "no loop count gives a 2^12-row CPU table");
    println!("pin_dump: fib_asm with {loops} iterations fills a 2^12-row CPU table");
    dump_named("fib_asm_2p12", "fib_asm.json", Some(words(&[loops, 1, 2, 4185064725])));
}
