/* Native trace generator: runs a small OlaVM program on the host and fills the 12 STARK tables so that every
 * constraint and cross-table lookup of OlaStark holds -- the job of the reference's Rust executor + circuits/src/generation
 * for the instructions it supports: MOV NOT ADD MUL EQ NEQ ASSERT JMP CJMP CALL RET MLOAD MSTORE RC AND OR XOR GTE POSEIDON
 * TSTORE TLOAD SSTORE SLOAD END -- everything but cross-contract calls (for which the reference AIR admits no trace, see
 * DESIGN.md) and SIGCHECK.  It reproduces the Python executor olavm_amd/air/miniexec.py word for word.  Host-only C ABI,
 * libola_tracegen.so; the traces go straight into ola_prove_with_traces.
 *
 * Restates: core/src/vm/opcodes.rs:81-114 (opcode masks), circuits/src/cpu/cpu_stark.rs:529-581 (instruction encoding),
 * executor/src/lib.rs (instruction semantics), circuits/src/generation/{cpu,memory,builtin,poseidon,poseidon_chunk,prog}.rs
 * (row layouts and padding), circuits/src/stark/lookup.rs:68-132 (permuted columns). */
#ifndef OLA_TRACEGEN_H
#define OLA_TRACEGEN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One instruction: op = bit position of the opcode mask (core/src/vm/opcodes.rs: ADD 31, MUL 30, EQ 29, ASSERT 28, MOV 27,
 * JMP 26, CJMP 25, CALL 24, RET 23, MLOAD 22, MSTORE 21, END 20, RC 19, AND 18, OR 17, XOR 16, NOT 15, NEQ 14, GTE 13,
 * POSEIDON 12, SLOAD 11, SSTORE 10, TLOAD 9, TSTORE 8); dst / op0 / op1 = register index 0..9 or -1; op1_is_imm != 0: the second operand is `imm` (a two-word
 * instruction). */
typedef struct OlaInstr {
    uint32_t op;
    int32_t dst, op0, op1;
    uint32_t op1_is_imm;
    uint64_t imm;
} OlaInstr;

typedef struct OlaTraceSet OlaTraceSet;

/* flags: close the program-hash chain with a result line and a state-tree proof that the hash is the leaf at the code
 * address (256 storage rows, 512 Poseidon rows) */
#define OLA_TRACEGEN_PROVE_PROGRAM_HASH 1u
/* flags: take the compress challenges of the bitwise and program tables from the bitwise_beta / program_beta arguments (tests).
 * Without it they are derived as the reference derives them -- a Fiat-Shamir transcript observes the limb columns of the bitwise
 * table (generation/builtin.rs:120-131) and the state roots before and after the run (generation/prog.rs:23-29) -- and the two
 * arguments are ignored; ola_tracegen_betas returns the values used (they are the proof's compress_challenges). */
#define OLA_TRACEGEN_EXPLICIT_BETAS 2u
/* flags: reproduce the two places where the reference's generators write rows its own AIR rejects, for comparing whole pipelines with
 * a build of the reference (integration/pin/) -- not for proving: (a) the fourth limb of a bitwise operand / result is left zero
 * (generation/builtin.rs:66,71,76 store it at `OP*_LIMBS.end`, the exclusive end of the column range, and the next write replaces it);
 * (b) the memory table of a run without memory cells is generation/memory.rs:95-153's: every row a prophet-region row, row 0 without
 * its selector (default: one stack-region row first, so that memory_stark.rs:265-270 hold on the wrap-around). */
#define OLA_TRACEGEN_REFERENCE_QUIRKS 4u

/* Executes the program (at most max_steps CPU rows) and builds the 12 tables of ola_stark(range_bits, limb_bits) in
 * `enum Table` order.  range_bits / limb_bits are 16 / 8 in the reference; smaller values give structurally identical
 * miniature fixed tables.  Returns 0, or a negative code with a message in ola_tracegen_last_error(). */
int32_t ola_tracegen_run(const OlaInstr* program, size_t n_instr, const uint64_t code_addr[4], const uint64_t storage_addr[4],
                         uint32_t range_bits, uint32_t limb_bits, uint64_t bitwise_beta, uint64_t program_beta, uint64_t max_steps,
                         uint32_t flags, OlaTraceSet** out);
/* Table t: column-major ncols x 2^log_n words, owned by the set. */
int32_t ola_tracegen_table(const OlaTraceSet* set, uint32_t table, uint32_t* ncols, uint32_t* log_n, const uint64_t** data);
/* Number of executed CPU rows (before padding). */
uint64_t ola_tracegen_cpu_rows(const OlaTraceSet* set);
/* out[0] = the bitwise table's compress challenge, out[1] = the program table's. */
int32_t ola_tracegen_betas(const OlaTraceSet* set, uint64_t out[2]);
void ola_tracegen_free(OlaTraceSet* set);
const char* ola_tracegen_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* OLA_TRACEGEN_H */
