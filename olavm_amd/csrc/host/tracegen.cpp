// Native trace generator behind include/ola_tracegen.h (host only, no HIP): the miniature executor
// olavm_amd/air/miniexec.py in C++, word for word -- tests/test_tracegen_native.py compares all twelve tables of both on
// the same programs.  Reference rules restated here are cited where they apply; the column indices come from the Python
// table descriptions through the generated header gen/ola_columns.h.
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/ola_poseidon_constants.h"
#include "../../../include/ola_tracegen.h"
#include "../gen/ola_columns.h"

namespace {

using namespace olacols;
typedef uint64_t u64;
typedef unsigned __int128 u128;

const u64 P = 0xFFFFFFFF00000001ULL;
// arguments may be any u64, results are canonical; the 128-bit reduction uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p)
inline u64 canon(u64 x) { return x >= P ? x - P : x; }
inline u64 addm(u64 a, u64 b) { a = canon(a); b = canon(b); const u64 s = a + b; return (s < a || s >= P) ? s - P : s; }
inline u64 subm(u64 a, u64 b) { a = canon(a); b = canon(b); return a >= b ? a - b : a + (P - b); }
inline u64 mulm(u64 a, u64 b) {
    const u128 x = (u128)a * b;
    const u64 lo = (u64)x, hi = (u64)(x >> 64), hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFULL;                 // borrowed 2^64 = p + (2^32 - 1): take the extra back
    const u64 t1 = hl * 0xFFFFFFFFULL;
    u64 t2 = t0 + t1;
    if (t2 < t1) t2 += 0xFFFFFFFFULL;                 // carried 2^64 = 2^32 - 1 (mod p)
    return canon(t2);
}
u64 powm(u64 b, u64 e) { u64 r = 1; b %= P; while (e) { if (e & 1) r = mulm(r, b); b = mulm(b, b); e >>= 1; } return r; }
inline u64 invm(u64 x) { return x % P ? powm(x % P, P - 2) : 0; }
inline size_t next_pow2(size_t n) { size_t k = 2; while (k < n) k <<= 1; return k; }

struct Err : std::runtime_error { using std::runtime_error::runtime_error; };
void need(bool ok, const char* what) { if (!ok) throw Err(what); }

// A table: ncols columns of n words, column-major (what ola_prove_with_traces takes).
struct Table {
    size_t ncols = 0, n = 0;
    std::vector<u64> d;
    void init(size_t c, size_t rows) { ncols = c; n = rows; d.assign(c * rows, 0); }
    u64& at(size_t col, size_t row) { return d[col * n + row]; }
    void fill(size_t col, u64 v) { std::fill(d.begin() + col * n, d.begin() + (col + 1) * n, v); }
    std::vector<u64> column(size_t col) const { return std::vector<u64>(d.begin() + col * n, d.begin() + (col + 1) * n); }
    void set_column(size_t col, const std::vector<u64>& v) { std::copy(v.begin(), v.end(), d.begin() + col * n); }
};

// ---- Poseidon with the S-box inputs recorded: one row of the Poseidon table (generation/poseidon.rs:5-80; the
// permutation is the plain round structure of plonky2/src/hash/poseidon.rs:617-627)
template <bool RECORD>
void poseidon_core(u64 s[12], u64* row) {
    auto sbox = [](u64 x) { u64 x2 = mulm(x, x), x4 = mulm(x2, x2), x3 = mulm(x2, x); return mulm(x3, x4); };
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        for (int i = 0; i < 12; i++) s[i] = addm(s[i], OLA_POSEIDON_RC[12 * r + i]);
        if (full) {
            if (RECORD) {
                if (r >= 1 && r <= 3) for (int i = 0; i < 12; i++) row[COL_POSEIDON_FULL_ROUND_0_1_STATE_RANGE_START + 12 * (r - 1) + i] = s[i];
                else if (r >= 26) for (int i = 0; i < 12; i++) row[COL_POSEIDON_FULL_ROUND_1_0_STATE_RANGE_START + 12 * (r - 26) + i] = s[i];
            }
            for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        } else {
            if (RECORD) row[COL_POSEIDON_PARTIAL_ROUND_ELEMENT_RANGE_START + (r - 4)] = s[0];
            s[0] = sbox(s[0]);
        }
        u64 o[12];
        for (int k = 0; k < 12; k++) {
            u128 acc = 0;
            for (int i = 0; i < 12; i++) acc += (u128)s[(i + k) % 12] * OLA_POSEIDON_MDS_CIRC[i];
            acc += (u128)s[k] * OLA_POSEIDON_MDS_DIAG[k];
            o[k] = addm(mulm((u64)(acc >> 64), 0xFFFFFFFFULL), canon((u64)acc));      // acc = hi * 2^64 + lo, 2^64 = 2^32 - 1
        }
        memcpy(s, o, sizeof(o));
    }
}

std::vector<u64> poseidon_row(const u64* in12, const u64 filters[4]) {
    std::vector<u64> row(NUM_POSEIDON_COLS, 0);
    for (int i = 0; i < 4; i++) row[i] = filters[i];
    u64 s[12];
    for (int i = 0; i < 12; i++) { s[i] = in12[i] % P; row[COL_POSEIDON_INPUT_RANGE_START + i] = s[i]; }
    poseidon_core<true>(s, row.data());
    for (int i = 0; i < 12; i++) row[COL_POSEIDON_OUTPUT_RANGE_START + i] = s[i];
    return row;
}

// the permutation alone (state in, state out)
void poseidon_permute(u64 s[12]) {
    for (int i = 0; i < 12; i++) s[i] %= P;
    poseidon_core<false>(s, nullptr);
}

// ---- lookup.rs:68-132
// Fiat-Shamir transcript of the trace generators (iop/challenger.rs:36-162: overwrite-mode duplex sponge, rate 8, challenges
// popped from the end of the output buffer) -- the compress challenges of the bitwise and program tables come out of it
// (generation/builtin.rs:120-131, generation/prog.rs:23-29), not from the caller.
struct HostChallenger {
    u64 state[12] = {0};
    u64 in[8], out[8];
    int nin = 0, nout = 0;
    void duplex() {
        for (int i = 0; i < nin; i++) state[i] = in[i];
        nin = 0;
        poseidon_permute(state);
        for (int i = 0; i < 8; i++) out[i] = state[i];
        nout = 8;
    }
    void observe(u64 e) {
        nout = 0;
        in[nin++] = e % P;
        if (nin == 8) duplex();
    }
    u64 get() {
        if (nin != 0 || nout == 0) duplex();
        return out[--nout];
    }
};

void permuted_cols(const std::vector<u64>& inputs, const std::vector<u64>& table, std::vector<u64>& pi, std::vector<u64>& pt) {
    const size_t n = inputs.size();
    pi = inputs;
    std::vector<u64> st = table;
    for (auto& x : pi) x %= P;
    for (auto& x : st) x %= P;
    std::sort(pi.begin(), pi.end());
    std::sort(st.begin(), st.end());
    pt.assign(n, 0);
    std::vector<size_t> unused_inds;
    std::vector<u64> unused_vals;
    size_t i = 0, j = 0;
    while (j < n && i < n) {
        const u64 a = pi[i], b = st[j];
        if (a > b) { unused_vals.push_back(b); j++; }
        else if (a < b) {
            if (!unused_vals.empty()) { pt[i] = unused_vals.back(); unused_vals.pop_back(); } else unused_inds.push_back(i);
            i++;
        } else { pt[i] = b; i++; j++; }
    }
    for (; j < n; j++) unused_vals.push_back(st[j]);
    for (; i < n; i++) unused_inds.push_back(i);
    need(unused_inds.size() == unused_vals.size(), "permuted_cols: unused slots and values differ in number");
    for (size_t k = 0; k < unused_inds.size(); k++) pt[unused_inds[k]] = unused_vals[k];
}

// ---- execution -----------------------------------------------------------------------------------------------------------------
enum MemOp { M_CALL, M_MLOAD, M_MSTORE, M_POSEIDON, M_RET, M_SLOAD, M_SSTORE, M_TLOAD, M_TSTORE };   // alphabetical: ties of (address, clock) sort by name, as in Python
struct MemCell { u64 addr, clk; MemOp op; u64 value; int is_write; };
struct PsdnChunk { u64 addr; u64 vals[8]; u64 cap[4]; std::vector<u64> row; };
struct PsdnCall { u64 clk, src, len, dst; std::vector<PsdnChunk> chunks; };
struct BwOp { uint32_t op; u64 a, b; };
struct TapeCell { u64 addr, seq; uint32_t op; u64 word; };
typedef std::array<u64, 4> Hash4;

// ---- the account-storage tree: 256 levels, key bits (four limbs, most significant first) choose the child, inner nodes
// Poseidon(left || right || [0,0,0,0])[:4], the lowest level hashes the two 4-word VALUES of a sibling pair with the
// capacity word 1, untouched leaves are [0,0,0,0] (builtins/storage/storage_access_stark.rs:110-334)
struct StorageRow { int layer, bit; Hash4 sib, pre_path, path, pre_hash, hash, pre_root, root, addr; int is_write; };
struct StorageTree {
    typedef std::array<u64, 5> NodeKey;                       // depth, prefix (bits below the depth cleared)
    std::map<NodeKey, Hash4> nodes;
    std::vector<Hash4> dflt;
    static Hash4 hash(const Hash4& l, const Hash4& r, bool leaf_level, std::vector<u64>* row_out = nullptr) {
        u64 in[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], (u64)leaf_level, 0, 0, 0};
        if (!row_out) { poseidon_permute(in); return Hash4{in[0], in[1], in[2], in[3]}; }
        const u64 f[4] = {0, 0, (u64)leaf_level, (u64)!leaf_level};
        *row_out = poseidon_row(in, f);
        const std::vector<u64>& row = *row_out;
        return Hash4{row[COL_POSEIDON_OUTPUT_RANGE_START], row[COL_POSEIDON_OUTPUT_RANGE_START + 1], row[COL_POSEIDON_OUTPUT_RANGE_START + 2],
                     row[COL_POSEIDON_OUTPUT_RANGE_START + 3]};
    }
    StorageTree() : dflt(257) {
        dflt[256] = Hash4{0, 0, 0, 0};
        for (int d = 255; d >= 0; d--) dflt[d] = hash(dflt[d + 1], dflt[d + 1], d == 255);
    }
    static Hash4 key_of(const Hash4& addr) { return Hash4{addr[0] % P, addr[1] % P, addr[2] % P, addr[3] % P}; }
    static int bit_at(const Hash4& k, int layer) { return (int)((k[(layer - 1) / 64] >> (63 - (layer - 1) % 64)) & 1); }   // layer-th bit from the top
    static NodeKey node_key(const Hash4& k, int depth) {
        NodeKey nk{(u64)depth, 0, 0, 0, 0};
        for (int w = 0; w < 4; w++) {
            const int keep = std::min(64, std::max(0, depth - 64 * w));
            nk[1 + w] = keep == 0 ? 0 : keep == 64 ? k[w] : (k[w] & ~((1ULL << (64 - keep)) - 1));
        }
        return nk;
    }
    static Hash4 flip(Hash4 k, int layer) { k[(layer - 1) / 64] ^= 1ULL << (63 - (layer - 1) % 64); return k; }
    Hash4 node(const Hash4& k, int depth) const { auto it = nodes.find(node_key(k, depth)); return it == nodes.end() ? dflt[depth] : it->second; }
    Hash4 root() const { return node(Hash4{0, 0, 0, 0}, 0); }
    void write(const Hash4& k, const Hash4& value) {
        nodes[node_key(k, 256)] = value;
        for (int d = 255; d >= 0; d--) {
            // children of the depth-d node on k's path: bit (d+1) cleared / set
            Hash4 l = k, r = k;
            if (bit_at(k, d + 1)) l = flip(k, d + 1); else r = flip(k, d + 1);
            nodes[node_key(k, d)] = hash(node(l, d + 1), node(r, d + 1), d == 255);
        }
    }
    // read (value == nullptr) or write of a leaf: 256 table rows (root side first) and, per layer, the Poseidon rows of the
    // new-tree and of the old-tree hash
    Hash4 access(const Hash4& addr, const Hash4* value, std::vector<StorageRow>& rows, std::vector<std::vector<u64>>& prows) {
        const Hash4 k = key_of(addr);
        struct Pre { int bit; Hash4 sib, path, hash; };
        std::vector<Pre> pre;
        for (int layer = 1; layer <= 256; layer++) pre.push_back({bit_at(k, layer), node(flip(k, layer), layer), node(k, layer), node(k, layer - 1)});
        const Hash4 pre_root = root();
        if (value) write(k, *value);
        for (int layer = 1; layer <= 256; layer++) {
            const Pre& p = pre[layer - 1];
            StorageRow r;
            r.layer = layer; r.bit = p.bit; r.sib = p.sib; r.pre_path = p.path; r.pre_hash = p.hash;
            r.path = node(k, layer); r.hash = node(k, layer - 1); r.pre_root = pre_root; r.root = root(); r.is_write = value != nullptr; r.addr = k;
            rows.push_back(r);
            const Hash4* child[2] = {&r.path, &r.pre_path};
            const Hash4* expect[2] = {&r.hash, &r.pre_hash};
            for (int v = 0; v < 2; v++) {
                std::vector<u64> prow;
                const Hash4 h = p.bit ? hash(p.sib, *child[v], layer == 256, &prow) : hash(*child[v], p.sib, layer == 256, &prow);
                need(h == *expect[v], "storage tree: inconsistent node hash");
                prows.push_back(std::move(prow));
            }
        }
        return node(k, 256);
    }
};

struct Run {
    // executed CPU rows, row-major (NUM_CPU_COLS words each), in chunks so that a long run never re-copies what it has
    static constexpr size_t CHUNK_ROWS = 1 << 14;
    std::vector<std::unique_ptr<u64[]>> row_chunks;
    size_t n_rows = 0;
    size_t nrows() const { return n_rows; }
    const u64* row(size_t i) const { return row_chunks[i / CHUNK_ROWS].get() + (i % CHUNK_ROWS) * NUM_CPU_COLS; }
    void push_row(const std::vector<u64>& r) {
        if (n_rows % CHUNK_ROWS == 0) row_chunks.emplace_back(new u64[CHUNK_ROWS * NUM_CPU_COLS]);
        std::copy(r.begin(), r.end(), row_chunks.back().get() + (n_rows % CHUNK_ROWS) * NUM_CPU_COLS);
        n_rows++;
    }
    std::vector<std::pair<u64, u64>> executed;           // fetched (pc, word)
    std::vector<u64> rc;
    std::vector<BwOp> bitwise;
    std::vector<std::pair<u64, u64>> cmp;
    std::vector<MemCell> mem;
    std::vector<PsdnCall> psdn;
    std::vector<TapeCell> tape;
    std::vector<std::vector<StorageRow>> storage;        // one 256-row proof per SSTORE / SLOAD
    std::vector<std::vector<u64>> storage_psdn;          // tree-key and state-tree rows of the Poseidon table
    std::vector<u64> words;
};

const int REG = 10;

void program_words(const OlaInstr* ins, size_t n, std::vector<u64>& words, std::vector<size_t>& pcs) {
    for (size_t k = 0; k < n; k++) {
        const OlaInstr& I = ins[k];
        need(I.op < 32 && I.dst < REG && I.op0 < REG && I.op1 < REG, "instruction field out of range");
        u64 w = I.op1_is_imm ? (1ULL << 62) : 0;
        if (I.op0 >= 0) w += 1ULL << (52 + I.op0);
        if (I.op1 >= 0 && !I.op1_is_imm) w += 1ULL << (42 + I.op1);
        if (I.dst >= 0) w += 1ULL << (32 + I.dst);
        w += 1ULL << I.op;
        pcs.push_back(words.size());
        words.push_back(w);
        if (I.op1_is_imm) words.push_back(I.imm % P);
    }
}

u64 selector_of(uint32_t op) {
    switch (op) {
        case OP_ADD: case OP_MUL: case OP_EQ: case OP_NEQ: case OP_ASSERT: return COL_S_SIMPLE_ARITHMATIC_OP;
        case OP_MOV: return COL_S_MOV;
        case OP_NOT: return COL_S_NOT;
        case OP_JMP: return COL_S_JMP;
        case OP_CJMP: return COL_S_CJMP;
        case OP_CALL: return COL_S_CALL;
        case OP_RET: return COL_S_RET;
        case OP_MLOAD: return COL_S_MLOAD;
        case OP_MSTORE: return COL_S_MSTORE;
        case OP_END: return COL_S_END;
        case OP_RC: return COL_S_RC;
        case OP_AND: case OP_OR: case OP_XOR: return COL_S_BITWISE;
        case OP_GTE: return COL_S_GTE;
        case OP_POSEIDON: return COL_S_PSDN;
        case OP_SSTORE: return COL_S_SSTORE;
        case OP_SLOAD: return COL_S_SLOAD;
        case OP_TSTORE: return COL_S_TSTORE;
        case OP_TLOAD: return COL_S_TLOAD;
        default: throw Err("instruction not supported by the native generator");
    }
}

void execute(const OlaInstr* ins, size_t n_ins, const u64 code_addr[4], const u64 storage_addr[4], u64 max_steps, StorageTree& tree, Run& R) {
    std::vector<size_t> pcs;
    program_words(ins, n_ins, R.words, pcs);
    std::map<u64, size_t> pc_to_idx;
    for (size_t k = 0; k < pcs.size(); k++) pc_to_idx[pcs[k]] = k;
    u64 regs[REG] = {0};
    u64 pc = 0, clk = 0, tp = 0, idx_storage = 0;
    std::map<u64, u64> memory, tape;
    std::vector<u64> r(NUM_CPU_COLS);
    auto mem_at = [&](u64 a, const char* what) -> u64 { auto it = memory.find(a); need(it != memory.end(), what); return it->second; };
    for (;;) {
        need(R.nrows() < max_steps, "program does not terminate");
        auto it = pc_to_idx.find(pc);
        need(it != pc_to_idx.end(), "jump into the middle of an instruction");
        const OlaInstr& I = ins[it->second];
        const bool imm = I.op1_is_imm != 0;
        std::fill(r.begin(), r.end(), 0);
        for (int i = 0; i < 4; i++) { r[COL_ADDR_STORAGE_RANGE_START + i] = storage_addr[i]; r[COL_ADDR_CODE_RANGE_START + i] = code_addr[i]; }
        r[COL_CLK] = clk; r[COL_PC] = pc; r[COL_TP] = tp; r[COL_IDX_STORAGE] = idx_storage;
        for (int i = 0; i < REG; i++) r[COL_REGS_START + i] = regs[i];
        r[COL_INST] = R.words[pc]; r[COL_OP1_IMM] = imm; r[COL_OPCODE] = 1ULL << I.op;
        r[selector_of(I.op)] = 1;
        r[COL_IS_ENTRY_SC] = r[COL_IS_NEXT_LINE_DIFF_INST] = r[COL_IS_NEXT_LINE_SAME_TX] = 1;
        const u64 v0 = I.op0 >= 0 ? regs[I.op0] : 0;
        const u64 v1 = imm ? I.imm % P : (I.op1 >= 0 ? regs[I.op1] : 0);
        if (I.op0 >= 0) { r[COL_S_OP0_START + I.op0] = 1; r[COL_OP0] = v0; }
        if (imm) { r[COL_IMM_VAL] = r[COL_OP1] = v1; r[COL_FILTER_LOOKING_PROG_IMM] = 1; }
        else if (I.op1 >= 0) { r[COL_S_OP1_START + I.op1] = 1; r[COL_OP1] = v1; }
        const u64 size = imm ? 2 : 1;
        R.executed.push_back({pc, R.words[pc]});
        if (imm) R.executed.push_back({pc + 1, v1});
        u64 next_pc = pc + size, res = 0;
        bool has_res = false;
        switch (I.op) {
            case OP_MOV: res = v1; has_res = true; break;
            case OP_NOT: res = subm(P - 1, v1); has_res = true; break;                          // executor/src/lib.rs:602-605
            case OP_ASSERT: need(v1 == 1, "ASSERT on a value other than 1"); break;               // :673-712
            case OP_ADD: res = addm(v0, v1); has_res = true; break;
            case OP_MUL: res = mulm(v0, v1); has_res = true; break;
            case OP_EQ: case OP_NEQ:
                res = ((v0 == v1) == (I.op == OP_EQ)) ? 1 : 0; has_res = true;
                r[COL_AUX0] = invm(subm(v0, v1));
                break;
            case OP_JMP: next_pc = v1; break;
            case OP_CJMP: need(v0 <= 1, "CJMP on a non-boolean"); next_pc = v0 ? v1 : pc + size; break;
            case OP_RC: need(v1 < (1ULL << 32), "RC operand too wide"); R.rc.push_back(v1); break;
            case OP_AND: case OP_OR: case OP_XOR:
                need(v0 < (1ULL << 32) && v1 < (1ULL << 32), "bitwise operand too wide");
                res = I.op == OP_AND ? (v0 & v1) : I.op == OP_OR ? (v0 | v1) : (v0 ^ v1); has_res = true;
                R.bitwise.push_back({I.op, v0, v1});
                break;
            case OP_GTE:
                need(v0 < (1ULL << 32) && v1 < (1ULL << 32), "GTE operand too wide");
                res = v0 >= v1; has_res = true;
                R.cmp.push_back({v0, v1});
                break;
            case OP_MSTORE: case OP_MLOAD: {
                // executor/src/lib.rs:868-995: address = op0 + immediate offset (aux1)
                need(imm && I.dst >= 0, "only the [reg + imm] addressing form is implemented");
                const u64 addr = addm(v0, v1);
                r[COL_AUX1] = addr;
                if (I.op == OP_MSTORE) { res = regs[I.dst]; memory[addr] = res; }
                else res = mem_at(addr, "load from an address that was never written");
                has_res = true;
                R.mem.push_back({addr, clk, I.op == OP_MSTORE ? M_MSTORE : M_MLOAD, res, I.op == OP_MSTORE});
                break;
            }
            case OP_CALL: {
                // executor/src/lib.rs:816-849: the return address goes to [fp - 1]; [fp - 2] (the caller's saved fp) is read
                const u64 fp = regs[REG - 1], a1 = subm(fp, 1), a2 = subm(fp, 2);
                need(imm, "CALL needs an immediate target");
                const u64 saved = mem_at(a2, "CALL needs a saved frame pointer at [fp - 2]");
                const u64 ret_pc = pc + size;
                r[COL_OP0] = a1; r[COL_DST] = ret_pc; r[COL_AUX0] = a2; r[COL_AUX1] = saved;
                memory[a1] = ret_pc;
                R.mem.push_back({a1, clk, M_CALL, ret_pc, 1});
                R.mem.push_back({a2, clk, M_CALL, saved, 0});
                next_pc = v1;
                break;
            }
            case OP_RET: {
                // executor/src/lib.rs:851-866: pc <- [fp - 1], fp <- [fp - 2]
                const u64 fp = regs[REG - 1], a1 = subm(fp, 1), a2 = subm(fp, 2);
                const u64 ret_pc = mem_at(a1, "RET without a return address"), old_fp = mem_at(a2, "RET without a saved frame pointer");
                r[COL_OP0] = a1; r[COL_DST] = ret_pc; r[COL_AUX0] = a2; r[COL_AUX1] = old_fp;
                R.mem.push_back({a1, clk, M_RET, ret_pc, 0});
                R.mem.push_back({a2, clk, M_RET, old_fp, 0});
                regs[REG - 1] = old_fp;
                next_pc = ret_pc;
                break;
            }
            case OP_POSEIDON: {
                // executor/src/lib.rs:1547-1700: hash `len` words at [op0..] eight at a time, capacity chained, digest to [dst..]
                need(I.dst >= 0, "POSEIDON needs a destination register");
                const u64 src = v0, length = v1, dst_addr = regs[I.dst];
                need(length && length % 8 == 0, "only whole 8-word blocks are implemented");
                PsdnCall call{clk, src, length, dst_addr, {}};
                u64 cap[4] = {0, 0, 0, 0};
                const u64 filt[4] = {1, 0, 0, 0};
                for (u64 k = 0; k < length; k += 8) {
                    PsdnChunk ch;
                    ch.addr = src + k;
                    u64 in[12];
                    for (int i = 0; i < 8; i++) {
                        ch.vals[i] = in[i] = mem_at(src + k + i, "hash input was never written");
                        R.mem.push_back({src + k + i, clk, M_POSEIDON, in[i], 0});
                    }
                    for (int i = 0; i < 4; i++) ch.cap[i] = in[8 + i] = cap[i];
                    ch.row = poseidon_row(in, filt);
                    for (int i = 0; i < 4; i++) cap[i] = ch.row[COL_POSEIDON_OUTPUT_RANGE_START + 8 + i];
                    call.chunks.push_back(std::move(ch));
                }
                for (int i = 0; i < 4; i++) {
                    const u64 o = call.chunks.back().row[COL_POSEIDON_OUTPUT_RANGE_START + i];
                    memory[dst_addr + i] = o;
                    R.mem.push_back({dst_addr + i, clk, M_POSEIDON, o, 1});
                }
                R.psdn.push_back(std::move(call));
                res = dst_addr; has_res = true;       // the CPU's dst column carries the destination address
                break;
            }
            case OP_END: case OP_TSTORE: case OP_SSTORE: case OP_SLOAD: break;
            case OP_TLOAD: need(I.dst >= 0, "TLOAD needs the register holding the memory base"); res = regs[I.dst]; has_res = true; break;
            default: throw Err("instruction not supported by the native generator");
        }
        if (I.dst >= 0) {
            need(has_res, "instruction has a destination but no result");
            r[COL_S_DST_START + I.dst] = 1; r[COL_DST] = res;
            regs[I.dst] = res;
        }
        if (I.op == OP_END) r[COL_IS_NEXT_LINE_SAME_TX] = 0;
        const bool multi_line = I.op == OP_TSTORE || I.op == OP_TLOAD || I.op == OP_SSTORE || I.op == OP_SLOAD;
        if (multi_line) r[COL_IS_NEXT_LINE_DIFF_INST] = 0;
        R.push_row(r);
        if (I.op == OP_END) break;
        if (multi_line) {
            // extension lines repeat the instruction's clk / pc / opcode / selectors / op0 / op1 and use the register-selector
            // columns as data carriers (cpu/tape.rs, cpu/storage.rs)
            std::vector<u64> e = r;
            for (u64 c = COL_S_OP0_START; c < COL_S_OP0_END; c++) e[c] = 0;
            for (u64 c = COL_S_OP1_START; c < COL_S_OP1_END; c++) e[c] = 0;
            for (u64 c = COL_S_DST_START; c < COL_S_DST_END; c++) e[c] = 0;
            e[COL_INST] = e[COL_IMM_VAL] = e[COL_FILTER_LOOKING_PROG_IMM] = e[COL_DST] = 0;
            e[COL_IS_EXT_LINE] = 1;
            if (I.op == OP_TSTORE || I.op == OP_TLOAD) {
                // executor/src/lib.rs:1687-1846: one extension line per word moved between memory and the tape
                u64 length, mem_base, tape_base;
                if (I.op == OP_TSTORE) { length = v1; mem_base = v0; tape_base = tp; }
                else {
                    need(v0 <= 1, "TLOAD flag must be 0 or 1");
                    if (v0) { length = v1; mem_base = regs[I.dst]; tape_base = tp - v1; } else { length = 1; mem_base = regs[I.dst]; tape_base = v1; }
                }
                need(length >= 1, "tape transfer of zero words");
                e[COL_FILTER_TAPE_LOOKING] = 1;
                for (u64 k = 0; k < length; k++) {
                    need(R.nrows() < max_steps, "program does not terminate");
                    const u64 maddr = mem_base + k, taddr = tape_base + k;
                    u64 word;
                    if (I.op == OP_TSTORE) {
                        word = mem_at(maddr, "tstore source was never written");
                        tape[taddr] = word;
                        R.mem.push_back({maddr, clk, M_TSTORE, word, 0});
                    } else {
                        auto tit = tape.find(taddr);
                        need(tit != tape.end(), "tload from a tape cell that was never written");
                        word = tit->second;
                        memory[maddr] = word;
                        R.mem.push_back({maddr, clk, M_TLOAD, word, 1});
                    }
                    R.tape.push_back({taddr, (u64)R.tape.size(), I.op, word});
                    e[COL_EXT_CNT] = k + 1;
                    e[COL_IS_NEXT_LINE_DIFF_INST] = k + 1 == length;
                    e[COL_AUX0] = maddr; e[COL_S_OP0_START] = taddr; e[COL_AUX1] = word;
                    R.push_row(e);
                }
                if (I.op == OP_TSTORE) tp += length;
            } else {
                // executor/src/lib.rs:1301-1545 + cpu/storage.rs: op0 / op1 = memory addresses of the 4-word slot key and value
                need(!imm && I.op0 >= 0 && I.op1 >= 0, "storage instructions take two registers");
                const MemOp mop = I.op == OP_SSTORE ? M_SSTORE : M_SLOAD;
                Hash4 key, value;
                for (int i = 0; i < 4; i++) {
                    key[i] = mem_at(addm(v0, i), "storage key was never written");
                    R.mem.push_back({addm(v0, i), clk, mop, key[i], 0});
                }
                const u64 kin[12] = {storage_addr[0], storage_addr[1], storage_addr[2], storage_addr[3], key[0], key[1], key[2], key[3], 0, 0, 0, 0};
                const u64 kf[4] = {0, 1, 0, 0};
                std::vector<u64> krow = poseidon_row(kin, kf);
                const Hash4 tree_key{krow[COL_POSEIDON_OUTPUT_RANGE_START], krow[COL_POSEIDON_OUTPUT_RANGE_START + 1],
                                     krow[COL_POSEIDON_OUTPUT_RANGE_START + 2], krow[COL_POSEIDON_OUTPUT_RANGE_START + 3]};
                std::vector<StorageRow> srows;
                std::vector<std::vector<u64>> prows;
                if (I.op == OP_SSTORE) {
                    for (int i = 0; i < 4; i++) {
                        value[i] = mem_at(addm(v1, i), "stored value was never written");
                        R.mem.push_back({addm(v1, i), clk, mop, value[i], 0});
                    }
                    tree.access(tree_key, &value, srows, prows);
                } else {
                    value = tree.access(tree_key, nullptr, srows, prows);
                    for (int i = 0; i < 4; i++) { memory[addm(v1, i)] = value[i]; R.mem.push_back({addm(v1, i), clk, mop, value[i], 1}); }
                }
                idx_storage += 1;
                R.storage.push_back(std::move(srows));
                R.storage_psdn.push_back(std::move(krow));
                for (auto& pr : prows) R.storage_psdn.push_back(std::move(pr));
                e[COL_EXT_CNT] = e[COL_IS_STORAGE_EXT_LINE] = e[COL_IS_NEXT_LINE_DIFF_INST] = 1;
                e[COL_IDX_STORAGE] = idx_storage;
                for (int i = 0; i < 4; i++) {
                    e[COL_S_OP0_START + i] = addm(v0, i); e[COL_S_OP0_START + 4 + i] = key[i];
                    e[COL_S_OP1_START + i] = addm(v1, i); e[COL_S_OP1_START + 4 + i] = value[i];
                    e[COL_S_DST_START + i] = tree_key[i];
                }
                R.push_row(e);
            }
        }
        pc = next_pc;
        clk += 1;
    }
}

// ---- tables ----------------------------------------------------------------------------------------------------------------------
void cpu_table(const Run& R, Table& t) {                                           // generation/cpu.rs:180-208 padding
    const size_t live = R.nrows();
    const size_t n = next_pow2(std::max<size_t>(live, 8));
    t.init(NUM_CPU_COLS, n);
    // row-major rows -> column-major table, a block of rows at a time so that both sides stay in cache
    const size_t B = 256;
    for (size_t i0 = 0; i0 < live; i0 += B) {
        const size_t i1 = std::min(live, i0 + B);
        for (size_t c = 0; c < NUM_CPU_COLS; c++) {
            u64* dst = &t.d[c * n];
            for (size_t i = i0; i < i1; i++) dst[i] = R.row(i)[c];
        }
    }
    const u64 pad_vals[][2] = {{COL_INST, 1048576}, {COL_OPCODE, 1ULL << OP_END}, {COL_S_END, 1}, {COL_IS_ENTRY_SC, 1}, {COL_IS_NEXT_LINE_DIFF_INST, 1},
                               {COL_IS_PADDING, 1}, {COL_IDX_STORAGE, R.row(live - 1)[COL_IDX_STORAGE]}};
    for (const auto& pv : pad_vals) std::fill(t.d.begin() + pv[0] * n + live, t.d.begin() + (pv[0] + 1) * n, pv[1]);
}

void program_table(const Run& R, const u64 code_addr[4], u64 beta, Table& t, std::vector<u64>& words) {
    words = R.words;
    while (words.size() % 8) words.push_back(0);                                   // prog_chunk hashes 8 words at a time
    const size_t n = next_pow2(std::max<size_t>(std::max(words.size(), R.executed.size()), 8));
    t.init(NUM_PROG_COLS, n);
    const u64 b = beta % P, b2 = mulm(b, b), b3 = mulm(b2, b), b4 = mulm(b3, b), b5 = mulm(b4, b);
    auto comp = [&](u64 pc, u64 w) {
        u64 acc = code_addr[0] % P;
        acc = addm(acc, mulm(code_addr[1], b)); acc = addm(acc, mulm(code_addr[2], b2)); acc = addm(acc, mulm(code_addr[3], b3));
        acc = addm(acc, mulm(pc % P, b4)); acc = addm(acc, mulm(w % P, b5));
        return acc;
    };
    for (size_t pc = 0; pc < words.size(); pc++) {
        for (int k = 0; k < 4; k++) t.at(COL_PROG_CODE_ADDR_RANGE_START + k, pc) = code_addr[k];
        t.at(COL_PROG_PC, pc) = pc; t.at(COL_PROG_INST, pc) = words[pc]; t.at(COL_PROG_COMP_PROG, pc) = comp(pc, words[pc]);
        t.at(COL_PROG_FILTER_PROG_CHUNK, pc) = 1;
    }
    for (size_t i = 0; i < n; i++) {
        const auto& e = i < R.executed.size() ? R.executed[i] : R.executed[0];      // filler rows repeat a listed word (filter 0)
        for (int k = 0; k < 4; k++) t.at(COL_PROG_EXEC_CODE_ADDR_RANGE_START + k, i) = code_addr[k];
        t.at(COL_PROG_EXEC_PC, i) = e.first; t.at(COL_PROG_EXEC_INST, i) = e.second; t.at(COL_PROG_EXEC_COMP_PROG, i) = comp(e.first, e.second);
        t.at(COL_PROG_FILTER_EXEC, i) = i < R.executed.size();
    }
    std::vector<u64> pi, pt;
    permuted_cols(t.column(COL_PROG_EXEC_COMP_PROG), t.column(COL_PROG_COMP_PROG), pi, pt);
    t.set_column(COL_PROG_EXEC_COMP_PROG_PERM, pi);
    t.set_column(COL_PROG_COMP_PROG_PERM, pt);
}

// program/prog_chunk_stark.rs + generation/prog.rs: one row per 8 program words, capacity chained; Poseidon rows of the
// chunk hashes followed by the builtin's rows, padded with the zero-input permutation
Hash4 program_hash(const std::vector<u64>& words) {
    // first four words of the chained chunk hash: what the state tree stores at the code address (prog_chunk_stark.rs:51-61)
    u64 cap[4] = {0, 0, 0, 0};
    const u64 f[4] = {0, 0, 0, 0};
    Hash4 h{0, 0, 0, 0};
    for (size_t i = 0; i + 8 <= words.size(); i += 8) {
        u64 in[12];
        for (int k = 0; k < 8; k++) in[k] = words[i + k];
        for (int k = 0; k < 4; k++) in[8 + k] = cap[k];
        const std::vector<u64> row = poseidon_row(in, f);
        for (int k = 0; k < 4; k++) { h[k] = row[COL_POSEIDON_OUTPUT_RANGE_START + k]; cap[k] = row[COL_POSEIDON_OUTPUT_RANGE_START + 8 + k]; }
    }
    return h;
}

void prog_chunk_and_poseidon(const u64 code_addr[4], const std::vector<u64>& words, const std::vector<std::vector<u64>>& extra_rows,
                             bool result_line, Table& chunk, Table& poseidon) {
    const size_t nchunks = words.size() / 8;
    const size_t n = next_pow2(std::max<size_t>(nchunks, 8));
    chunk.init(NUM_PROG_CHUNK_COLS, n);
    chunk.fill(COL_PROG_CHUNK_IS_PADDING_LINE, 1);
    std::vector<std::vector<u64>> prow;
    u64 cap[4] = {0, 0, 0, 0};
    const u64 filt[4] = {1, 0, 0, 0};
    for (size_t i = 0; i < nchunks; i++) {
        u64 in[12];
        for (int k = 0; k < 8; k++) in[k] = words[8 * i + k];
        for (int k = 0; k < 4; k++) in[8 + k] = cap[k];
        std::vector<u64> row = poseidon_row(in, filt);
        chunk.at(COL_PROG_CHUNK_IS_PADDING_LINE, i) = 0;
        for (int k = 0; k < 4; k++) chunk.at(COL_PROG_CHUNK_CODE_ADDR_RANGE_START + k, i) = code_addr[k];
        chunk.at(COL_PROG_CHUNK_START_PC, i) = 8 * i;
        for (int k = 0; k < 8; k++) chunk.at(COL_PROG_CHUNK_INST_RANGE_START + k, i) = words[8 * i + k];
        for (int k = 0; k < 4; k++) chunk.at(COL_PROG_CHUNK_CAP_RANGE_START + k, i) = cap[k];
        for (int k = 0; k < 12; k++) chunk.at(COL_PROG_CHUNK_HASH_RANGE_START + k, i) = row[COL_POSEIDON_OUTPUT_RANGE_START + k];
        chunk.at(COL_PROG_CHUNK_IS_FIRST_LINE, i) = i == 0;
        chunk.at(COL_PROG_CHUNK_IS_RESULT_LINE, i) = result_line && i + 1 == nchunks;
        for (u64 k = COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE_START; k < COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE_END; k++) chunk.at(k, i) = 1;
        for (int k = 0; k < 4; k++) cap[k] = row[COL_POSEIDON_OUTPUT_RANGE_START + 8 + k];
        prow.push_back(std::move(row));
    }
    for (const auto& r : extra_rows) prow.push_back(r);
    const size_t np = next_pow2(std::max<size_t>(prow.size(), 8));
    poseidon.init(NUM_POSEIDON_COLS, np);
    const u64 zero_in[12] = {0}, zero_f[4] = {0, 0, 0, 0};
    const std::vector<u64> zero_row = poseidon_row(zero_in, zero_f);               // generation/poseidon.rs: ZERO-hash padding rows
    for (size_t i = 0; i < np; i++) {
        const std::vector<u64>& r = i < prow.size() ? prow[i] : zero_row;
        for (size_t c = 0; c < NUM_POSEIDON_COLS; c++) poseidon.at(c, i) = r[c];
    }
}

u64 mem_selector(MemOp op) {
    switch (op) {
        case M_CALL: return COL_MEM_S_CALL;
        case M_MLOAD: return COL_MEM_S_MLOAD;
        case M_MSTORE: return COL_MEM_S_MSTORE;
        case M_POSEIDON: return COL_MEM_S_POSEIDON;
        case M_SLOAD: return COL_MEM_S_SLOAD;
        case M_SSTORE: return COL_MEM_S_SSTORE;
        case M_TLOAD: return COL_MEM_S_TLOAD;
        case M_TSTORE: return COL_MEM_S_TSTORE;
        default: return COL_MEM_S_RET;
    }
}
uint32_t mem_opcode(MemOp op) {
    switch (op) {
        case M_CALL: return OP_CALL;
        case M_MLOAD: return OP_MLOAD;
        case M_MSTORE: return OP_MSTORE;
        case M_POSEIDON: return OP_POSEIDON;
        case M_SLOAD: return OP_SLOAD;
        case M_SSTORE: return OP_SSTORE;
        case M_TLOAD: return OP_TLOAD;
        case M_TSTORE: return OP_TSTORE;
        default: return OP_RET;
    }
}

// generation/memory.rs:5-95: cells sorted by (address, clock); stack and heap regions live, prophet-region padding after them
void memory_table(std::vector<MemCell> cells, Table& t, std::vector<u64>& rc_vals, std::vector<u64>& cond_vals, bool quirks) {
    std::sort(cells.begin(), cells.end(), [](const MemCell& a, const MemCell& b) {
        if (a.addr != b.addr) return a.addr < b.addr;
        if (a.clk != b.clk) return a.clk < b.clk;
        if (a.op != b.op) return a.op < b.op;
        if (a.value != b.value) return a.value < b.value;
        return a.is_write < b.is_write;
    });
    const size_t n = next_pow2(std::max<size_t>(cells.size() + 1, 8));
    t.init(NUM_MEM_COLS, n);
    const u64 span = 0xFFFFFFFFULL;
    bool have_prev = false, prev_heap = false;
    u64 prev_addr = 0, prev_clk = 0;
    for (size_t i = 0; i < cells.size(); i++) {
        const MemCell& c = cells[i];
        t.at(COL_MEM_IS_RW, i) = 1;
        t.at(COL_MEM_ADDR, i) = c.addr; t.at(COL_MEM_CLK, i) = c.clk; t.at(COL_MEM_OP, i) = 1ULL << mem_opcode(c.op); t.at(COL_MEM_VALUE, i) = c.value;
        t.at(mem_selector(c.op), i) = 1;
        t.at(COL_MEM_IS_WRITE, i) = c.is_write;
        const bool heap = c.addr >= ADDR_HEAP_PTR;
        if (heap) {
            const u64 cond = subm(subm(0, span), c.addr);
            t.at(COL_MEM_REGION_HEAP, i) = 1; t.at(COL_MEM_DIFF_ADDR_COND, i) = cond; t.at(COL_MEM_FILTER_LOOKING_RC_COND, i) = 1;
            cond_vals.push_back(cond);
        }
        if (have_prev && heap && !prev_heap) {
            // first heap row: the address gap to the stack region is not range-checked (generation/memory.rs:77-86)
            t.at(COL_MEM_DIFF_ADDR, i) = c.addr - prev_addr; t.at(COL_MEM_DIFF_ADDR_INV, i) = invm(c.addr - prev_addr);
        } else if (have_prev) {
            const bool same = c.addr == prev_addr;
            const u64 d_addr = c.addr - prev_addr;
            t.at(COL_MEM_DIFF_ADDR, i) = d_addr; t.at(COL_MEM_DIFF_ADDR_INV, i) = invm(d_addr);
            t.at(COL_MEM_DIFF_CLK, i) = same ? c.clk - prev_clk : 0;
            t.at(COL_MEM_RW_ADDR_UNCHANGED, i) = same;
            const u64 rc = same ? c.clk - prev_clk : d_addr;
            t.at(COL_MEM_RC_VALUE, i) = rc; t.at(COL_MEM_FILTER_LOOKING_RC, i) = 1;
            rc_vals.push_back(rc);
        }
        have_prev = true; prev_addr = c.addr; prev_clk = c.clk; prev_heap = heap;
    }
    u64 a = subm(0, span);
    if (quirks && cells.empty()) {       // OLA_TRACEGEN_REFERENCE_QUIRKS: generation/memory.rs:95-153 as it is
        for (size_t i = 0; i < n; i++) {
            t.at(COL_MEM_ADDR, i) = a; t.at(COL_MEM_IS_WRITE, i) = t.at(COL_MEM_REGION_PROPHET, i) = 1;
            t.at(COL_MEM_DIFF_ADDR_COND, i) = t.at(COL_MEM_RC_VALUE, i) = subm(0, a);
            if (i) t.at(COL_MEM_S_PROPHET, i) = t.at(COL_MEM_DIFF_ADDR, i) = t.at(COL_MEM_DIFF_ADDR_INV, i) = 1;
            a = addm(a, 1);
        }
        return;
    }
    const u64 last_addr = have_prev ? prev_addr : 0;
    const size_t start = cells.empty() ? 1 : cells.size();
    if (cells.empty()) { t.at(COL_MEM_S_PROPHET, 0) = 1; t.at(COL_MEM_IS_WRITE, 0) = 1; }
    for (size_t i = start; i < n; i++) {
        t.at(COL_MEM_S_PROPHET, i) = t.at(COL_MEM_IS_WRITE, i) = t.at(COL_MEM_REGION_PROPHET, i) = 1;
        t.at(COL_MEM_ADDR, i) = a;
        const u64 d = i == start ? subm(a, last_addr) : 1;
        t.at(COL_MEM_DIFF_ADDR, i) = d; t.at(COL_MEM_DIFF_ADDR_INV, i) = invm(d);
        t.at(COL_MEM_DIFF_ADDR_COND, i) = t.at(COL_MEM_RC_VALUE, i) = subm(0, a);
        a = addm(a, 1);
    }
}

// generation/poseidon_chunk.rs; builtins/poseidon/poseidon_chunk_stark.rs:98-284
void poseidon_chunk_table(const std::vector<PsdnCall>& calls, Table& t, std::vector<std::vector<u64>>& prow) {
    std::vector<std::map<u64, u64>> rows;
    for (const PsdnCall& c : calls) {
        std::map<u64, u64> base{{COL_POSEIDON_CHUNK_CLK, c.clk}, {COL_POSEIDON_CHUNK_OPCODE, 1ULL << OP_POSEIDON}, {COL_POSEIDON_CHUNK_OP1, c.len},
                                {COL_POSEIDON_CHUNK_DST, c.dst}};
        std::map<u64, u64> head = base;
        head[COL_POSEIDON_CHUNK_OP0] = c.src;
        head[COL_POSEIDON_CHUNK_FILTER_LOOKED_CPU] = 1;
        rows.push_back(head);
        for (size_t j = 0; j < c.chunks.size(); j++) {
            const PsdnChunk& ch = c.chunks[j];
            std::map<u64, u64> r = base;
            r[COL_POSEIDON_CHUNK_OP0] = ch.addr;
            r[COL_POSEIDON_CHUNK_ACC_CNT] = 8 * (j + 1);
            for (int k = 0; k < 8; k++) r[COL_POSEIDON_CHUNK_VALUE_RANGE_START + k] = ch.vals[k];
            for (int k = 0; k < 4; k++) r[COL_POSEIDON_CHUNK_CAP_RANGE_START + k] = ch.cap[k];
            for (int k = 0; k < 12; k++) r[COL_POSEIDON_CHUNK_HASH_RANGE_START + k] = ch.row[COL_POSEIDON_OUTPUT_RANGE_START + k];
            r[COL_POSEIDON_CHUNK_IS_EXT_LINE] = 1;
            r[COL_POSEIDON_CHUNK_IS_RESULT_LINE] = j + 1 == c.chunks.size();
            for (u64 k = COL_POSEIDON_CHUNK_FILTER_LOOKING_MEM_RANGE_START; k < COL_POSEIDON_CHUNK_FILTER_LOOKING_MEM_RANGE_END; k++) r[k] = 1;
            r[COL_POSEIDON_CHUNK_FILTER_LOOKING_POSEIDON] = 1;
            rows.push_back(r);
            prow.push_back(ch.row);
        }
    }
    const size_t n = next_pow2(std::max<size_t>(rows.size(), 8));
    t.init(NUM_POSEIDON_CHUNK_COLS, n);
    t.fill(COL_POSEIDON_CHUNK_IS_PADDING_LINE, 1);
    for (size_t i = 0; i < rows.size(); i++) {
        t.at(COL_POSEIDON_CHUNK_IS_PADDING_LINE, i) = 0;
        for (const auto& kv : rows[i]) t.at(kv.first, i) = kv.second;
    }
}

// generation/storage.rs:7-123: 256 rows per proof, the CPU's accesses in execution order, then the program-hash reads
void storage_table(const std::vector<std::vector<StorageRow>>& accesses, const std::vector<std::vector<StorageRow>>& prog_reads, Table& t) {
    size_t total = 0;
    for (const auto& a : accesses) total += a.size();
    for (const auto& a : prog_reads) total += a.size();
    const size_t n = next_pow2(std::max<size_t>(total, 8));
    t.init(NUM_COL_ST, n);
    t.fill(COL_ST_IS_PADDING, 1);
    size_t i = 0;
    u64 acc = 0;
    const Hash4* last_root = nullptr;
    auto emit = [&](const std::vector<StorageRow>& rows, u64 idx, bool for_prog) {
        for (const StorageRow& r : rows) {
            acc = r.layer % 64 == 1 ? (u64)r.bit : addm(addm(acc, acc), (u64)r.bit);
            t.at(COL_ST_IS_PADDING, i) = 0;
            t.at(COL_ST_ACCESS_IDX, i) = idx; t.at(COL_ST_IS_WRITE, i) = r.is_write; t.at(COL_ST_LAYER, i) = r.layer; t.at(COL_ST_LAYER_BIT, i) = r.bit;
            t.at(COL_ST_ADDR_ACC, i) = acc; t.at(COL_ST_HASH_TYPE, i) = r.layer == 256;
            const Hash4* src[8] = {&r.pre_root, &r.root, &r.addr, &r.pre_path, &r.path, &r.sib, &r.pre_hash, &r.hash};
            const u64 dst[8] = {COL_ST_PRE_ROOT_RANGE_START, COL_ST_ROOT_RANGE_START, COL_ST_ADDR_RANGE_START, COL_ST_PRE_PATH_RANGE_START,
                                COL_ST_PATH_RANGE_START, COL_ST_SIB_RANGE_START, COL_ST_PRE_HASH_RANGE_START, COL_ST_HASH_RANGE_START};
            for (int g = 0; g < 8; g++) for (int k = 0; k < 4; k++) t.at(dst[g] + k, i) = (*src[g])[k];
            t.at(COL_ST_IS_LAYER_1, i) = r.layer == 1; t.at(COL_ST_IS_LAYER_64, i) = r.layer == 64; t.at(COL_ST_IS_LAYER_128, i) = r.layer == 128;
            t.at(COL_ST_IS_LAYER_192, i) = r.layer == 192; t.at(COL_ST_IS_LAYER_256, i) = r.layer == 256;
            t.at(COL_ST_ACC_LAYER_MARKER, i) = 1 + r.layer / 64;
            t.at(COL_ST_FILTER_IS_HASH_BIT_0, i) = 1 - r.bit; t.at(COL_ST_FILTER_IS_HASH_BIT_1, i) = r.bit;
            t.at(COL_ST_FILTER_IS_FOR_PROG, i) = for_prog && r.layer == 256;
            last_root = &r.root;
            i++;
        }
    };
    for (size_t a = 0; a < accesses.size(); a++) emit(accesses[a], a + 1, false);
    for (size_t a = 0; a < prog_reads.size(); a++) emit(prog_reads[a], accesses.size() + a + 1, true);
    if (last_root)
        for (int k = 0; k < 4; k++) std::fill(t.d.begin() + (COL_ST_ROOT_RANGE_START + k) * n + i, t.d.begin() + (COL_ST_ROOT_RANGE_START + k + 1) * n, (*last_root)[k]);
}

// builtins/tape/tape_stark.rs:44-143: cells sorted by tape address, the write first, then its reads; padding repeats the last
// cell as an unfiltered TLOAD
void tape_table(std::vector<TapeCell> cells, Table& t) {
    std::sort(cells.begin(), cells.end(), [](const TapeCell& a, const TapeCell& b) { return a.addr != b.addr ? a.addr < b.addr : a.seq < b.seq; });
    const size_t n = next_pow2(std::max<size_t>(cells.size(), 8));
    t.init(NUM_COL_TAPE, n);
    if (cells.empty()) { t.fill(COL_TAPE_OPCODE, 1ULL << OP_TLOAD); return; }
    for (size_t i = 0; i < n; i++) {
        const TapeCell& c = cells[std::min(i, cells.size() - 1)];
        const bool live = i < cells.size();
        t.at(COL_TAPE_OPCODE, i) = 1ULL << (live ? c.op : OP_TLOAD);
        t.at(COL_TAPE_ADDR, i) = c.addr; t.at(COL_TAPE_VALUE, i) = c.word; t.at(COL_TAPE_FILTER_LOOKED, i) = live;
    }
}

// generation/builtin.rs:208-247
void cmp_table(const std::vector<std::pair<u64, u64>>& ops, Table& t, std::vector<u64>& abs_diffs) {
    const size_t n = next_pow2(ops.size());
    t.init(COL_NUM_CMP, n);
    for (size_t i = 0; i < ops.size(); i++) {
        const u64 a = ops[i].first, b = ops[i].second, d = a >= b ? a - b : b - a;
        t.at(COL_CMP_OP0, i) = a; t.at(COL_CMP_OP1, i) = b; t.at(COL_CMP_GTE, i) = a >= b; t.at(COL_CMP_ABS_DIFF, i) = d;
        t.at(COL_CMP_ABS_DIFF_INV, i) = invm(d); t.at(COL_CMP_FILTER_LOOKING_RC, i) = 1;
        abs_diffs.push_back(d);
    }
    for (size_t i = ops.size(); i < n; i++)
        t.at(COL_CMP_OP0, i) = t.at(COL_CMP_GTE, i) = t.at(COL_CMP_ABS_DIFF, i) = t.at(COL_CMP_ABS_DIFF_INV, i) = 1;
}

// generation/builtin.rs:249-316: value, filters (cpu, memory sort, memory region, cmp), 16-bit limbs, fixed table, permuted columns
struct RcRow { u64 v; int f[4]; };
void rc_table(const std::vector<RcRow>& rows, uint32_t range_bits, Table& t) {
    const size_t size = (size_t)1 << range_bits;
    const size_t n = next_pow2(std::max(rows.size(), size));
    t.init(COL_NUM_RC, n);
    for (size_t i = 0; i < rows.size(); i++) {
        need(rows[i].v < (u64)size * size, "range-checked value does not fit two limbs");
        t.at(RC_CPU_FILTER, i) = rows[i].f[0]; t.at(RC_MEMORY_SORT_FILTER, i) = rows[i].f[1]; t.at(RC_MEMORY_REGION_FILTER, i) = rows[i].f[2];
        t.at(RC_CMP_FILTER, i) = rows[i].f[3];
        t.at(RC_VAL, i) = rows[i].v; t.at(RC_LIMB_LO, i) = rows[i].v % size; t.at(RC_LIMB_HI, i) = rows[i].v / size;
    }
    std::vector<u64> fix(n);
    for (size_t i = 0; i < n; i++) fix[i] = i < size ? i : size - 1;
    t.set_column(RC_FIX_RANGE_CHECK_U16, fix);
    std::vector<u64> pi, pt;
    permuted_cols(t.column(RC_LIMB_LO), fix, pi, pt);
    t.set_column(RC_LIMB_LO_PERMUTED, pi); t.set_column(RC_FIX_RANGE_CHECK_U16_PERMUTED_LO, pt);
    permuted_cols(t.column(RC_LIMB_HI), fix, pi, pt);
    t.set_column(RC_LIMB_HI_PERMUTED, pi); t.set_column(RC_FIX_RANGE_CHECK_U16_PERMUTED_HI, pt);
}

// generation/builtin.rs:35-205 with limb_bits-wide limbs
// `derive`: the compress challenge is drawn from a transcript that has observed the twelve limb columns (OP0, OP1, RES limbs
// over the whole padded height), as generation/builtin.rs:120-131 does; otherwise the caller's `beta` is used (tests only)
u64 bitwise_table(u64 beta, bool derive, uint32_t limb_bits, const std::vector<BwOp>& ops, Table& t, bool quirks) {
    const size_t size = (size_t)1 << limb_bits, per = size * size;
    const size_t n = next_pow2(std::max(std::max(size, 3 * per), ops.size()));
    t.init(COL_NUM_BITWISE, n);
    size_t index = 0;
    for (size_t a = 0; a < size; a++) {
        t.at(BW_FIX_RANGE_CHECK_U8, a) = a;
        for (size_t b = 0; b < size; b++) {
            const u64 res[3] = {a & b, a | b, a ^ b};
            const uint32_t tag[3] = {OP_AND, OP_OR, OP_XOR};
            for (int k = 0; k < 3; k++) {
                const size_t r = k * per + index;
                t.at(BW_FIX_BITWSIE_OP0, r) = a; t.at(BW_FIX_BITWSIE_OP1, r) = b; t.at(BW_FIX_BITWSIE_RES, r) = res[k]; t.at(BW_FIX_TAG, r) = 1ULL << tag[k];
            }
            index++;
        }
    }
    for (size_t r = 0; r < ops.size(); r++) {
        const u64 x = ops[r].a, y = ops[r].b;
        need(limb_bits >= 16 || (x < ((u64)1 << (4 * limb_bits)) && y < ((u64)1 << (4 * limb_bits))), "bitwise operand does not fit four limbs");
        const u64 z = ops[r].op == OP_AND ? (x & y) : ops[r].op == OP_OR ? (x | y) : (x ^ y);
        const u64 tag = 1ULL << ops[r].op;
        t.at(BW_TAG, r) = tag; t.at(BW_OP0, r) = x; t.at(BW_OP1, r) = y; t.at(BW_RES, r) = z;
        t.at(BW_FILTER, r) = 1;                                                     // looked up by the CPU's AND / OR / XOR rows
        for (int i = 0; i < (quirks ? 3 : 4); i++) {      // OLA_TRACEGEN_REFERENCE_QUIRKS: the fourth limb is lost (generation/builtin.rs:66,71,76)
            const u64 lx = (x >> (limb_bits * i)) & (size - 1), ly = (y >> (limb_bits * i)) & (size - 1), lz = (z >> (limb_bits * i)) & (size - 1);
            t.at(BW_OP0_LIMBS_START + i, r) = lx; t.at(BW_OP1_LIMBS_START + i, r) = ly; t.at(BW_RES_LIMBS_START + i, r) = lz;
        }
    }
    if (derive) {
        HostChallenger ch;
        const u64 starts[3] = {BW_OP0_LIMBS_START, BW_OP1_LIMBS_START, BW_RES_LIMBS_START};
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < 4; i++)
                for (size_t r = 0; r < n; r++) ch.observe(t.at(starts[k] + i, r));
        beta = ch.get();
    }
    const u64 b1 = beta % P, b2 = mulm(b1, b1), b3 = mulm(b2, b1);
    auto compress = [&](u64 tag, u64 x, u64 y, u64 z) { return addm(addm(tag % P, mulm(x, b1)), addm(mulm(y, b2), mulm(z, b3))); };
    std::vector<u64> fix(n);
    for (size_t i = 0; i < n; i++) fix[i] = compress(t.at(BW_FIX_TAG, i), t.at(BW_FIX_BITWSIE_OP0, i), t.at(BW_FIX_BITWSIE_OP1, i), t.at(BW_FIX_BITWSIE_RES, i));
    t.set_column(BW_FIX_COMPRESS, fix);
    for (size_t r = 0; r < ops.size(); r++)
        for (int i = 0; i < 4; i++)
            t.at(BW_COMPRESS_LIMBS_START + i, r) = compress(t.at(BW_TAG, r), t.at(BW_OP0_LIMBS_START + i, r), t.at(BW_OP1_LIMBS_START + i, r), t.at(BW_RES_LIMBS_START + i, r));
    const std::vector<u64> rc8 = t.column(BW_FIX_RANGE_CHECK_U8);
    std::vector<u64> pi, pt;
    for (int i = 0; i < 4; i++) {
        const u64 src[3] = {BW_OP0_LIMBS_START, BW_OP1_LIMBS_START, BW_RES_LIMBS_START};
        const u64 dst[3] = {BW_OP0_LIMBS_PERMUTED_START, BW_OP1_LIMBS_PERMUTED_START, BW_RES_LIMBS_PERMUTED_START};
        for (int k = 0; k < 3; k++) {
            permuted_cols(t.column(src[k] + i), rc8, pi, pt);
            t.set_column(dst[k] + i, pi);
            t.set_column(BW_FIX_RANGE_CHECK_U8_PERMUTED_START + 4 * k + i, pt);
        }
        permuted_cols(t.column(BW_COMPRESS_LIMBS_START + i), fix, pi, pt);
        t.set_column(BW_COMPRESS_PERMUTED_START + i, pi);
        t.set_column(BW_FIX_COMPRESS_PERMUTED_START + i, pt);
    }
    return beta % P;
}

void flag_padding(Table& t, size_t ncols, size_t n, size_t flag_col) { t.init(ncols, n); t.fill(flag_col, 1); }

thread_local std::string g_err;

}  // namespace

struct OlaTraceSet {
    std::array<Table, 12> tables;
    uint64_t cpu_rows = 0;
    uint64_t bitwise_beta = 0, program_beta = 0;   // the compress challenges the tables were built with
};

extern "C" {

const char* ola_tracegen_last_error(void) { return g_err.c_str(); }

int32_t ola_tracegen_run(const OlaInstr* program, size_t n_instr, const uint64_t code_addr[4], const uint64_t storage_addr[4],
                         uint32_t range_bits, uint32_t limb_bits, uint64_t bitwise_beta, uint64_t program_beta, uint64_t max_steps,
                         uint32_t flags, OlaTraceSet** out) {
    try {
        need(program && n_instr && code_addr && storage_addr && out, "null argument");
        need(range_bits >= 1 && range_bits <= 16 && limb_bits >= 1 && limb_bits <= 8, "range_bits / limb_bits out of range");
        std::unique_ptr<OlaTraceSet> set(new OlaTraceSet());
        const bool prove_program_hash = flags & OLA_TRACEGEN_PROVE_PROGRAM_HASH;
        const bool explicit_betas = flags & OLA_TRACEGEN_EXPLICIT_BETAS;
        const bool quirks = flags & OLA_TRACEGEN_REFERENCE_QUIRKS;
        Run R;
        StorageTree tree;
        const Hash4 code_key{code_addr[0], code_addr[1], code_addr[2], code_addr[3]};
        if (prove_program_hash) {
            std::vector<u64> listing;
            std::vector<size_t> pcs;
            program_words(program, n_instr, listing, pcs);
            while (listing.size() % 8) listing.push_back(0);
            tree.write(StorageTree::key_of(code_key), program_hash(listing));
        }
        const Hash4 start_root = tree.root();
        execute(program, n_instr, code_addr, storage_addr, max_steps, tree, R);
        set->cpu_rows = R.nrows();
        if (!explicit_betas) {
            // generation/prog.rs:23-29: the transcript observes the state roots before and after the run, limb by limb
            const Hash4 end_root = tree.root();
            HostChallenger ch;
            for (int i = 0; i < 4; i++) { ch.observe(start_root[i]); ch.observe(end_root[i]); }
            program_beta = ch.get();
        }
        set->program_beta = program_beta % P;
        auto& T = set->tables;
        cpu_table(R, T[CPU]);
        std::vector<u64> words;
        program_table(R, code_addr, program_beta, T[PROGRAM], words);
        std::vector<std::vector<u64>> builtin_rows;
        poseidon_chunk_table(R.psdn, T[POSEIDON_CHUNK], builtin_rows);
        std::vector<std::vector<StorageRow>> prog_reads;
        if (prove_program_hash) {
            prog_reads.emplace_back();
            std::vector<std::vector<u64>> prows;
            tree.access(code_key, nullptr, prog_reads.back(), prows);
            for (auto& pr : prows) R.storage_psdn.push_back(std::move(pr));
        }
        for (auto& pr : R.storage_psdn) builtin_rows.push_back(std::move(pr));
        prog_chunk_and_poseidon(code_addr, words, builtin_rows, prove_program_hash, T[PROG_CHUNK], T[POSEIDON]);
        std::vector<u64> abs_diffs, mem_rc, mem_cond;
        cmp_table(R.cmp, T[CMP], abs_diffs);
        memory_table(R.mem, T[MEMORY], mem_rc, mem_cond, quirks);
        std::vector<RcRow> rc;
        for (u64 v : R.rc) rc.push_back({v, {1, 0, 0, 0}});
        for (u64 v : abs_diffs) rc.push_back({v, {0, 0, 0, 1}});
        for (u64 v : mem_rc) rc.push_back({v, {0, 1, 0, 0}});
        for (u64 v : mem_cond) rc.push_back({v, {0, 0, 1, 0}});
        rc_table(rc, range_bits, T[RANGECHECK]);
        set->bitwise_beta = bitwise_table(bitwise_beta, !explicit_betas, limb_bits, R.bitwise, T[BITWISE], quirks);
        storage_table(R.storage, prog_reads, T[STORAGE_ACCESS]);
        tape_table(R.tape, T[TAPE]);
        flag_padding(T[SCCALL], NUM_COL_SCCALL, 8, COL_SCCALL_IS_PADDING);
        *out = set.release();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

int32_t ola_tracegen_table(const OlaTraceSet* set, uint32_t table, uint32_t* ncols, uint32_t* log_n, const uint64_t** data) {
    if (!set || table >= 12 || !ncols || !log_n || !data) { g_err = "invalid argument"; return -1; }
    const Table& t = set->tables[table];
    uint32_t l = 0;
    while (((size_t)1 << l) < t.n) l++;
    *ncols = (uint32_t)t.ncols; *log_n = l; *data = t.d.data();
    return 0;
}

uint64_t ola_tracegen_cpu_rows(const OlaTraceSet* set) { return set ? set->cpu_rows : 0; }

int32_t ola_tracegen_betas(const OlaTraceSet* set, uint64_t out[2]) {
    if (!set || !out) { g_err = "invalid argument"; return -1; }
    out[0] = set->bitwise_beta; out[1] = set->program_beta;
    return 0;
}

void ola_tracegen_free(OlaTraceSet* set) { delete set; }

}  // extern "C"
