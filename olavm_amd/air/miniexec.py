"""A miniature executor for OlaVM's register instructions (SURVEY f-1): runs a small program and fills the STARK tables it
touches so that EVERY constraint and EVERY cross-table lookup holds -- a real (if tiny) execution trace without the Rust
executor.  Supported: MOV, NOT, ADD, MUL, EQ, NEQ, ASSERT, JMP, CJMP, RC, AND / OR / XOR, GTE, MSTORE / MLOAD ([reg + imm], stack and
heap regions), CALL / RET, TSTORE / TLOAD and SSTORE / SLOAD (with their CPU extension lines), POSEIDON (whole 8-word
blocks), END, with register or immediate second operands.  Tables that receive live rows: CPU, memory, program,
prog_chunk (program hashing), poseidon_chunk (the builtin), Poseidon (chunk, builtin, tree-key and state-tree hashes),
storage_access (256-level proofs), tape, range-check (RC, GTE, the memory table's sort and region columns), bitwise, cmp;
only the cross-contract-call table keeps its padding rows (olavm_amd/air/tracegen.py) -- SCCALL is not executed because
the reference constraints contradict each other on it: cpu_stark.rs:432-437 makes the callee's env_idx the caller's
call_sc_cnt (0 at a transaction's first call), the cpu<->sccall lookup (:238-242) makes it env_idx + 1.

Restated from (reference paths): core/src/vm/opcodes.rs (opcode bit masks), circuits/src/cpu/cpu_stark.rs:529-581
(instruction word = op1_imm * 2^62 + one-hot register selectors at bits 52+r / 42+r / 32+r + opcode mask),
circuits/src/generation/{cpu,memory,prog,builtin,poseidon,storage}.rs (row layouts and padding), circuits/src/program/*.rs
(program / prog_chunk tables: 8 instruction words per Poseidon-hashed chunk, capacity chained from the previous chunk's
hash).  By default the program-hash chain stops short of a result line (prog_chunk_stark.rs never forces one);
instance(prove_program_hash=True) closes it with the state-tree proof of the code-address leaf.  Validity is not taken on
trust: tests check every table with the oracle's `check_constraints` and the whole proof -- including the cross-table
products -- with its verifier."""
import numpy as np

from . import ola_tables as T
from . import tracegen as TG
from .dsl import P
from . import poseidon_params as PP

REG = 10


def _inv(x):
    return pow(x % P, P - 2, P) if x % P else 0


# ---- Poseidon with the S-box inputs recorded (the 134-column row of the Poseidon table), plain Python
def _mds(s):
    out = []
    for r in range(12):
        v = sum(s[(i + r) % 12] * PP.MDS_CIRC[i] for i in range(12)) + s[r] * PP.MDS_DIAG[r]
        out.append(v % P)
    return out


def _sbox(x):
    return pow(x, 7, P)


def poseidon_row(inp, filters=(0, 0, 0, 0)):
    row = [0] * T.NUM_POSEIDON_COLS
    row[0:4] = list(filters)
    s = [int(x) % P for x in inp]
    row[4:16] = s
    for r in range(30):
        full = r < 4 or r >= 26
        s = [(x + PP.RC[12 * r + i]) % P for i, x in enumerate(s)]
        if full:
            if 1 <= r <= 3:
                row[28 + 12 * (r - 1):28 + 12 * r] = s
            elif r >= 26:
                row[86 + 12 * (r - 26):86 + 12 * (r - 25)] = s
            s = [_sbox(x) for x in s]
        else:
            row[64 + r - 4] = s[0]
            s[0] = _sbox(s[0])
        s = _mds(s)
    row[16:28] = s
    return row


# ---- the account-storage tree (a 256-level sparse Merkle tree over Poseidon)
class StorageTree:
    """State tree behind SSTORE / SLOAD and the program-hash read (builtins/storage/storage_access_stark.rs:110-334): 256
    levels below the root, the key's bits (four 64-bit limbs, most significant bit first) choose the child at every level,
    inner nodes are Poseidon(left || right || [0,0,0,0])[:4] and the lowest level hashes the two 4-word VALUES of a sibling
    pair with the capacity word set to 1.  Untouched leaves hold [0,0,0,0].  `access` returns the 256 table rows of one
    proof (root side first) together with the Poseidon-table rows -- old and new hash of every level -- they look up."""
    DEPTH = 256

    def __init__(self):
        self.nodes = {}
        self.default = [None] * (self.DEPTH + 1)
        self.default[self.DEPTH] = (0, 0, 0, 0)
        for d in range(self.DEPTH - 1, -1, -1):
            c = self.default[d + 1]
            self.default[d] = self.hash(c, c, d == self.DEPTH - 1)[0]

    @staticmethod
    def hash(left, right, leaf_level):
        row = poseidon_row(list(left) + list(right) + [int(leaf_level), 0, 0, 0], filters=(0, 0, int(leaf_level), 1 - int(leaf_level)))
        return tuple(row[16:20]), row

    @staticmethod
    def key_bits(addr):
        k = 0
        for limb in addr:
            k = (k << 64) | (int(limb) % P)
        return k

    def node(self, depth, prefix):
        return self.nodes.get((depth, prefix), self.default[depth])

    def root(self):
        return self.node(0, 0)

    def _write(self, k, value):
        """Sets leaf `k`; -> per level (depth DEPTH-1 .. 0) the new hash and its Poseidon row."""
        self.nodes[(self.DEPTH, k)] = tuple(value)
        out = []
        for d in range(self.DEPTH - 1, -1, -1):
            pre = k >> (self.DEPTH - d)
            h, row = self.hash(self.node(d + 1, 2 * pre), self.node(d + 1, 2 * pre + 1), d == self.DEPTH - 1)
            self.nodes[(d, pre)] = h
            out.append((h, row))
        return out[::-1]                      # index = depth of the hashed node

    def set(self, addr, value):
        """State present before the execution starts (no proof rows)."""
        self._write(self.key_bits(addr), value)

    def get(self, addr):
        return self.node(self.DEPTH, self.key_bits(addr))

    def access(self, addr, value=None):
        """Read (value None) or write of leaf `addr` -> (rows, Poseidon rows, value read / written).  rows[l-1] is layer l:
        dict(bit, sib, pre_path, path, pre_hash, hash); pre_* describe the tree before a write (equal to the others on a
        read).  Every layer contributes its old-tree and its new-tree hash to the Poseidon table."""
        k = self.key_bits(addr)
        pre = []
        for layer in range(1, self.DEPTH + 1):
            prefix = k >> (self.DEPTH - layer)
            pre.append((prefix & 1, self.node(layer, prefix ^ 1), self.node(layer, prefix), self.node(layer - 1, prefix >> 1)))
        pre_root = self.root()
        is_write = value is not None
        if is_write:
            self._write(k, value)
        rows, prows = [], []
        for layer in range(1, self.DEPTH + 1):
            bit, sib, pre_path, pre_hash = pre[layer - 1]
            prefix = k >> (self.DEPTH - layer)
            path, hsh = self.node(layer, prefix), self.node(layer - 1, prefix >> 1)
            rows.append(dict(layer=layer, bit=bit, sib=sib, pre_path=pre_path, path=path, pre_hash=pre_hash, hash=hsh,
                             pre_root=pre_root, root=self.root(), is_write=int(is_write), addr=tuple(int(a) % P for a in addr)))
            for child, expect in ((path, hsh), (pre_path, pre_hash)):
                pair = (sib, child) if bit else (child, sib)
                h, prow = self.hash(pair[0], pair[1], layer == self.DEPTH)
                assert h == expect
                prows.append(prow)
        return rows, prows, self.node(self.DEPTH, k)


# ---- program
class Program:
    """Instructions: (op, dst, op0, op1) with register indices or None; op1 may be ('imm', value)."""

    def __init__(self, code_addr=(11, 22, 33, 44), storage_addr=(55, 66, 77, 88)):
        self.ins, self.code_addr, self.storage_addr = [], tuple(code_addr), tuple(storage_addr)

    def add(self, op, dst=None, op0=None, op1=None):
        self.ins.append((op, dst, op0, op1))
        return self

    def words(self):
        """-> (list of program words, pc of every instruction)."""
        words, pcs = [], []
        for op, dst, op0, op1 in self.ins:
            imm = isinstance(op1, tuple)
            w = (1 << 62) if imm else 0
            if op0 is not None:
                w += 1 << (52 + op0)
            if op1 is not None and not imm:
                w += 1 << (42 + op1)
            if dst is not None:
                w += 1 << (32 + dst)
            w += T.op_mask(op)
            pcs.append(len(words))
            words.append(w)
            if imm:
                words.append(int(op1[1]) % P)
        return words, pcs


SELECTOR_MEM = {"MLOAD": T.COL_S_MLOAD, "MSTORE": T.COL_S_MSTORE}
SELECTOR = {"NOT": T.COL_S_NOT, "ASSERT": T.COL_S_SIMPLE_ARITHMATIC_OP, "SSTORE": T.COL_S_SSTORE, "SLOAD": T.COL_S_SLOAD, "TSTORE": T.COL_S_TSTORE, "TLOAD": T.COL_S_TLOAD, "CALL": T.COL_S_CALL, "RET": T.COL_S_RET, "POSEIDON": T.COL_S_PSDN, "MLOAD": T.COL_S_MLOAD, "MSTORE": T.COL_S_MSTORE, "ADD": T.COL_S_SIMPLE_ARITHMATIC_OP, "MUL": T.COL_S_SIMPLE_ARITHMATIC_OP, "EQ": T.COL_S_SIMPLE_ARITHMATIC_OP,
            "NEQ": T.COL_S_SIMPLE_ARITHMATIC_OP, "MOV": T.COL_S_MOV, "JMP": T.COL_S_JMP, "CJMP": T.COL_S_CJMP, "END": T.COL_S_END,
            "RC": T.COL_S_RC, "AND": T.COL_S_BITWISE, "OR": T.COL_S_BITWISE, "XOR": T.COL_S_BITWISE, "GTE": T.COL_S_GTE}


def execute(prog, max_steps=1 << 16, tree=None):
    """-> (cpu rows as dicts of column -> value, side effects {'rc': [...], 'bitwise': [...], 'cmp': [...], ...}, executed
    words).  `tree`: the StorageTree SSTORE / SLOAD work on (a fresh empty one by default)."""
    words, pcs = prog.words()
    pc_to_idx = {pc: i for i, pc in enumerate(pcs)}
    regs = [0] * REG
    pc = clk = 0
    rows, side, executed = [], {"rc": [], "bitwise": [], "cmp": [], "mem": [], "psdn": [], "tape": [], "storage": [], "storage_psdn": []}, []
    memory = {}
    tape, tp = {}, 0
    tree = tree if tree is not None else StorageTree()
    idx_storage = 0
    while True:
        assert len(rows) < max_steps, "program does not terminate"
        op, dst, op0, op1 = prog.ins[pc_to_idx[pc]]
        imm = isinstance(op1, tuple)
        r = {c: 0 for c in range(T.NUM_CPU_COLS)}
        for k, v in zip(T.COL_ADDR_STORAGE_RANGE, prog.storage_addr):
            r[k] = v
        for k, v in zip(T.COL_ADDR_CODE_RANGE, prog.code_addr):
            r[k] = v
        r[T.COL_CLK], r[T.COL_PC], r[T.COL_TP], r[T.COL_IDX_STORAGE] = clk, pc, tp, idx_storage
        for i in range(REG):
            r[T.COL_REGS.start + i] = regs[i]
        r[T.COL_INST], r[T.COL_OP1_IMM], r[T.COL_OPCODE] = words[pc], int(imm), T.op_mask(op)
        r[SELECTOR[op]] = 1
        r[T.COL_IS_ENTRY_SC] = r[T.COL_IS_NEXT_LINE_DIFF_INST] = r[T.COL_IS_NEXT_LINE_SAME_TX] = 1
        v0 = regs[op0] if op0 is not None else 0
        v1 = (int(op1[1]) % P) if imm else (regs[op1] if op1 is not None else 0)
        if op0 is not None:
            r[T.COL_S_OP0.start + op0], r[T.COL_OP0] = 1, v0
        if imm:
            r[T.COL_IMM_VAL] = r[T.COL_OP1] = v1
            r[T.COL_FILTER_LOOKING_PROG_IMM] = 1
        elif op1 is not None:
            r[T.COL_S_OP1.start + op1], r[T.COL_OP1] = 1, v1
        size = 2 if imm else 1
        executed.append((pc, words[pc]))
        if imm:
            executed.append((pc + 1, v1))
        next_pc, res = pc + size, None
        if op == "MOV":
            res = v1
        elif op == "NOT":               # executor/src/lib.rs:602-605: dst = -1 - op1
            res = (P - 1 - v1) % P
        elif op == "ASSERT":            # executor/src/lib.rs:673-712: the operand must be 1 (cpu/simple_arithmatic_op.rs:52)
            assert v1 == 1, "ASSERT on a value other than 1"
        elif op == "ADD":
            res = (v0 + v1) % P
        elif op == "MUL":
            res = v0 * v1 % P
        elif op in ("EQ", "NEQ"):
            res = int((v0 == v1) == (op == "EQ"))
            r[T.COL_AUX0] = _inv(v0 - v1)
        elif op == "JMP":
            next_pc = v1
        elif op == "CJMP":
            assert v0 in (0, 1)
            next_pc = v1 if v0 else pc + size
        elif op == "RC":
            assert v1 < 1 << 32
            side["rc"].append(v1)
        elif op in ("AND", "OR", "XOR"):
            assert v0 < 1 << 32 and v1 < 1 << 32
            res = {"AND": v0 & v1, "OR": v0 | v1, "XOR": v0 ^ v1}[op]
            side["bitwise"].append((op, v0, v1))
        elif op == "GTE":
            assert v0 < 1 << 32 and v1 < 1 << 32
            res = int(v0 >= v1)
            side["cmp"].append((v0, v1))
        elif op in ("MSTORE", "MLOAD"):
            # executor/src/lib.rs:868-995: address = op0 + immediate offset (aux1); MSTORE writes the `dst` register, MLOAD
            # loads into it; both are two-word instructions
            assert imm, "only the [reg + imm] addressing form is implemented"
            addr = (v0 + v1) % P
            r[T.COL_AUX1] = addr
            if op == "MSTORE":
                res = regs[dst]
                memory[addr] = res
            else:
                assert addr in memory, "load from an address that was never written"
                res = memory[addr]
            side["mem"].append((addr, clk, op, res, int(op == "MSTORE")))
        elif op == "CALL":
            # executor/src/lib.rs:816-849: the return address goes to [fp - 1]; [fp - 2] (the caller's saved fp) is read
            fp = regs[REG - 1]
            assert imm and (fp - 2) % P in memory, "CALL needs an immediate target and a saved frame pointer at [fp - 2]"
            ret_pc = pc + size
            r[T.COL_OP0], r[T.COL_DST], r[T.COL_AUX0], r[T.COL_AUX1] = (fp - 1) % P, ret_pc, (fp - 2) % P, memory[(fp - 2) % P]
            memory[(fp - 1) % P] = ret_pc
            side["mem"].append(((fp - 1) % P, clk, "CALL", ret_pc, 1))
            side["mem"].append(((fp - 2) % P, clk, "CALL", memory[(fp - 2) % P], 0))
            next_pc = v1
        elif op == "RET":
            # executor/src/lib.rs:851-866: pc <- [fp - 1], fp <- [fp - 2]
            fp = regs[REG - 1]
            ret_pc, old_fp = memory[(fp - 1) % P], memory[(fp - 2) % P]
            r[T.COL_OP0], r[T.COL_DST], r[T.COL_AUX0], r[T.COL_AUX1] = (fp - 1) % P, ret_pc, (fp - 2) % P, old_fp
            side["mem"].append(((fp - 1) % P, clk, "RET", ret_pc, 0))
            side["mem"].append(((fp - 2) % P, clk, "RET", old_fp, 0))
            regs[REG - 1] = old_fp
            next_pc = ret_pc
        elif op == "POSEIDON":
            # executor/src/lib.rs:1547-1700: hash `len` (= op1, a multiple of 8 here) words at [op0..] eight at a time, the
            # capacity chained through the calls, and write the first four words of the last output to [dst..]
            src, length, dst_addr = v0, v1, regs[dst]
            assert length and length % 8 == 0, "only whole 8-word blocks are implemented"
            chunks, cap = [], [0, 0, 0, 0]
            for k in range(0, length, 8):
                vals = []
                for i in range(8):
                    a = src + k + i
                    assert a in memory, "hash input was never written"
                    vals.append(memory[a])
                    side["mem"].append((a, clk, "POSEIDON", memory[a], 0))
                row = poseidon_row(vals + cap, filters=(1, 0, 0, 0))
                chunks.append((src + k, vals, cap, row))
                cap = row[16:28][8:12]
            out = chunks[-1][3][16:20]
            for i in range(4):
                memory[dst_addr + i] = out[i]
                side["mem"].append((dst_addr + i, clk, "POSEIDON", out[i], 1))
            side["psdn"].append({"clk": clk, "src": src, "len": length, "dst": dst_addr, "chunks": chunks})
            res = dst_addr          # the CPU's dst column carries the destination address; registers do not change
        if op == "TLOAD":
            res = regs[dst]         # dst names the register holding the memory base; it does not change
        if dst is not None:
            assert res is not None
            r[T.COL_S_DST.start + dst], r[T.COL_DST] = 1, res
            regs[dst] = res
        if op == "END":
            r[T.COL_IS_NEXT_LINE_SAME_TX] = 0
            rows.append(r)
            break
        rows.append(r)
        if op in ("TSTORE", "TLOAD"):
            # executor/src/lib.rs:1687-1846 + cpu/tape.rs: one extension line per word moved between memory and the tape.
            # Extension lines repeat the instruction's clk / pc / opcode / selectors / op0 / op1; aux0 walks the memory
            # addresses, the first register-selector column (COL_S_OP0.start) the tape addresses, aux1 carries the word.
            if op == "TSTORE":
                length, mem_base, tape_base = v1, v0, tp
            else:
                assert v0 in (0, 1)
                length, mem_base, tape_base = (v1, regs[dst], tp - v1) if v0 else (1, regs[dst], v1)
            assert length >= 1
            r[T.COL_IS_NEXT_LINE_DIFF_INST] = 0
            for k in range(length):
                e = dict(r)
                for c in list(T.COL_S_OP0) + list(T.COL_S_OP1) + list(T.COL_S_DST):
                    e[c] = 0
                e[T.COL_INST] = e[T.COL_IMM_VAL] = e[T.COL_FILTER_LOOKING_PROG_IMM] = e[T.COL_DST] = 0
                e[T.COL_IS_EXT_LINE], e[T.COL_EXT_CNT], e[T.COL_FILTER_TAPE_LOOKING] = 1, k + 1, 1
                e[T.COL_IS_NEXT_LINE_DIFF_INST] = int(k == length - 1)
                maddr, taddr = mem_base + k, tape_base + k
                if op == "TSTORE":
                    assert maddr in memory, "tstore source was never written"
                    word = memory[maddr]
                    tape[taddr] = word
                    side["mem"].append((maddr, clk, "TSTORE", word, 0))
                else:
                    assert taddr in tape, "tload from a tape cell that was never written"
                    word = tape[taddr]
                    memory[maddr] = word
                    side["mem"].append((maddr, clk, "TLOAD", word, 1))
                side["tape"].append((taddr, len(side["tape"]), op, word))
                e[T.COL_AUX0], e[T.COL_S_OP0.start], e[T.COL_AUX1] = maddr, taddr, word
                rows.append(e)
            if op == "TSTORE":
                tp += length
        if op in ("SSTORE", "SLOAD"):
            # executor/src/lib.rs:1301-1545 + cpu/storage.rs: op0 / op1 hold the memory addresses of the 4-word slot key and
            # the 4-word value.  One extension line carries, in the (otherwise idle) register-selector columns, the eight
            # memory addresses, the key and value words and the tree key Poseidon(contract storage address || slot key);
            # it looks up memory (8 cells), the Poseidon table (tree key) and the storage table (leaf row of the proof).
            assert not imm and op0 is not None and op1 is not None
            key = []
            for i in range(4):
                assert (v0 + i) % P in memory, "storage key was never written"
                key.append(memory[(v0 + i) % P])
                side["mem"].append(((v0 + i) % P, clk, op, key[i], 0))
            krow = poseidon_row(list(prog.storage_addr) + key + [0, 0, 0, 0], filters=(0, 1, 0, 0))
            tree_key = krow[16:20]
            if op == "SSTORE":
                value = []
                for i in range(4):
                    assert (v1 + i) % P in memory, "stored value was never written"
                    value.append(memory[(v1 + i) % P])
                    side["mem"].append(((v1 + i) % P, clk, op, value[i], 0))
                srows, prows, _ = tree.access(tree_key, value)
            else:
                srows, prows, value = tree.access(tree_key)
                for i in range(4):
                    memory[(v1 + i) % P] = value[i]
                    side["mem"].append(((v1 + i) % P, clk, op, value[i], 1))
            idx_storage += 1
            side["storage"].append(srows)
            side["storage_psdn"] += [krow] + prows
            r[T.COL_IS_NEXT_LINE_DIFF_INST] = 0
            e = dict(r)
            for c in list(T.COL_S_OP0) + list(T.COL_S_OP1) + list(T.COL_S_DST):
                e[c] = 0
            e[T.COL_INST] = e[T.COL_IMM_VAL] = e[T.COL_FILTER_LOOKING_PROG_IMM] = e[T.COL_DST] = 0
            e[T.COL_IS_EXT_LINE] = e[T.COL_EXT_CNT] = e[T.COL_IS_STORAGE_EXT_LINE] = e[T.COL_IS_NEXT_LINE_DIFF_INST] = 1
            e[T.COL_IDX_STORAGE] = idx_storage
            for i in range(4):
                e[T.COL_S_OP0.start + i], e[T.COL_S_OP0.start + 4 + i] = (v0 + i) % P, key[i]
                e[T.COL_S_OP1.start + i], e[T.COL_S_OP1.start + 4 + i] = (v1 + i) % P, value[i]
                e[T.COL_S_DST.start + i] = tree_key[i]
            rows.append(e)
        pc, clk = next_pc, clk + 1
    return rows, side, executed


def cpu_trace(rows):
    n = TG.next_pow2(max(len(rows), 8))
    t = TG.cpu_padding_trace(n)
    for i, r in enumerate(rows):
        for c, v in r.items():
            t[c, i] = v
    t[T.COL_IDX_STORAGE, len(rows):] = rows[-1][T.COL_IDX_STORAGE]        # generation/cpu.rs:191-202
    return t


def program_trace(prog, executed, beta):
    """program/columns.rs: the listing side (every word of the program, looked up by prog_chunk) and the executed side
    (every instruction / immediate word the CPU fetched), each compressed with beta and linked by an in-table lookup."""
    words, _ = prog.words()
    words = words + [0] * (-len(words) % 8)              # prog_chunk hashes 8 words at a time
    n = TG.next_pow2(max(len(words), len(executed), 8))
    t = np.zeros((T.NUM_PROG_COLS, n), dtype=np.uint64)
    b = int(beta) % P

    def comp(pc, w):
        a = prog.code_addr
        return (a[0] + a[1] * b + a[2] * b ** 2 + a[3] * b ** 3 + pc * b ** 4 + w * b ** 5) % P
    for pc, w in enumerate(words):
        for k, v in zip(T.COL_PROG_CODE_ADDR_RANGE, prog.code_addr):
            t[k, pc] = v
        t[T.COL_PROG_PC, pc], t[T.COL_PROG_INST, pc], t[T.COL_PROG_COMP_PROG, pc], t[T.COL_PROG_FILTER_PROG_CHUNK, pc] = pc, w, comp(pc, w), 1
    for i in range(n):
        pc, w = executed[i] if i < len(executed) else executed[0]     # filler rows repeat a listed word (filter 0)
        for k, v in zip(T.COL_PROG_EXEC_CODE_ADDR_RANGE, prog.code_addr):
            t[k, i] = v
        t[T.COL_PROG_EXEC_PC, i], t[T.COL_PROG_EXEC_INST, i], t[T.COL_PROG_EXEC_COMP_PROG, i] = pc, w, comp(pc, w)
        t[T.COL_PROG_FILTER_EXEC, i] = int(i < len(executed))
    pi, pt = TG.permuted_cols([int(x) for x in t[T.COL_PROG_EXEC_COMP_PROG]], [int(x) for x in t[T.COL_PROG_COMP_PROG]])
    t[T.COL_PROG_EXEC_COMP_PROG_PERM], t[T.COL_PROG_COMP_PROG_PERM] = pi, pt
    return t, words


def program_hash(words):
    """First four words of the chained chunk hash of a (zero-padded) program listing: what the state tree stores at the
    contract's code address (program/prog_chunk_stark.rs:51-61 ties the result line to that leaf)."""
    cap = [0, 0, 0, 0]
    for i in range(0, len(words), 8):
        h = poseidon_row(list(words[i:i + 8]) + cap)[16:28]
        cap = h[8:12]
    return tuple(h[0:4])


def prog_chunk_and_poseidon(prog, words, n_poseidon_min=8, extra_rows=(), result_line=False):
    """One prog_chunk row per 8 program words, hashed with Poseidon (capacity = last third of the previous chunk's hash);
    with `result_line` the last row is marked as the one carrying the program hash, which the storage table must then
    prove to be the leaf at the code address.  -> (prog_chunk trace, Poseidon-table trace carrying those permutations)."""
    chunks = [words[i:i + 8] for i in range(0, len(words), 8)]
    n = TG.next_pow2(max(len(chunks), 8))
    t = TG.flag_padding_trace(T.NUM_PROG_CHUNK_COLS, n, T.COL_PROG_CHUNK_IS_PADDING_LINE)
    prow, cap = [], [0, 0, 0, 0]
    for i, ch in enumerate(chunks):
        row = poseidon_row(list(ch) + cap, filters=(1, 0, 0, 0))
        prow.append(row)
        h = row[16:28]
        t[T.COL_PROG_CHUNK_IS_PADDING_LINE, i] = 0
        for k, v in zip(T.COL_PROG_CHUNK_CODE_ADDR_RANGE, prog.code_addr):
            t[k, i] = v
        t[T.COL_PROG_CHUNK_START_PC, i] = 8 * i
        for k, v in zip(T.COL_PROG_CHUNK_INST_RANGE, ch):
            t[k, i] = v
        for k, v in zip(T.COL_PROG_CHUNK_CAP_RANGE, cap):
            t[k, i] = v
        for k, v in zip(T.COL_PROG_CHUNK_HASH_RANGE, h):
            t[k, i] = v
        t[T.COL_PROG_CHUNK_IS_FIRST_LINE, i] = int(i == 0)
        t[T.COL_PROG_CHUNK_IS_RESULT_LINE, i] = int(result_line and i == len(chunks) - 1)
        for k in T.COL_PROG_CHUNK_FILTER_LOOKING_PROG_RANGE:
            t[k, i] = 1
        cap = h[8:12]
    prow = prow + list(extra_rows)        # permutations of the Poseidon builtin share the table
    np_ = TG.next_pow2(max(len(prow), n_poseidon_min))
    pt = TG.poseidon_padding_trace(np_)
    for i, row in enumerate(prow):
        pt[:, i] = np.array(row, dtype=np.uint64)
    return t, pt


def memory_trace(cells, reference_quirks=False):
    """Memory table of stack- and heap-region accesses (generation/memory.rs:5-95; cell rules of core's memory trace): rows
    sorted by (address, clk); diff / rw_addr_unchanged / rc_value columns relate each row to its predecessor, rc_value
    (the clock difference on an unchanged address, else the address difference) is range-checked through the
    memory<->rangecheck lookup for every row but the first and the first heap row.  Heap cells (addresses from
    ADDR_HEAP_PTR up to p - 2^32) additionally range-check their distance to the top of the region (diff_addr_cond,
    memory_stark.rs:232,330).  The live rows are followed by the prophet-region padding of tracegen.py.
    -> (trace, range-checked sort values, range-checked region values)"""
    cells = sorted(cells)
    n = TG.next_pow2(max(len(cells) + 1, 8))
    if reference_quirks and not cells:          # generation/memory.rs:95-153 as it is (tracegen.memory_padding_trace)
        return TG.memory_padding_trace(n, reference_quirks=True), [], []
    t = np.zeros((T.NUM_MEM_COLS, n), dtype=np.uint64)
    rc_vals, cond_vals = [], []
    prev = None
    span = 2**32 - 1
    sel = {"SSTORE": T.COL_MEM_S_SSTORE, "SLOAD": T.COL_MEM_S_SLOAD, "TSTORE": T.COL_MEM_S_TSTORE, "TLOAD": T.COL_MEM_S_TLOAD, "MSTORE": T.COL_MEM_S_MSTORE, "MLOAD": T.COL_MEM_S_MLOAD, "POSEIDON": T.COL_MEM_S_POSEIDON, "CALL": T.COL_MEM_S_CALL,
           "RET": T.COL_MEM_S_RET}
    for i, (addr, clk, op, value, is_write) in enumerate(cells):
        t[T.COL_MEM_IS_RW, i] = 1
        t[T.COL_MEM_ADDR, i], t[T.COL_MEM_CLK, i], t[T.COL_MEM_OP, i], t[T.COL_MEM_VALUE, i] = addr, clk, T.op_mask(op), value
        t[sel[op], i] = 1
        t[T.COL_MEM_IS_WRITE, i] = is_write
        heap = addr >= T.ADDR_HEAP_PTR
        if heap:
            cond = (0 - span - addr) % P
            t[T.COL_MEM_REGION_HEAP, i], t[T.COL_MEM_DIFF_ADDR_COND, i], t[T.COL_MEM_FILTER_LOOKING_RC_COND, i] = 1, cond, 1
            cond_vals.append(cond)
        if prev is not None and heap and not prev[2]:
            # first heap row: the address gap to the stack region is not range-checked (generation/memory.rs:77-86)
            t[T.COL_MEM_DIFF_ADDR, i], t[T.COL_MEM_DIFF_ADDR_INV, i] = addr - prev[0], _inv(addr - prev[0])
        elif prev is not None:
            same = int(addr == prev[0])
            d_addr = addr - prev[0]
            t[T.COL_MEM_DIFF_ADDR, i], t[T.COL_MEM_DIFF_ADDR_INV, i] = d_addr, _inv(d_addr)
            t[T.COL_MEM_DIFF_CLK, i] = (clk - prev[1]) if same else 0
            t[T.COL_MEM_RW_ADDR_UNCHANGED, i] = same
            rc = (clk - prev[1]) if same else d_addr
            t[T.COL_MEM_RC_VALUE, i], t[T.COL_MEM_FILTER_LOOKING_RC, i] = rc, 1
            rc_vals.append(rc)
        prev = (addr, clk, heap)
    # prophet-region padding (tracegen.memory_padding_trace), continuing from the last live address
    a = (0 - span) % P
    last_addr = prev[0] if prev else 0
    start = len(cells) if cells else 1
    if not cells:
        t[T.COL_MEM_S_PROPHET, 0] = t[T.COL_MEM_IS_WRITE, 0] = 1
    for i in range(start, n):
        t[T.COL_MEM_S_PROPHET, i] = t[T.COL_MEM_IS_WRITE, i] = t[T.COL_MEM_REGION_PROPHET, i] = 1
        t[T.COL_MEM_ADDR, i] = a
        d = (a - last_addr) % P if i == start else 1
        t[T.COL_MEM_DIFF_ADDR, i], t[T.COL_MEM_DIFF_ADDR_INV, i] = d, _inv(d)
        t[T.COL_MEM_DIFF_ADDR_COND, i] = t[T.COL_MEM_RC_VALUE, i] = (0 - a) % P
        a = (a + 1) % P
    return t, rc_vals, cond_vals


def poseidon_chunk_trace(calls):
    """Poseidon-builtin table (generation/poseidon_chunk.rs; builtins/poseidon/poseidon_chunk_stark.rs:98-284): per POSEIDON
    instruction one header row (looked up by the CPU) and one extension row per 8-word block (looking up its 8 source
    words in memory and its permutation in the Poseidon table); the last one is the result line (its first 4 hash words go
    to memory).  -> (trace, Poseidon-table rows)"""
    rows, prow = [], []
    for c in calls:
        base = {T.COL_POSEIDON_CHUNK_CLK: c["clk"], T.COL_POSEIDON_CHUNK_OPCODE: T.op_mask("POSEIDON"),
                T.COL_POSEIDON_CHUNK_OP1: c["len"], T.COL_POSEIDON_CHUNK_DST: c["dst"]}
        head = dict(base)
        head[T.COL_POSEIDON_CHUNK_OP0] = c["src"]
        head[T.COL_POSEIDON_CHUNK_FILTER_LOOKED_CPU] = 1
        rows.append(head)
        for j, (addr, vals, cap, prow_j) in enumerate(c["chunks"]):
            r = dict(base)
            r[T.COL_POSEIDON_CHUNK_OP0] = addr
            r[T.COL_POSEIDON_CHUNK_ACC_CNT] = 8 * (j + 1)
            for k, v in zip(T.COL_POSEIDON_CHUNK_VALUE_RANGE, vals):
                r[k] = v
            for k, v in zip(T.COL_POSEIDON_CHUNK_CAP_RANGE, cap):
                r[k] = v
            for k, v in zip(T.COL_POSEIDON_CHUNK_HASH_RANGE, prow_j[16:28]):
                r[k] = v
            r[T.COL_POSEIDON_CHUNK_IS_EXT_LINE] = 1
            r[T.COL_POSEIDON_CHUNK_IS_RESULT_LINE] = int(j == len(c["chunks"]) - 1)
            for k in T.COL_POSEIDON_CHUNK_FILTER_LOOKING_MEM_RANGE:
                r[k] = 1
            r[T.COL_POSEIDON_CHUNK_FILTER_LOOKING_POSEIDON] = 1
            rows.append(r)
            prow.append(prow_j)
    n = TG.next_pow2(max(len(rows), 8))
    t = TG.flag_padding_trace(T.NUM_POSEIDON_CHUNK_COLS, n, T.COL_POSEIDON_CHUNK_IS_PADDING_LINE)
    for i, r in enumerate(rows):
        t[T.COL_POSEIDON_CHUNK_IS_PADDING_LINE, i] = 0
        for k, v in r.items():
            t[k, i] = v
    return t, prow


def storage_trace(accesses, prog_reads=()):
    """Storage-access table (generation/storage.rs:7-123; builtins/storage/storage_access_stark.rs:110-334): 256 rows per
    proof, the CPU's accesses in execution order followed by the program-hash reads; the access index counts proofs from
    1, addr_acc rebuilds each 64-bit limb of the tree key from the layer bits, the layer-256 row is the one the CPU (or
    the prog_chunk result line) looks up."""
    flat = [(r, i + 1, False) for i, rows in enumerate(accesses) for r in rows]
    flat += [(r, len(accesses) + i + 1, True) for i, rows in enumerate(prog_reads) for r in rows]
    n = TG.next_pow2(max(len(flat), 8))
    t = TG.flag_padding_trace(T.NUM_COL_ST, n, T.COL_ST_IS_PADDING)
    acc = 0
    for i, (r, idx, for_prog) in enumerate(flat):
        layer = r["layer"]
        acc = r["bit"] if layer % 64 == 1 else (2 * acc + r["bit"]) % P
        t[T.COL_ST_IS_PADDING, i] = 0
        t[T.COL_ST_ACCESS_IDX, i], t[T.COL_ST_IS_WRITE, i], t[T.COL_ST_LAYER, i], t[T.COL_ST_LAYER_BIT, i] = idx, r["is_write"], layer, r["bit"]
        t[T.COL_ST_ADDR_ACC, i], t[T.COL_ST_HASH_TYPE, i] = acc, int(layer == 256)
        for rng, key in ((T.COL_ST_PRE_ROOT_RANGE, "pre_root"), (T.COL_ST_ROOT_RANGE, "root"), (T.COL_ST_ADDR_RANGE, "addr"),
                         (T.COL_ST_PRE_PATH_RANGE, "pre_path"), (T.COL_ST_PATH_RANGE, "path"), (T.COL_ST_SIB_RANGE, "sib"),
                         (T.COL_ST_PRE_HASH_RANGE, "pre_hash"), (T.COL_ST_HASH_RANGE, "hash")):
            for c, v in zip(rng, r[key]):
                t[c, i] = v
        for c, at in ((T.COL_ST_IS_LAYER_1, 1), (T.COL_ST_IS_LAYER_64, 64), (T.COL_ST_IS_LAYER_128, 128), (T.COL_ST_IS_LAYER_192, 192),
                      (T.COL_ST_IS_LAYER_256, 256)):
            t[c, i] = int(layer == at)
        t[T.COL_ST_ACC_LAYER_MARKER, i] = 1 + layer // 64
        t[T.COL_ST_FILTER_IS_HASH_BIT_0, i], t[T.COL_ST_FILTER_IS_HASH_BIT_1, i] = 1 - r["bit"], r["bit"]
        t[T.COL_ST_FILTER_IS_FOR_PROG, i] = int(for_prog and layer == 256)
    if flat:
        for c, v in zip(T.COL_ST_ROOT_RANGE, flat[-1][0]["root"]):
            t[c, len(flat):] = v
    return t


def tape_trace(cells):
    """Tape table (builtins/tape/tape_stark.rs:44-143): cells sorted by tape address, the write (TSTORE) of an address
    first, then its reads (TLOAD); all are looked up by the CPU's extension lines.  Padding repeats the last cell as an
    unfiltered TLOAD."""
    cells = sorted(cells, key=lambda c: (c[0], c[1]))
    n = TG.next_pow2(max(len(cells), 8))
    if not cells:
        return TG.tape_padding_trace(n)
    t = np.zeros((T.NUM_COL_TAPE, n), dtype=np.uint64)
    for i in range(n):
        addr, _, op, word = cells[min(i, len(cells) - 1)]
        live = i < len(cells)
        t[T.COL_TAPE_OPCODE, i] = T.op_mask(op if live else "TLOAD")
        t[T.COL_TAPE_ADDR, i], t[T.COL_TAPE_VALUE, i], t[T.COL_TAPE_FILTER_LOOKED, i] = addr, word, int(live)
    return t


def _transcript():
    """Fiat-Shamir transcript for the generators' compress challenges: the library's host-side Challenger
    (iop/challenger.rs:36-162; host code of libola_gpu.so, no GPU involved)."""
    from olavm_amd.backend import Challenger
    return Challenger()


def derive_program_beta(start_root, end_root):
    """generation/prog.rs:23-29: the transcript observes the state roots before and after the run, limb by limb."""
    ch = _transcript()
    for a, b in zip(start_root, end_root):
        ch.observe([int(a) % P, int(b) % P])
    return ch.get()


def instance(prog, range_bits=4, limb_bits=2, bitwise_beta=None, program_beta=None, prove_program_hash=False, max_steps=1 << 16, reference_quirks=False):
    """The 12 traces (enum Table order), params and compress challenges of ola_stark(range_bits, limb_bits) for one run of
    `prog`.  reference_quirks: the two places where the reference's generators write something its own AIR rejects are reproduced
    (bitwise limb 3, memory table of a run without memory cells -- tracegen.bitwise_trace / memory_padding_trace): for comparing whole
    pipelines with a build of the reference, not for proving.  The compress challenges of the bitwise and program tables are derived as the reference derives them (a
    transcript over the bitwise limb columns, generation/builtin.rs:120-131, and over the start / end state roots,
    generation/prog.rs:23-29); explicit values are for tests only.  With miniature fixed tables, RC / GTE operands must stay below 2^(2*range_bits) and bitwise operands below
    2^(4*limb_bits).  `prove_program_hash`: close the program-hash chain with a result line and a state-tree proof that
    the hash is the leaf at the code address (256 storage rows, 512 Poseidon rows)."""
    listing = prog.words()[0]
    listing = listing + [0] * (-len(listing) % 8)
    tree = StorageTree()
    if prove_program_hash:
        tree.set(prog.code_addr, program_hash(listing))
    start_root = tree.root()
    rows, side, executed = execute(prog, max_steps=max_steps, tree=tree)
    if program_beta is None:
        program_beta = derive_program_beta(start_root, tree.root())
    cpu = cpu_trace(rows)
    program, words = program_trace(prog, executed, program_beta)
    pchunk, builtin_rows = poseidon_chunk_trace(side["psdn"])
    prog_reads = []
    if prove_program_hash:
        srows, prows, _ = tree.access(prog.code_addr)
        prog_reads.append(srows)
        side["storage_psdn"] += prows
    chunk, poseidon = prog_chunk_and_poseidon(prog, words, extra_rows=builtin_rows + side["storage_psdn"], result_line=prove_program_hash)
    cmp_rows = []
    for a, b in side["cmp"]:
        d = abs(a - b)
        cmp_rows.append((a, b, int(a >= b), d, _inv(d), 1))
    mem, mem_rc, mem_cond = memory_trace(side["mem"], reference_quirks=reference_quirks)
    rc_rows = ([(v, 1, 0, 0, 0) for v in side["rc"]] + [(r[3], 0, 0, 0, 1) for r in cmp_rows] + [(v, 0, 1, 0, 0) for v in mem_rc]
               + [(v, 0, 0, 1, 0) for v in mem_cond])
    bitwise = TG.bitwise_trace(bitwise_beta, limb_bits, side["bitwise"], looked_by_cpu=True, transcript=_transcript if bitwise_beta is None else None,
                               reference_quirks=reference_quirks)
    if bitwise_beta is None:
        bitwise, bitwise_beta = bitwise
    traces = [
        cpu, mem, bitwise,
        TG.generate_cmp_trace(cmp_rows), TG.generate_rc_trace(rc_rows, range_bits), poseidon,
        pchunk,
        storage_trace(side["storage"], prog_reads),
        tape_trace(side["tape"]),
        TG.flag_padding_trace(T.NUM_COL_SCCALL, 8, T.COL_SCCALL_IS_PADDING),
        program, chunk,
    ]
    return traces, [bitwise_beta, program_beta], [0, 0, bitwise_beta, 0, 0, 0, 0, 0, 0, 0, program_beta, 0]


def fibonacci(count, a_reg=1, b_reg=2):
    """r1, r2 <- consecutive Fibonacci numbers, `count` loop iterations driven by a counter, EQ and CJMP."""
    p = Program()
    p.add("MOV", dst=a_reg, op1=("imm", 0)).add("MOV", dst=b_reg, op1=("imm", 1)).add("MOV", dst=3, op1=("imm", 0))
    loop = len(p.words()[0])
    p.add("ADD", dst=4, op0=a_reg, op1=b_reg).add("MOV", dst=a_reg, op1=b_reg).add("MOV", dst=b_reg, op1=4)
    p.add("ADD", dst=3, op0=3, op1=("imm", 1)).add("NEQ", dst=5, op0=3, op1=("imm", count)).add("CJMP", op0=5, op1=("imm", loop))
    p.add("END")
    return p


def fibonacci_loop(n=47, reps=1000):
    """The shape of the reference's headline benchmark (circuits/benches/fibo_loop.rs:46, README.md:69: calldata [47, 1000] =
    Fibonacci(47) computed 1000 times): an outer counter around the Fibonacci loop.  r1, r2 <- consecutive Fibonacci numbers
    (F(47) = 2971215073 fits 32 bits), 6 instructions per inner step, 6 per repetition."""
    p = Program()
    p.add("MOV", dst=6, op1=("imm", 0))
    outer = len(p.words()[0])
    p.add("MOV", dst=1, op1=("imm", 0)).add("MOV", dst=2, op1=("imm", 1)).add("MOV", dst=3, op1=("imm", 0))
    loop = len(p.words()[0])
    p.add("ADD", dst=4, op0=1, op1=2).add("MOV", dst=1, op1=2).add("MOV", dst=2, op1=4)
    p.add("ADD", dst=3, op0=3, op1=("imm", 1)).add("NEQ", dst=5, op0=3, op1=("imm", n)).add("CJMP", op0=5, op1=("imm", loop))
    p.add("ADD", dst=6, op0=6, op1=("imm", 1)).add("NEQ", dst=5, op0=6, op1=("imm", reps)).add("CJMP", op0=5, op1=("imm", outer))
    p.add("END")
    return p


def mixed_program():
    """Every supported instruction at least once, with values that fit the miniature fixed tables (< 2^8): arithmetic,
    comparisons, the three bitwise operations, a range check, a taken and a not-taken conditional jump and a jump."""
    p = Program()
    p.add("MOV", dst=0, op1=("imm", 200)).add("MOV", dst=1, op1=("imm", 77))
    p.add("AND", dst=2, op0=0, op1=1).add("OR", dst=3, op0=0, op1=1).add("XOR", dst=4, op0=0, op1=("imm", 15))
    p.add("GTE", dst=5, op0=0, op1=1).add("GTE", dst=6, op0=1, op1=0).add("GTE", dst=7, op0=1, op1=("imm", 77))
    p.add("RC", op1=2).add("RC", op1=("imm", 255))
    p.add("MUL", dst=8, op0=2, op1=3).add("EQ", dst=8, op0=8, op1=8).add("ASSERT", op1=8).add("NEQ", dst=8, op0=5, op1=6)
    p.add("NOT", dst=7, op1=0).add("NOT", dst=7, op1=("imm", 5))
    skip = len(p.words()[0]) + 2 + 2          # target: the instruction after the next (2-word) MOV
    p.add("CJMP", op0=6, op1=("imm", 0))      # r6 = 0: not taken
    p.add("CJMP", op0=5, op1=("imm", skip + 2))   # r5 = 1: taken, jumps over the MOV below
    p.add("MOV", dst=8, op1=("imm", 999))
    end = len(p.words()[0]) + 2
    p.add("JMP", op1=("imm", end))
    p.add("END")
    return p


def memory_program(count=6):
    """Stores the first `count` Fibonacci numbers to memory cells 1..count (stack region), then reads them back in reverse
    order and sums them: MSTORE / MLOAD rows behind the cpu<->memory lookup, a memory table whose sort columns are
    range-checked, several accesses per address."""
    p = Program()
    p.add("MOV", dst=1, op1=("imm", 0)).add("MOV", dst=2, op1=("imm", 1)).add("MOV", dst=3, op1=("imm", 0))   # r3: address cursor
    loop = len(p.words()[0])
    p.add("ADD", dst=3, op0=3, op1=("imm", 1)).add("MSTORE", dst=1, op0=3, op1=("imm", 0))
    p.add("ADD", dst=4, op0=1, op1=2).add("MOV", dst=1, op1=2).add("MOV", dst=2, op1=4)
    p.add("NEQ", dst=5, op0=3, op1=("imm", count)).add("CJMP", op0=5, op1=("imm", loop))
    p.add("MOV", dst=6, op1=("imm", 0))                                                                         # r6: running sum
    back = len(p.words()[0])
    p.add("MLOAD", dst=7, op0=3, op1=("imm", 0)).add("ADD", dst=6, op0=6, op1=7)
    p.add("MSTORE", dst=6, op0=3, op1=("imm", 0))                                                               # overwrite with the partial sum
    p.add("MLOAD", dst=8, op0=3, op1=("imm", 0))
    p.add("ADD", dst=3, op0=3, op1=("imm", P - 1)).add("NEQ", dst=5, op0=3, op1=("imm", 0)).add("CJMP", op0=5, op1=("imm", back))
    p.add("END")
    return p


def hash_program(blocks=2):
    """Writes 8*blocks words to memory and hashes them with the POSEIDON builtin, then loads the digest back: the
    cpu<->poseidon_chunk, poseidon_chunk<->memory (8 source + 4 destination words) and poseidon_chunk<->poseidon lookups."""
    p = Program()
    n = 8 * blocks
    p.add("MOV", dst=1, op1=("imm", 1)).add("MOV", dst=2, op1=("imm", 3))        # r1: address cursor, r2: running value
    loop = len(p.words()[0])
    p.add("MSTORE", dst=2, op0=1, op1=("imm", 0)).add("MUL", dst=2, op0=2, op1=("imm", 5)).add("ADD", dst=1, op0=1, op1=("imm", 1))
    p.add("NEQ", dst=5, op0=1, op1=("imm", n + 1)).add("CJMP", op0=5, op1=("imm", loop))
    p.add("MOV", dst=3, op1=("imm", 1)).add("MOV", dst=4, op1=("imm", 100))      # r3: source, r4: destination
    p.add("POSEIDON", dst=4, op0=3, op1=("imm", n))
    p.add("MLOAD", dst=6, op0=4, op1=("imm", 0)).add("MLOAD", dst=7, op0=4, op1=("imm", 3))
    p.add("END")
    return p


def call_program():
    """A function call: the caller sets up a frame (fp = r9), saves its frame pointer at [fp - 2], CALLs a routine that
    doubles r1 three times and RETurns; executed twice."""
    p = Program()
    p.add("MOV", dst=9, op1=("imm", 10)).add("MOV", dst=8, op1=("imm", 4))          # fp = 10; the value RET restores into fp
    p.add("MSTORE", dst=8, op0=9, op1=("imm", P - 2))                                # [fp - 2] <- 4
    p.add("MOV", dst=1, op1=("imm", 3))
    words_before = len(p.words()[0])
    routine = words_before + 2 + 2 + 2 + 2 + 2 + 1                                   # after: CALL, MOV, MSTORE, CALL, END-jump
    p.add("CALL", op1=("imm", routine))
    p.add("MOV", dst=9, op1=("imm", 10)).add("MSTORE", dst=8, op0=9, op1=("imm", P - 2))   # same frame again
    p.add("CALL", op1=("imm", routine))
    p.add("JMP", op1=("imm", routine + 4))                                           # over the routine to END
    # pad so that the routine starts exactly at `routine`
    while len(p.words()[0]) < routine:
        p.add("ADD", dst=7, op0=7, op1=7)
    p.add("ADD", dst=1, op0=1, op1=1).add("ADD", dst=1, op0=1, op1=1).add("ADD", dst=1, op0=1, op1=1).add("RET")
    p.add("END")
    return p


def tape_program():
    """Writes three words to memory, appends them to the tape (TSTORE), reads the last two back to another place (TLOAD,
    flag 1) and one by absolute tape address (TLOAD, flag 0), then adds what it loaded: multi-line instructions (CPU
    extension lines), the tape table and the cpu<->tape / cpu<->memory lookups of the extension lines."""
    p = Program()
    p.add("MOV", dst=1, op1=("imm", 20)).add("MOV", dst=2, op1=("imm", 7))
    for k, v in enumerate((7, 11, 13)):
        p.add("MOV", dst=2, op1=("imm", v)).add("MSTORE", dst=2, op0=1, op1=("imm", k))
    p.add("TSTORE", op0=1, op1=("imm", 3))                                  # tape[0..3) <- mem[20..23)
    p.add("MOV", dst=3, op1=("imm", 40)).add("MOV", dst=4, op1=("imm", 1)).add("MOV", dst=5, op1=("imm", 0))
    p.add("TLOAD", dst=3, op0=4, op1=("imm", 2))                            # mem[40..42) <- tape[tp-2..tp)
    p.add("MOV", dst=3, op1=("imm", 50)).add("TLOAD", dst=3, op0=5, op1=("imm", 0))   # mem[50] <- tape[0]
    p.add("MOV", dst=6, op1=("imm", 40)).add("MLOAD", dst=7, op0=6, op1=("imm", 0)).add("MLOAD", dst=8, op0=6, op1=("imm", 1))
    p.add("MLOAD", dst=6, op0=6, op1=("imm", 10)).add("ADD", dst=7, op0=7, op1=8).add("ADD", dst=7, op0=7, op1=6)
    p.add("END")
    return p


def storage_program():
    """Writes a slot key and two values to memory, SSTOREs the first value, SLOADs it back to another place, overwrites the
    slot with the second value and reads that too: CPU extension lines of the storage instructions, three state-tree
    updates / reads of 256 levels each (storage table), their 2 x 256 Poseidon rows each, the tree-key hash, and the
    cpu<->storage, storage<->poseidon, cpu<->poseidon(tree key) and cpu<->memory lookups."""
    p = Program()
    p.add("MOV", dst=1, op1=("imm", 10)).add("MOV", dst=2, op1=("imm", 20)).add("MOV", dst=3, op1=("imm", 30)).add("MOV", dst=4, op1=("imm", 40))
    for k in range(4):
        p.add("MOV", dst=5, op1=("imm", 1000 + k)).add("MSTORE", dst=5, op0=1, op1=("imm", k))       # slot key at [10..14)
        p.add("MOV", dst=5, op1=("imm", 7 * k + 1)).add("MSTORE", dst=5, op0=2, op1=("imm", k))      # first value at [20..24)
        p.add("MOV", dst=5, op1=("imm", P - 1 - k)).add("MSTORE", dst=5, op0=3, op1=("imm", k))      # second value at [30..34)
    p.add("SSTORE", op0=1, op1=2).add("SLOAD", op0=1, op1=4)
    p.add("SSTORE", op0=1, op1=3).add("SLOAD", op0=1, op1=4)
    p.add("MLOAD", dst=6, op0=4, op1=("imm", 0)).add("MLOAD", dst=7, op0=4, op1=("imm", 3)).add("ADD", dst=6, op0=6, op1=7)
    p.add("END")
    return p


def heap_program():
    """Stores to and loads from the heap region (the addresses just below p - 2^32 + 1): memory rows whose distance to the
    top of the region goes through the memory<->rangecheck region lookup, after stack-region rows of the same run."""
    p = Program()
    top = P - (2**32 - 1)
    p.add("MOV", dst=1, op1=("imm", 5)).add("MOV", dst=2, op1=("imm", 42)).add("MSTORE", dst=2, op0=1, op1=("imm", 0))     # stack cell
    p.add("MOV", dst=3, op1=("imm", top - 9))
    p.add("MSTORE", dst=2, op0=3, op1=("imm", 0)).add("ADD", dst=2, op0=2, op1=2).add("MSTORE", dst=2, op0=3, op1=("imm", 4))
    p.add("MLOAD", dst=4, op0=3, op1=("imm", 0)).add("MLOAD", dst=5, op0=3, op1=("imm", 4)).add("MLOAD", dst=6, op0=1, op1=("imm", 0))
    p.add("ADD", dst=4, op0=4, op1=5).add("MSTORE", dst=4, op0=3, op1=("imm", 0))
    p.add("END")
    return p


def wide_program():
    """For the full-size fixed tables (range_bits 16, limb_bits 8) only: 32-bit operands through the bitwise, comparison and
    range-check instructions (all four 8-bit limbs and both 16-bit limbs of every looked-up value are non-trivial), kept
    in memory between uses."""
    p = Program()
    vals = (0xDEADBEEF, 0x12345678, 0xFFFFFFFF, 0x80000001, 0x0000FFFF, 0xA5A5A5A5)
    p.add("MOV", dst=1, op1=("imm", 100))
    for k, v in enumerate(vals):
        p.add("MOV", dst=2, op1=("imm", v)).add("MSTORE", dst=2, op0=1, op1=("imm", k))
    for k in range(len(vals) - 1):
        p.add("MLOAD", dst=2, op0=1, op1=("imm", k)).add("MLOAD", dst=3, op0=1, op1=("imm", k + 1))
        p.add("AND", dst=4, op0=2, op1=3).add("OR", dst=5, op0=2, op1=3).add("XOR", dst=6, op0=2, op1=3)
        p.add("GTE", dst=7, op0=2, op1=3).add("GTE", dst=8, op0=6, op1=4)
        p.add("RC", op1=4).add("RC", op1=5).add("RC", op1=6)
        p.add("MSTORE", dst=6, op0=1, op1=("imm", 10 + k))
    p.add("XOR", dst=2, op0=2, op1=("imm", 0xFFFFFFFF)).add("RC", op1=2).add("GTE", dst=7, op0=2, op1=("imm", 0x7FFFFFFF))
    p.add("END")
    return p


def storage_heavy_program(slots, cells=0):
    """BASELINE config 4 as an execution: `slots` iterations that each SSTORE a fresh slot and SLOAD it back -- two 256-level
    state-tree proofs, 1026 Poseidon-table rows and 512 storage-table rows per iteration -- followed, when cells > 0, by the
    store / load loops of memory_program(cells) to fill the CPU and memory tables.  slots = 4085 makes a 2^22-row Poseidon
    table and a 2^21-row storage table."""
    p = Program()
    p.add("MOV", dst=1, op1=("imm", 10)).add("MOV", dst=2, op1=("imm", 20)).add("MOV", dst=4, op1=("imm", 40)).add("MOV", dst=3, op1=("imm", 0))
    for k in range(1, 4):
        p.add("MOV", dst=5, op1=("imm", 1000 + k)).add("MSTORE", dst=5, op0=1, op1=("imm", k))        # key words 1..3
        p.add("MOV", dst=5, op1=("imm", 7 * k + 1)).add("MSTORE", dst=5, op0=2, op1=("imm", k))       # value words 1..3
    loop = len(p.words()[0])
    p.add("ADD", dst=3, op0=3, op1=("imm", 1)).add("MSTORE", dst=3, op0=1, op1=("imm", 0)).add("MUL", dst=6, op0=3, op1=("imm", 3))
    p.add("MSTORE", dst=6, op0=2, op1=("imm", 0)).add("SSTORE", op0=1, op1=2).add("SLOAD", op0=1, op1=4)
    p.add("NEQ", dst=5, op0=3, op1=("imm", slots)).add("CJMP", op0=5, op1=("imm", loop))
    if cells:
        # memory_program's two loops, on addresses above everything used so far
        p.add("MOV", dst=1, op1=("imm", 0)).add("MOV", dst=2, op1=("imm", 1)).add("MOV", dst=3, op1=("imm", 100))
        first = len(p.words()[0])
        p.add("ADD", dst=3, op0=3, op1=("imm", 1)).add("MSTORE", dst=1, op0=3, op1=("imm", 0))
        p.add("ADD", dst=4, op0=1, op1=2).add("MOV", dst=1, op1=2).add("MOV", dst=2, op1=4)
        p.add("NEQ", dst=5, op0=3, op1=("imm", 100 + cells)).add("CJMP", op0=5, op1=("imm", first))
        p.add("MOV", dst=6, op1=("imm", 0))
        back = len(p.words()[0])
        p.add("MLOAD", dst=7, op0=3, op1=("imm", 0)).add("ADD", dst=6, op0=6, op1=7).add("MSTORE", dst=6, op0=3, op1=("imm", 0))
        p.add("MLOAD", dst=8, op0=3, op1=("imm", 0))
        p.add("ADD", dst=3, op0=3, op1=("imm", P - 1)).add("NEQ", dst=5, op0=3, op1=("imm", 100)).add("CJMP", op0=5, op1=("imm", back))
    p.add("END")
    return p


# name -> (program factory, keyword arguments of instance()): the executions the tests prove
EXAMPLES = {"fibonacci": (lambda: fibonacci(5), {}), "mixed": (mixed_program, {}), "memory": (memory_program, {}), "hash": (hash_program, {}),
            "call": (call_program, {}), "tape": (tape_program, {}), "storage": (storage_program, {"prove_program_hash": True}),
            "heap": (heap_program, {}), "storage_heavy": (lambda: storage_heavy_program(3, 5), {"prove_program_hash": True})}
