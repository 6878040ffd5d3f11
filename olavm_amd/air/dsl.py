"""Constraint DSL -> AIR-set blob.

The reference writes each table's constraints as Rust generic code over `PackedField`
(`Stark::eval_packed_generic`, circuits/src/stark/stark.rs:24-40) feeding a `ConstraintConsumer`
(circuits/src/stark/constraint_consumer.rs:34-78).  Here the same bodies are written over `Expr` objects; `compile()`
lowers the DAG (with common-subexpression elimination and linear-scan register allocation) to a straight-line program

    op word0 = opcode | kind << 8 | dst << 16 | a << 32 | b << 48 ;  word1 = immediate

    LOCAL dst, col      dst = local_values[col]           NEXT dst, col    dst = next_values[col]
    CONST dst, imm      dst = imm (canonical)             PARAM dst, idx   dst = per-proof table parameter idx
    ADD/SUB/MUL dst,a,b                                   EMIT kind, a     consumer.<kind>(reg a)
    ISZERO dst, a       dst = (a == 0) ? 1 : 0   (the data-dependent indicator of MemoryStark, memory_stark.rs:290-298)
        kind: 0 constraint, 1 constraint_transition, 2 constraint_first_row, 3 constraint_last_row

and the whole configuration (tables, permutation pairs, cross-table lookups) to one u64 array:

    [MAGIC, VERSION, num_tables, num_ctls]
    table*: [ncols, constraint_degree, n_regs, n_params, n_perm_pairs, (len, (lhs, rhs)*len)*, n_ops, (w0, w1)*n_ops]
    ctl*:   [n_looking, twc*(n_looking), twc(looked)]
    twc:    [table, n_columns, column*, has_filter, column?]     column: [n_terms, (col, coeff)*, constant]
"""
import numpy as np

P = 0xFFFFFFFF00000001
MAGIC = 0x4F4C41414952  # "OLAAIR"
VERSION = 1

OP_LOCAL, OP_NEXT, OP_CONST, OP_PARAM, OP_ADD, OP_SUB, OP_MUL, OP_EMIT, OP_ISZERO = range(9)
KIND_ALL, KIND_TRANSITION, KIND_FIRST, KIND_LAST = range(4)


class Expr:
    __slots__ = ("t", "key", "id")

    def __init__(self, t, key, id_):
        self.t, self.key, self.id = t, key, id_

    def _b(self):
        return self.t

    def __add__(self, o): return self.t.binop(OP_ADD, self, o)
    def __radd__(self, o): return self.t.binop(OP_ADD, o, self)
    def __sub__(self, o): return self.t.binop(OP_SUB, self, o)
    def __rsub__(self, o): return self.t.binop(OP_SUB, o, self)
    def __mul__(self, o): return self.t.binop(OP_MUL, self, o)
    def __rmul__(self, o): return self.t.binop(OP_MUL, o, self)
    def __neg__(self): return self.t.binop(OP_SUB, 0, self)

    def square(self): return self * self

    def exp(self, e):
        r, b = self.t.const(1), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r


class AirTable:
    """One STARK table (the `Stark` trait impl of the reference): columns, degree, permutation pairs, constraints."""

    def __init__(self, name, ncols, constraint_degree, n_params=0):
        self.name, self.ncols, self.constraint_degree, self.n_params = name, ncols, constraint_degree, n_params
        self.nodes = []        # (op, a, b) with a/b node ids, or leaf payload
        self.cse = {}
        self.emits = []        # (kind, node id)
        self.permutation_pairs = []  # list of lists of (lhs, rhs)

    # ---- leaves ----
    def _node(self, key):
        if key in self.cse:
            return self.cse[key]
        e = Expr(self, key, len(self.nodes))
        self.nodes.append(key)
        self.cse[key] = e
        return e

    def local(self, c):
        assert 0 <= c < self.ncols
        return self._node((OP_LOCAL, c, 0))

    def next(self, c):
        assert 0 <= c < self.ncols
        return self._node((OP_NEXT, c, 0))

    def const(self, v):
        return self._node((OP_CONST, int(v) % P, 0))

    def param(self, i):
        assert 0 <= i < self.n_params
        return self._node((OP_PARAM, i, 0))

    def is_zero(self, e):
        return self._node((OP_ISZERO, self.lift(e).id, 0))

    def lift(self, x):
        return x if isinstance(x, Expr) else self.const(x)

    def binop(self, op, a, b):
        a, b = self.lift(a), self.lift(b)
        if op in (OP_ADD, OP_MUL) and a.id > b.id:  # commutative: canonical order for CSE
            a, b = b, a
        return self._node((op, a.id, b.id))

    # ---- consumer (constraint_consumer.rs:57-78) ----
    def constraint(self, e): self.emits.append((KIND_ALL, self.lift(e).id))
    def constraint_transition(self, e): self.emits.append((KIND_TRANSITION, self.lift(e).id))
    def constraint_first_row(self, e): self.emits.append((KIND_FIRST, self.lift(e).id))
    def constraint_last_row(self, e): self.emits.append((KIND_LAST, self.lift(e).id))

    # lookup.rs:13-34
    def eval_lookups(self, col_permuted_input, col_permuted_table):
        local_perm_input = self.local(col_permuted_input)
        next_perm_table = self.next(col_permuted_table)
        next_perm_input = self.next(col_permuted_input)
        diff_input_prev = next_perm_input - local_perm_input
        diff_input_table = next_perm_input - next_perm_table
        self.constraint(diff_input_prev * diff_input_table)
        self.constraint_last_row(diff_input_table)

    def permutation_pair(self, pairs):
        self.permutation_pairs.append([(int(a), int(b)) for a, b in pairs])

    # ---- derived quantities (stark.rs:76-84, 218-245) ----
    @property
    def quotient_degree_factor(self):
        return max(1, self.constraint_degree - 1)

    def num_permutation_batches(self, num_challenges=2):
        inst = len(self.permutation_pairs) * num_challenges
        return -(-inst // self.quotient_degree_factor) if inst else 0

    # ---- lowering ----
    def schedule(self):
        """-> sequence of ('node', id) / ('emit', kind, id): emits in program order, every node right before its first
        use."""
        nodes, emits = self.nodes, self.emits
        order = []
        done = set()

        def visit(i):
            stack = [i]
            while stack:
                j = stack[-1]
                if j in done:
                    stack.pop()
                    continue
                op, a, b = nodes[j]
                deps = [a, b] if op in (OP_ADD, OP_SUB, OP_MUL) else ([a] if op == OP_ISZERO else [])
                pend = [d for d in deps if d not in done]
                if pend:
                    stack.extend(pend)
                    continue
                done.add(j)
                order.append(("node", j))
                stack.pop()

        for kind, i in emits:
            visit(i)
            order.append(("emit", kind, i))
        return order

    def compile(self):
        """-> (ops as list of (w0, w1), n_regs): the schedule lowered to the register machine the interpreter kernel
        runs; registers are recycled after their last use."""
        nodes = self.nodes
        order = self.schedule()
        last_use = {}
        for pos, it in enumerate(order):
            if it[0] == "node":
                op, a, b = nodes[it[1]]
                if op in (OP_ADD, OP_SUB, OP_MUL):
                    last_use[a] = pos
                    last_use[b] = pos
                elif op == OP_ISZERO:
                    last_use[a] = pos
            else:
                last_use[it[2]] = pos
        free, reg, n_regs, ops = [], {}, 0, []
        for pos, it in enumerate(order):
            if it[0] == "node":
                j = it[1]
                op, a, b = nodes[j]
                srcs = [a, b] if op in (OP_ADD, OP_SUB, OP_MUL) else ([a] if op == OP_ISZERO else [])
                ra = reg[a] if srcs else a
                rb = reg[b] if len(srcs) == 2 else 0
                # release sources whose last use is here (dst may reuse them)
                for s in set(srcs):
                    if last_use.get(s) == pos:
                        free.append(reg[s])
                if free:
                    d = free.pop()
                else:
                    d = n_regs
                    n_regs += 1
                reg[j] = d
                if op == OP_CONST:
                    ops.append((op | (d << 16), ra))
                elif op in (OP_LOCAL, OP_NEXT, OP_PARAM):
                    ops.append((op | (d << 16) | (ra << 32), 0))
                else:
                    ops.append((op | (d << 16) | (ra << 32) | (rb << 48), 0))
                if last_use.get(j) is None:  # dead value
                    free.append(d)
            else:
                _, kind, i = it
                ops.append((OP_EMIT | (kind << 8) | (reg[i] << 32), 0))
                if last_use.get(i) == pos:
                    free.append(reg[i])
        assert n_regs < 65536
        return ops, max(n_regs, 1)

    def words(self):
        ops, n_regs = self.compile()
        w = [self.ncols, self.constraint_degree, n_regs, self.n_params, len(self.permutation_pairs)]
        for pair in self.permutation_pairs:
            w.append(len(pair))
            for a, b in pair:
                w += [a, b]
        w.append(len(ops))
        for w0, w1 in ops:
            w += [w0, w1]
        return w


class Col:
    """Linear combination of columns (cross_table_lookup.rs:27-112)."""

    def __init__(self, terms, constant=0):
        self.terms = [(int(c), int(f) % P) for c, f in terms]
        self.constant = int(constant) % P

    @staticmethod
    def single(c): return Col([(c, 1)])
    @staticmethod
    def singles(cs): return [Col.single(c) for c in cs]
    @staticmethod
    def constant_(v): return Col([], v)
    @staticmethod
    def zero(): return Col([], 0)
    @staticmethod
    def one(): return Col([], 1)
    @staticmethod
    def linear_combination(terms, constant=0): return Col(list(terms), constant)
    @staticmethod
    def le_bits(cs): return Col([(c, pow(2, i, P)) for i, c in enumerate(cs)])
    @staticmethod
    def sum(cs): return Col([(c, 1) for c in cs])

    def words(self):
        w = [len(self.terms)]
        for c, f in self.terms:
            w += [c, f]
        w.append(self.constant)
        return w


class TableWithColumns:
    def __init__(self, table, columns, filter_column=None):
        self.table, self.columns, self.filter_column = int(table), list(columns), filter_column

    def words(self):
        w = [self.table, len(self.columns)]
        for c in self.columns:
            w += c.words()
        if self.filter_column is None:
            w.append(0)
        else:
            w.append(1)
            w += self.filter_column.words()
        return w


class CrossTableLookup:
    def __init__(self, looking_tables, looked_table):
        assert all(len(t.columns) == len(looked_table.columns) for t in looking_tables)
        self.looking_tables, self.looked_table = list(looking_tables), looked_table

    def words(self):
        w = [len(self.looking_tables)]
        for t in self.looking_tables:
            w += t.words()
        w += self.looked_table.words()
        return w


class AirSet:
    """A multi-table STARK configuration (the reference's `OlaStark`, stark/ola_stark.rs:29-64)."""

    def __init__(self, tables, ctls):
        self.tables, self.ctls = list(tables), list(ctls)

    def blob(self):
        w = [MAGIC, VERSION, len(self.tables), len(self.ctls)]
        for t in self.tables:
            w += t.words()
        for c in self.ctls:
            w += c.words()
        return np.array(w, dtype=np.uint64)

    def ctl_jobs(self, table, num_challenges=2):
        """The TableWithColumns behind each CTL Z column of `table`, in the order the prover creates them
        (cross_table_lookup.rs:224-282: per lookup, per challenge, looking tables then the looked table)."""
        jobs = []
        for c in self.ctls:
            for _ in range(num_challenges):
                jobs += [t for t in c.looking_tables + [c.looked_table] if t.table == table]
        return jobs

    def signature_words(self, table):
        """What a specialised quotient kernel depends on: the table's description and the static part of its CTL jobs.
        Hashed (FNV-1a over the little-endian bytes) on both sides of the C ABI to pair a blob with a generated kernel."""
        w = list(self.tables[table].words())
        jobs = self.ctl_jobs(table)
        w.append(len(jobs))
        for j in jobs:
            w += j.words()[1:]          # without the table index
        return w

    def signature(self, table):
        h = 0xCBF29CE484222325
        for x in self.signature_words(table):
            for k in range(8):
                h = ((h ^ ((int(x) >> (8 * k)) & 0xFF)) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h

    def num_ctl_zs(self, table, num_challenges=2):
        n = 0
        for c in self.ctls:
            n += sum(1 for t in c.looking_tables + [c.looked_table] if t.table == table)
        return n * num_challenges
