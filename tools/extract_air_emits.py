#!/usr/bin/env python3
"""Which constraints the reference emits, in which order and of which kind -- read off its Rust SOURCE, not off this repo's transcription.

The proof bytes depend on every emitted constraint's position (the alpha power it is combined with) and kind (`constraint`,
`constraint_transition`, `constraint_first_row`, `constraint_last_row`: constraint_consumer.rs:34-78).  olavm_amd/air/ola_tables.py
restates the twelve AIRs by hand; oracle and GPU both consume that restatement (include/ola_airset.bin), so a miscounted loop or a
swapped pair of constraints would be common mode.  This script walks `eval_packed_generic` of the twelve `*_stark.rs` files (and the
functions they call: the `cpu/*.rs` opcode files, `lookup.rs`) as a Rust-subset interpreter that only cares about control flow:

  * `yield_constr.constraint*( ... )`  -> one emit of that kind;
  * `for PAT in ITER { ... }`, `ITER.for_each(|..| ...)`: the body once per item, where the item count of ITER is evaluated from the
    source's own constants (`const X: usize = ..`, `const R: Range<usize> = a..b`), array types (`[P; REGISTER_NUM]`), array
    literals, slices (`&x[..N - 1]`, `lv[RANGE]`) and the adaptors `iter / rev / enumerate / skip / take / zip / izip! / map / ...`;
  * `if COND { ... }` with COND over constants and range-loop variables (`if r != 0`);
  * calls that pass the consumer on (`Self::constraint_ext_lines(&wrapper, yield_constr)`, `mov::eval_packed_generic(lv, nv, yield_constr)`,
    `eval_lookups(vars, yield_constr, a, b)`) -> the callee's body, found by name in the reference tree.

Of the arithmetic inside a constraint one thing is read: the trace cells its argument names DIRECTLY (`lv[COL_X]`,
`wrapper.nv[COL_Y.start + i]`, indices evaluated from constants and loop variables) -- a subset of the cells the constraint reads
(values that arrive through local variables are not followed), compared with the transcription's expression for the same emit.
Everything else (the arithmetic itself) is skipped: it is pinned by other means (golden rows, executed programs, the
verifier's extension-field evaluation; PARITY.md).  Per table the script also reads `COLUMNS`, `constraint_degree()`, the number of
`PermutationPair`s, and from stark/ola_stark.rs:122-560 the cross-table lookups: per CTL the looked table and how many looking
entries each table contributes.

    python tools/extract_air_emits.py [--reference /root/reference] [--out tests/golden/air_emit_kinds.json] [--check]

Runs in the build container only (it reads /root/reference); the fixture it writes is data: lists of kinds and counts.  A construct
the walker does not understand raises with file:line -- nothing is guessed.
"""
import argparse
import json
import os
import re
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "air_emit_kinds.json")

# enum Table (stark/ola_stark.rs:103-118) -> the file whose `impl Stark` is walked
TABLES = [("Cpu", "cpu/cpu_stark.rs"), ("Memory", "memory/memory_stark.rs"), ("Bitwise", "builtins/bitwise/bitwise_stark.rs"),
          ("Cmp", "builtins/cmp/cmp_stark.rs"), ("RangeCheck", "builtins/rangecheck/rangecheck_stark.rs"),
          ("Poseidon", "builtins/poseidon/poseidon_stark.rs"), ("PoseidonChunk", "builtins/poseidon/poseidon_chunk_stark.rs"),
          ("StorageAccess", "builtins/storage/storage_access_stark.rs"), ("Tape", "builtins/tape/tape_stark.rs"),
          ("SCCall", "builtins/sccall/sccall_stark.rs"), ("Program", "program/program_stark.rs"), ("ProgChunk", "program/prog_chunk_stark.rs")]
KINDS = {"constraint": "all", "constraint_transition": "transition", "constraint_first_row": "first_row", "constraint_last_row": "last_row"}

TOKEN = re.compile(r"""\s+|//[^\n]*|/\*.*?\*/|(?P<str>b?"(?:\\.|[^"\\])*")|(?P<chr>'(?:\\.|[^'\\])')|(?P<life>'[A-Za-z_]\w*)|(?P<num>0x[0-9a-fA-F_]+(?:_?[ui]\d+|usize)?|0b[01_]+(?:_?[ui]\d+|usize)?|\d[\d_]*(?:\.\d+)?(?:_?[uif]\d+|_?usize|_?isize)?)"""
                   r"""|(?P<id>[A-Za-z_]\w*!?)|(?P<op>\.\.=|\.\.\.|\.\.|::|->|=>|==|!=|<=|>=|&&|\|\||<<|>>|\+=|-=|\*=|/=|[-+*/%=<>!&|^~.,;:#?@$(){}\[\]])""", re.S)


class Unsupported(Exception):
    pass


class Src:
    """One reference file as a token list [(text, line)]."""
    cache = {}

    def __init__(self, path):
        self.path = path
        text = open(path, errors="replace").read()
        self.toks, pos, line = [], 0, 1
        while pos < len(text):
            m = TOKEN.match(text, pos)
            if not m:
                raise Unsupported(f"{path}:{line}: cannot tokenise {text[pos:pos + 20]!r}")
            if m.lastgroup:
                self.toks.append((m.group(m.lastgroup), line))
            line += text.count("\n", pos, m.end())
            pos = m.end()

    @classmethod
    def get(cls, path):
        path = os.path.realpath(path)
        if path not in cls.cache:
            cls.cache[path] = Src(path)
        return cls.cache[path]

    def match(self, i):
        """index of the bracket closing the one at i"""
        pairs = {"(": ")", "[": "]", "{": "}"}
        open_, close, depth = self.toks[i][0], pairs[self.toks[i][0]], 0
        for j in range(i, len(self.toks)):
            t = self.toks[j][0]
            if t == open_:
                depth += 1
            elif t == close:
                depth -= 1
                if depth == 0:
                    return j
        raise Unsupported(f"{self.path}:{self.toks[i][1]}: unbalanced {open_}")

    def functions(self):
        """name -> (params (lo, hi), body (lo, hi)) for every `fn` with a body"""
        if getattr(self, "_functions", None) is not None:
            return self._functions
        out = {}
        for i, (t, _) in enumerate(self.toks):
            if t == "fn" and i + 1 < len(self.toks):
                name = self.toks[i + 1][0]
                j = i + 2
                if self.toks[j][0] == "<":          # generics may contain parentheses: `<I: IntoIterator<Item = (usize, F)>>`
                    depth = 0
                    while True:
                        t = self.toks[j][0]
                        depth += {"<": 1, ">": -1, "<<": 2, ">>": -2}.get(t, 0)
                        j += 1
                        if depth <= 0:
                            break
                while self.toks[j][0] != "(":
                    j += 1
                pe = self.match(j)
                k = pe + 1
                while self.toks[k][0] not in ("{", ";"):
                    k = self.match(k) + 1 if self.toks[k][0] in ("[", "(") else k + 1      # `-> [P; N]`: the `;` of an array type is not the end
                if self.toks[k][0] == "{":
                    out.setdefault(name, ((j, pe), (k, self.match(k))))
        self._functions = out
        return out


class Ref:
    """The reference tree: constants, functions, structs -- looked up by name, nearest file first."""

    def __init__(self, root):
        self.root = os.path.join(root, "circuits", "src")
        self.files = []
        for base in (self.root, os.path.join(root, "core", "src"), os.path.join(root, "plonky2", "plonky2", "src", "hash")):
            for d, _, fs in os.walk(base):
                self.files += [os.path.join(d, f) for f in sorted(fs) if f.endswith(".rs")]
        self.values = {}          # (name, file) -> evaluated constant
        self.consts = {}          # name -> [(file, expression tokens)]
        self.const_files = set()
        for f in self.files:
            self.index_consts(f)

    def index_consts(self, f):
        """the `const NAME: T = expr;` items of one file (also called for files outside the AIR tree a driver adds later)"""
        if f in self.const_files:
            return
        self.const_files.add(f)
        if True:
            s = Src.get(f)
            T = s.toks
            for i, (t, _) in enumerate(T):
                if t == "const" and i + 2 < len(T) and T[i + 2][0] == ":" and re.match(r"^[A-Z_][A-Z0-9_]*$", T[i + 1][0]):
                    j = i + 3
                    depth = 0
                    while j < len(T) and not (T[j][0] == "=" and depth == 0):
                        if depth == 0 and T[j][0] in (",", ">", ")", "{", ";"):
                            j = len(T)            # a const generic parameter, not an item
                            break
                        depth += T[j][0] in "<[(" and 1 or 0
                        depth -= T[j][0] in ">])" and 1 or 0
                        j += 1
                    if j >= len(T):
                        continue
                    k = j + 1
                    while T[k][0] != ";":
                        k = s.match(k) if T[k][0] in "([{" else k
                        k += 1
                    self.consts.setdefault(T[i + 1][0], []).append((f, T[j + 1:k]))

    def near(self, cands, here):
        """the candidate file closest to `here` (same file, same directory, then a unique one)"""
        here = os.path.realpath(here)
        same = [c for c in cands if os.path.realpath(c[0]) == here]
        if same:
            return same[0]
        d = os.path.dirname(here)
        sib = [c for c in cands if os.path.dirname(os.path.realpath(c[0])) == d]
        if len(sib) == 1:
            return sib[0]
        circ = [c for c in cands if os.path.realpath(c[0]).startswith(os.path.realpath(self.root))] or cands
        vals = {json.dumps([t for t, _ in c[1]]) if isinstance(c[1], list) else c[1] for c in circ}
        if len(vals) == 1:
            return circ[0]
        return None

    def find_fn(self, name, here, module=None):
        cands = []
        for f in self.files:
            if module and os.path.basename(f) != module + ".rs" and os.path.basename(os.path.dirname(f)) + "/mod.rs" != module + "/mod.rs":
                continue
            fns = Src.get(f).functions()
            if name in fns:
                cands.append((f, name))
        if not cands:
            return None
        c = self.near(cands, here)
        return c[0] if c else None


class R:
    """a Rust Range<usize>"""

    def __init__(self, a, b):
        self.start, self.end = a, b

    def __len__(self):
        return max(0, self.end - self.start)


class Items:
    """what an iterator yields: how many, and -- for ranges -- which integers"""

    def __init__(self, n, ints=None):
        self.n, self.ints = n, ints


class Walker:
    def __init__(self, ref, emit_names=KINDS, consumer_type="ConstraintConsumer"):
        self.ref, self.emit_names, self.consumer_type = ref, emit_names, consumer_type
        self.emits = []
        self.stack = []

    # ---------------------------------------------------------------- constant expressions
    def const(self, name, src):
        cands = self.ref.consts.get(name)
        if not cands:
            raise Unsupported(f"{src.path}: constant {name} not found")
        c = self.ref.near(cands, src.path)
        if c is None:
            raise Unsupported(f"{src.path}: constant {name} is defined differently in {[x[0] for x in cands]}")
        key = (name, c[0])
        if key not in self.ref.values:
            self.ref.values[key] = self.expr(c[1], Src.get(c[0]), {})
        return self.ref.values[key]

    def expr(self, toks, src, env):
        """value of a constant expression (int, bool or R) given as [(text, line)]"""
        pos = [0]
        T = [t for t, _ in toks]
        line = toks[0][1] if toks else 0

        def peek():
            return T[pos[0]] if pos[0] < len(T) else None

        def take(x=None):
            t = peek()
            if x is not None and t != x:
                raise Unsupported(f"{src.path}:{line}: expected {x} in {' '.join(T)}")
            pos[0] += 1
            return t

        def primary():
            t = take()
            if t is None:
                raise Unsupported(f"{src.path}:{line}: truncated expression {' '.join(T)}")
            if t == "(":
                v = rng()
                take(")")
            elif t == "-":
                return -primary()
            elif t == "!":
                return not primary()
            elif t in ("&", "*"):
                return primary()
            elif re.match(r"^\d", t):
                v = int(re.sub(r"_?(?:[uif]\d+|usize|isize)$", "", t).replace("_", ""), 0)
            elif re.match(r"^[A-Za-z_]", t):
                path = [t]
                while peek() == "::":
                    take()
                    if peek() == "<":          # turbofish
                        depth = 0
                        while True:
                            x = take()
                            depth += x == "<"
                            depth -= x == ">"
                            if depth == 0:
                                break
                        continue
                    path.append(take())
                name = path[-1]
                if name in env and len(path) == 1:
                    v = env[name]
                elif name in ("true", "false"):
                    v = name == "true"
                elif re.match(r"^[A-Z_][A-Z0-9_]*$", name):
                    v = self.const(name, src)
                else:
                    raise Unsupported(f"{src.path}:{line}: `{'::'.join(path)}` is not a constant (in {' '.join(T)})")
            else:
                raise Unsupported(f"{src.path}:{line}: unexpected `{t}` in {' '.join(T)}")
            while True:
                if peek() == "." and pos[0] + 1 < len(T) and T[pos[0] + 1] in ("start", "end"):
                    take()
                    v = getattr(v, take())
                elif peek() == "." and pos[0] + 1 < len(T) and T[pos[0] + 1] == "len":
                    take(); take(); take("("); take(")")
                    v = len(v)
                elif peek() == "." and pos[0] + 1 < len(T) and T[pos[0] + 1] == "pow":
                    take(); take(); take("(")
                    e = rng()
                    take(")")
                    v = v ** e
                elif peek() == "as":
                    take(); take()
                else:
                    return v

        def binary(level):
            ops = [["||"], ["&&"], ["==", "!=", "<", ">", "<=", ">="], ["|"], ["^"], ["&"], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]
            if level == len(ops):
                return primary()
            v = binary(level + 1)
            while peek() in ops[level]:
                o = take()
                w = binary(level + 1)
                v = {"||": lambda: v or w, "&&": lambda: v and w, "==": lambda: v == w, "!=": lambda: v != w, "<": lambda: v < w, ">": lambda: v > w,
                     "<=": lambda: v <= w, ">=": lambda: v >= w, "|": lambda: v | w, "^": lambda: v ^ w, "&": lambda: v & w, "<<": lambda: v << w,
                     ">>": lambda: v >> w, "+": lambda: v + w, "-": lambda: v - w, "*": lambda: v * w, "/": lambda: v // w, "%": lambda: v % w}[o]()
            return v

        def rng():
            a = None if peek() in ("..", "..=") else binary(0)
            if peek() in ("..", "..="):
                inc = take() == "..="
                b = None if peek() in (None, ")", "]", ",") else binary(0)
                if a is None or b is None:
                    return ("open", a, b, inc)
                return R(a, b + 1 if inc else b)
            return a

        v = rng()
        if pos[0] != len(T):
            raise Unsupported(f"{src.path}:{line}: cannot evaluate `{' '.join(T)}` (stopped at `{peek()}`)")
        return v

    # ---------------------------------------------------------------- iterators
    def split(self, toks, sep=","):
        """top-level split of a token list"""
        out, cur, depth = [], [], 0
        for t in toks:
            if t[0] in "([{":
                depth += 1
            elif t[0] in ")]}":
                depth -= 1
            if t[0] == sep and depth == 0:
                out.append(cur)
                cur = []
            else:
                cur.append(t)
        if cur:
            out.append(cur)
        return out

    def length_of_type(self, toks, src, env):
        """[T; N] (possibly behind & / mut) -> N"""
        T = [t for t in toks if t[0] not in ("&", "mut")]
        if T and T[0][0] == "[" and T[-1][0] == "]":
            parts = self.split(T[1:-1], ";")
            if len(parts) == 2:
                return self.expr(parts[1], src, env)
        return None

    def local_len(self, name, src, fn_range, env, upto):
        """length of the local array / slice `name`, from its `let` (type annotation or initialiser) before token `upto`"""
        T = src.toks
        lo = fn_range[0]
        best = None
        for i in range(lo, upto):
            if T[i][0] == "let":
                j = i + 1
                if T[j][0] == "mut":
                    j += 1
                if T[j][0] == name and T[j + 1][0] in (":", "="):
                    best = j
        if best is None:
            return None
        j = best + 1
        ty = None
        if T[j][0] == ":":
            k = j + 1
            depth = 0
            while not (T[k][0] == "=" and depth == 0):
                depth += T[k][0] in "([<"
                depth -= T[k][0] in ")]>"
                k += 1
            ty = T[j + 1:k]
            j = k
        if ty:
            n = self.length_of_type(ty, src, env)
            if n is not None:
                return Items(n)
        k = j + 1
        while T[k][0] != ";":
            k = src.match(k) if T[k][0] in "([{" else k
            k += 1
        return self.items(T[j + 1:k], src, fn_range, env, best)

    def field_len(self, field, src, env):
        """`x.field` where some struct of this file declares `field: [T; N]`"""
        T = src.toks
        for i in range(len(T) - 2):
            if T[i][0] == field and T[i + 1][0] == ":" and T[i + 2][0] in ("[", "&"):
                k = i + 2
                while T[k][0] != "[":
                    k += 1
                n = self.length_of_type(T[k:src.match(k) + 1], src, env)
                if n is not None:
                    return Items(n)
        return None

    def items(self, toks, src, fn_range, env, upto):
        """what the iterator expression `toks` yields"""
        toks = list(toks)
        line = toks[0][1]
        text = " ".join(t for t, _ in toks)
        while toks and toks[0][0] in ("&", "*", "mut"):
            toks = toks[1:]
        # postfix chain: primary (. method ( args ))*
        i = 0
        if toks[0][0] == "izip!":
            e = self.match_in(toks, 1)
            cur = Items(min(self.items(a, src, fn_range, env, upto).n for a in self.split(toks[2:e])))
            i = e + 1
        elif toks[0][0] == "(":
            e = self.match_in(toks, 0)
            cur = self.items(toks[1:e], src, fn_range, env, upto)
            i = e + 1
        elif toks[0][0] == "[":
            e = self.match_in(toks, 0)
            inner = toks[1:e]
            parts = self.split(inner, ";")
            cur = Items(self.expr(parts[1], src, env)) if len(parts) == 2 else Items(len(self.split(inner)))
            i = e + 1
        elif toks[0][0] == "vec!":
            e = self.match_in(toks, 1)
            cur = Items(len(self.split(toks[2:e])))
            i = e + 1
        else:
            # a path / identifier / field access, possibly a range expression: take everything up to the first `.method(` or `[`
            j, depth = 0, 0
            while j < len(toks):
                t = toks[j][0]
                if depth == 0 and t == "[":
                    break
                if depth == 0 and t == "." and j + 2 < len(toks) and toks[j + 2][0] == "(" and toks[j + 1][0] not in ("start", "end"):
                    break
                if depth == 0 and t == "." and j + 1 < len(toks) and re.match(r"^[a-z_]", toks[j + 1][0]) and toks[j + 1][0] not in ("start", "end") and \
                        not (j + 2 < len(toks) and toks[j + 2][0] == "("):
                    j += 2          # field access a.b
                    continue
                depth += t in "(["
                depth -= t in ")]"
                j += 1
            head = toks[:j]
            i = j
            names = [t for t, _ in head]
            if any(t in ("..", "..=") for t in names) or all(re.match(r"^[A-Z_0-9:]+$|^::$|^[+\-*/()]$|^\d", t) or t in ("start", "end", ".", "Self") for t in names):
                v = self.expr(head, src, env)
                if isinstance(v, R):
                    cur = Items(len(v), list(range(v.start, v.end)))
                else:
                    raise Unsupported(f"{src.path}:{line}: `{text}` is not a range")
            else:
                base = names[-1]
                cur = None
                if len(names) == 1:
                    try:
                        cur = self.local_len(base, src, fn_range, env, upto)
                    except Unsupported:
                        if not (i < len(toks) and toks[i][0] == "["):
                            raise
                        cur = None         # `let nv = vars.next_values;` -- a row: only its slices have a known length
                if cur is None and len(names) >= 3 and names[-2] == ".":
                    cur = self.field_len(base, src, env)
                if cur is None and i < len(toks) and toks[i][0] == "[":
                    cur = Items(None)      # a row (`lv`, `vars.local_values`): only its slices have a known length
                if cur is None:
                    raise Unsupported(f"{src.path}:{line}: do not know how many items `{text}` has")
        while i < len(toks):
            t = toks[i][0]
            if t == "[":
                e = self.match_in(toks, i)
                v = self.expr(toks[i + 1:e], src, env)
                if isinstance(v, R):
                    cur = Items(len(v))
                elif isinstance(v, tuple) and v[0] == "open":
                    _, a, b, inc = v
                    if cur.n is None and b is None:
                        raise Unsupported(f"{src.path}:{line}: open slice of a row in `{text}`")
                    hi = (b + 1 if inc else b) if b is not None else cur.n
                    cur = Items(hi - (a or 0))
                else:
                    raise Unsupported(f"{src.path}:{line}: element access in iterator `{text}`")
                i = e + 1
            elif t == "." and i + 1 < len(toks):
                m = toks[i + 1][0]
                if i + 2 < len(toks) and toks[i + 2][0] == "::":      # collect::<Vec<_>>()
                    k = i + 3
                    depth = 0
                    while True:
                        depth += toks[k][0] == "<"
                        depth -= toks[k][0] == ">"
                        k += 1
                        if depth == 0:
                            break
                    args_lo = k
                else:
                    args_lo = i + 2
                if args_lo >= len(toks) or toks[args_lo][0] != "(":
                    raise Unsupported(f"{src.path}:{line}: field `{m}` in iterator `{text}`")
                e = self.match_in(toks, args_lo)
                args = toks[args_lo + 1:e]
                if cur.n is None:
                    raise Unsupported(f"{src.path}:{line}: iterating a whole row in `{text}`")
                if m in ("iter", "into_iter", "iter_mut", "enumerate", "copied", "cloned", "map", "by_ref", "try_into", "unwrap", "collect", "to_vec", "into_par_iter", "par_iter"):
                    if m == "enumerate":
                        cur = Items(cur.n, list(range(cur.n)))
                elif m == "rev":
                    cur = Items(cur.n, cur.ints[::-1] if cur.ints else None)
                elif m == "skip":
                    k = self.expr(args, src, env)
                    cur = Items(max(0, cur.n - k), cur.ints[k:] if cur.ints else None)
                elif m == "take":
                    k = self.expr(args, src, env)
                    cur = Items(min(cur.n, k), cur.ints[:k] if cur.ints else None)
                elif m == "step_by":
                    k = self.expr(args, src, env)
                    cur = Items(-(-cur.n // k), cur.ints[::k] if cur.ints else None)
                elif m == "zip":
                    cur = Items(min(cur.n, self.items(args, src, fn_range, env, upto).n))
                elif m == "chain":
                    cur = Items(cur.n + self.items(args, src, fn_range, env, upto).n)
                else:
                    raise Unsupported(f"{src.path}:{line}: iterator adaptor `.{m}()` in `{text}`")
                i = e + 1
            else:
                raise Unsupported(f"{src.path}:{line}: cannot read iterator `{text}` at `{t}`")
        if cur.n is None:
            raise Unsupported(f"{src.path}:{line}: iterating a whole row in `{text}`")
        return cur

    @staticmethod
    def match_in(toks, i):
        pairs = {"(": ")", "[": "]", "{": "}"}
        o, c, depth = toks[i][0], pairs[toks[i][0]], 0
        for j in range(i, len(toks)):
            depth += toks[j][0] == o
            depth -= toks[j][0] == c
            if depth == 0:
                return j
        raise Unsupported(f"unbalanced {o} at line {toks[i][1]}")

    # ---------------------------------------------------------------- control flow
    def consumer_of(self, src, params):
        """name of the parameter whose type mentions the consumer"""
        T = src.toks
        for part in self.split(T[params[0] + 1:params[1]]):
            names = [t for t, _ in part]
            if self.consumer_type in names and ":" in names:
                return names[names.index(":") - 1]
        return None

    def call(self, path, fn_name):
        src = Src.get(path)
        fns = src.functions()
        if fn_name not in fns:
            raise Unsupported(f"{path}: fn {fn_name} not found")
        params, body = fns[fn_name]
        consumer = self.consumer_of(src, params)
        key = (os.path.realpath(path), fn_name)
        if key in self.stack:
            raise Unsupported(f"{path}: recursion into {fn_name}")
        self.stack.append(key)
        self.block(src, body[0] + 1, body[1], {}, (body[0], body[1]), consumer)
        self.stack.pop()

    def has_emit(self, src, lo, hi, consumer):
        T = src.toks
        return any(T[i][0] == consumer for i in range(lo, hi))

    def block(self, src, lo, hi, env, fn_range, consumer):
        T = src.toks
        i = lo
        stmt = lo          # first token of the current statement
        while i < hi:
            t, line = T[i]
            if t in (";",):
                i += 1
                stmt = i
            elif t == "for" and T[i + 1][0] != "<":
                j = i + 1
                while T[j][0] != "in":
                    j = src.match(j) if T[j][0] in "([" else j
                    j += 1
                pat = T[i + 1:j]
                k = j + 1
                while T[k][0] != "{":
                    k = src.match(k) if T[k][0] in "([" else k
                    k += 1
                e = src.match(k)
                if self.has_emit(src, k, e, consumer):
                    its = self.items(T[j + 1:k], src, fn_range, env, i)
                    names = [p for p, _ in pat if re.match(r"^[a-z_]\w*$", p) and p != "mut"]
                    for n in range(its.n):
                        env2 = dict(env)
                        if its.ints is not None and len(pat) == 1:
                            env2[names[0]] = its.ints[n]
                        elif its.ints is not None and pat[0][0] == "(" and names and [t for t, _ in T[j + 1:k]][-3:] == ["enumerate", "(", ")"]:
                            env2[names[0]] = its.ints[n]
                        self.block(src, k + 1, e, env2, fn_range, consumer)
                i = e + 1
                stmt = i
            elif t == "if":
                # if COND { A } [else if ... | else { B }]
                taken = False
                while True:
                    k = i + 1
                    while T[k][0] != "{":
                        k = src.match(k) if T[k][0] in "([" else k
                        k += 1
                    e = src.match(k)
                    chain_has_emit = self.has_emit(src, k, e, consumer)
                    cond = None
                    if T[i + 1][0] == "let":
                        if chain_has_emit:
                            raise Unsupported(f"{src.path}:{line}: `if let` around constraints")
                    elif chain_has_emit or not taken:
                        try:
                            cond = self.expr(T[i + 1:k], src, env)
                        except Unsupported:
                            if chain_has_emit:
                                raise
                    if cond and not taken:
                        taken = True
                        self.block(src, k + 1, e, env, fn_range, consumer)
                    i = e + 1
                    if i < hi and T[i][0] == "else":
                        if T[i + 1][0] == "if":
                            i += 1
                            continue
                        k = i + 1
                        e = src.match(k)
                        if not taken:
                            if cond is None and self.has_emit(src, k, e, consumer):
                                raise Unsupported(f"{src.path}:{line}: undecided `if` with constraints in its else branch")
                            self.block(src, k + 1, e, env, fn_range, consumer)
                        i = e + 1
                    break
                stmt = i
            elif t in ("match", "while", "loop"):
                k = i + 1
                while T[k][0] != "{":
                    k = src.match(k) if T[k][0] in "([" else k
                    k += 1
                e = src.match(k)
                if self.has_emit(src, i, e, consumer):
                    raise Unsupported(f"{src.path}:{line}: `{t}` around constraints")
                i = e + 1
            elif t == consumer and T[i + 1][0] == "." and T[i + 2][0] in self.emit_names and T[i + 3][0] == "(":
                e = src.match(i + 3)
                if self.has_emit(src, i + 4, e, consumer):
                    raise Unsupported(f"{src.path}:{line}: a constraint inside a constraint's argument")
                self.emits.append((self.emit_names[T[i + 2][0]], os.path.relpath(src.path, self.ref.root), line, self.direct_cells(src, i + 4, e, env, fn_range)))
                i = e + 1
            elif t == "." and T[i + 1][0] == "for_each" and T[i + 2][0] == "(":
                e = src.match(i + 2)
                if self.has_emit(src, i + 3, e, consumer):
                    its = self.items(T[stmt:i], src, fn_range, env, stmt)
                    k = i + 3
                    param = None
                    if T[k][0] == "move":
                        k += 1
                    if T[k][0] == "||":
                        k += 1
                    else:
                        assert T[k][0] == "|", f"{src.path}:{line}: closure expected"
                        k += 1
                        p0 = k
                        while T[k][0] != "|":
                            k = src.match(k) if T[k][0] in "([" else k
                            k += 1
                        if k - p0 == 1 and re.match(r"^[a-z_]\w*$", T[p0][0]):
                            param = T[p0][0]          # |col| over a range: the closure's argument is the integer
                        k += 1
                    for n in range(its.n):
                        env2 = dict(env)
                        if param and its.ints is not None:
                            env2[param] = its.ints[n]
                        if T[k][0] == "{":
                            self.block(src, k + 1, src.match(k), env2, fn_range, consumer)
                        else:
                            self.block(src, k, e, env2, fn_range, consumer)
                i = e + 1
            elif t == "|" and i > lo and T[i - 1][0] in ("(", ",", "=") and self.closure_with_emit(src, i, hi, consumer):
                raise Unsupported(f"{src.path}:{line}: a closure other than for_each's emits constraints")
            elif t == "(" and i > lo and re.match(r"^[A-Za-z_]\w*$", T[i - 1][0]) and T[i - 1][0] not in ("if", "for", "in", "return", "let", "mut", "as") and \
                    not (i - 2 >= lo and T[i - 2][0] == "." ):
                e = src.match(i)
                args = [x for x, _ in T[i + 1:e]]
                if consumer in args:
                    name = T[i - 1][0]
                    module = None
                    if i - 2 >= lo and T[i - 2][0] == "::":
                        q = T[i - 3][0]
                        module = None if q == "Self" else q
                    target = src.path if module is None and name in src.functions() else self.ref.find_fn(name, src.path, module)
                    if target is None:
                        raise Unsupported(f"{src.path}:{line}: callee `{(module + '::') if module else ''}{name}` not found")
                    self.call(target, name)
                    i = e + 1
                else:
                    i += 1
            elif t == "{":
                e = src.match(i)
                # a plain nested block or a struct literal: walk it (struct literals contain no constraints)
                self.block(src, i + 1, e, env, fn_range, consumer)
                i = e + 1
                if i < hi and T[i][0] != ";" and T[i][0] not in (".", ")", ",", "?"):
                    stmt = i
            else:
                i += 1

    ROWS = {"lv": "L", "local_values": "L", "nv": "N", "next_values": "N"}

    def direct_cells(self, src, lo, hi, env, fn_range=None, depth=0, upto=None):
        """Trace cells a constraint's argument names -- `lv[COL_X]`, `wrapper.nv[COL_Y.start + i]`, `vars.local_values[..]` with an index
        that evaluates from constants and loop variables -- directly or through immutable locals of the same function
        (`let lv_is_padding = lv[COL_IS_PADDING];`: the initialiser is scanned the same way).  Mutable variables, struct fields and
        function results are not followed: this is a subset of the cells the constraint reads, enough to catch a column mix-up."""
        T = src.toks
        out = set()
        upto = lo if upto is None else upto
        i = lo
        while i < hi:
            t = T[i][0]
            if t == "[" and i > lo and T[i - 1][0] in self.ROWS:
                e = src.match(i)
                try:
                    v = self.expr(T[i + 1:e], src, env)
                    if isinstance(v, int) and not isinstance(v, bool):
                        out.add(self.ROWS[T[i - 1][0]] + str(v))
                except Unsupported:
                    pass
            elif fn_range is not None and depth < 6 and re.match(r"^[a-z_]\w*$", t) and t not in self.ROWS and t not in env and \
                    T[i - 1][0] not in (".", "::") and T[i + 1][0] not in ("(", "::", "!", "["):
                # an immutable local of this function, defined before the use: follow its initialiser
                best = None
                for k in range(fn_range[0], upto):
                    if T[k][0] == "let" and T[k + 1][0] == t and T[k + 2][0] in ("=", ":"):
                        best = k
                if best is not None:
                    k = best + 2
                    while T[k][0] != "=":
                        k += 1
                    e = k + 1
                    while T[e][0] != ";":
                        e = src.match(e) if T[e][0] in "([{" else e
                        e += 1
                    out |= set(self.direct_cells(src, k + 1, e, env, fn_range, depth + 1, best))
            i += 1
        return sorted(out, key=lambda c: (c[0], int(c[1:])))

    def closure_with_emit(self, src, i, hi, consumer):
        """does the closure whose parameter list opens at i mention the consumer in its body?"""
        T = src.toks
        k = i + 1
        while k < hi and T[k][0] != "|":
            k += 1
        k += 1
        if k >= hi:
            return False
        if T[k][0] == "{":
            return self.has_emit(src, k, src.match(k), consumer)
        depth, j = 0, k
        while j < hi:
            if T[j][0] in "([{":
                depth += 1
            elif T[j][0] in ")]}":
                if depth == 0:
                    break
                depth -= 1
            elif T[j][0] in (",", ";") and depth == 0:
                break
            j += 1
        return self.has_emit(src, k, j, consumer)


def table_facts(ref, name, rel):
    path = os.path.join(ref.root, rel)
    src = Src.get(path)
    w = Walker(ref)
    T = src.toks
    # the trait impl's eval_packed_generic is the one that takes `&self` (cpu/*.rs have free functions of the same name)
    start = None
    for i, (t, _) in enumerate(T):
        if t == "fn" and T[i + 1][0] == "eval_packed_generic":
            j = i + 2
            while T[j][0] != "(":
                j += 1
            if T[j + 1][0] == "&" and T[j + 2][0] == "self":
                start = (j, src.match(j))
                break
    if start is None:
        raise Unsupported(f"{path}: no eval_packed_generic(&self, ...)")
    k = start[1] + 1
    while T[k][0] != "{":
        k += 1
    body = (k, src.match(k))
    consumer = w.consumer_of(src, start)
    w.stack.append((os.path.realpath(path), "eval_packed_generic"))
    w.block(src, body[0] + 1, body[1], {}, body, consumer)
    facts = {"file": rel, "emit_kinds": [k for k, _, _, _ in w.emits], "emit_sites": ["%s:%d" % (f, l) for _, f, l, _ in w.emits],
             "emit_direct_cells": [c for _, _, _, c in w.emits]}
    # COLUMNS, constraint_degree(), permutation pairs
    for i, (t, _) in enumerate(T):
        if t == "const" and T[i + 1][0] == "COLUMNS":
            j = i + 2
            while T[j][0] != "=":
                j += 1
            e = j
            while T[e][0] != ";":
                e += 1
            facts["columns"] = w.expr(T[j + 1:e], src, {})
    fns = src.functions()
    if "constraint_degree" in fns:
        lo, hi = fns["constraint_degree"][1]
        facts["constraint_degree"] = w.expr(T[lo + 1:hi], src, {})
    facts["permutation_pairs"] = 0
    if "permutation_pairs" in fns:
        lo, hi = fns["permutation_pairs"][1]
        facts["permutation_pairs"] = sum(1 for i in range(lo, hi) if T[i][0] == "PermutationPair" and T[i + 1][0] == "::" and T[i + 2][0] in ("singletons", "new"))
        # PermutationPair::singletons(lhs, rhs): the two columns, evaluated
        pairs = []
        for i in range(lo, hi):
            if T[i][0] == "PermutationPair" and T[i + 1][0] == "::" and T[i + 2][0] == "singletons" and T[i + 3][0] == "(":
                a, b = w.split(T[i + 4:src.match(i + 3)])
                pairs.append([w.expr(a, src, {}), w.expr(b, src, {})])
        if len(pairs) == facts["permutation_pairs"]:
            facts["permutation_pair_columns"] = pairs
        if any(T[i][0] in ("for", "map", "for_each", "extend") for i in range(lo, hi)):
            raise Unsupported(f"{path}: permutation_pairs() builds its list in a loop")
    return facts


def named_columns(ref, path, fn_name, argval):
    """Column indices a `ctl_data_*` / `ctl_filter_*` function of the reference names: every constant expression of its body (`COL_X`,
    `COL_R.start + i`, a range `A..B` or a bare range constant = all of its columns) evaluated with the function's usize parameter bound to `argval`.
    -> (sorted indices, exact): exact is False when the body selects among the named columns (`match`, `skip`, `take`, `filter`, `if`):
    the set is then an upper bound of what the returned columns read."""
    src = Src.get(path)
    T = src.toks
    params, body = src.functions()[fn_name]
    w = Walker(ref)
    env = {}
    ptoks = [t for t, _ in T[params[0] + 1:params[1]]]
    if "usize" in ptoks and ":" in ptoks:
        env[ptoks[ptoks.index(":") - 1]] = argval if argval is not None else 0
    cols, exact = set(), True
    i = body[0] + 1
    while i < body[1]:
        t = T[i][0]
        if t in ("match", "skip", "take", "filter", "if", "step_by"):
            exact = False
        if re.match(r"^[A-Z][A-Z0-9_]*$", t) and T[i - 1][0] != "::" and T[i + 1][0] != "::":
            # the longest run of tokens from here that still evaluates: CONST (.start | .end)? ((+|-) term)*
            j, best = i + 1, None
            while j <= body[1]:
                try:
                    v = w.expr(T[i:j], src, env)
                    best = (j, v)
                except (Unsupported, TypeError, AttributeError, KeyError):
                    pass
                if j < body[1] and (T[j][0] in (".", "start", "end", "+", "-", "..") or re.match(r"^\d", T[j][0]) or T[j][0] in env or re.match(r"^[A-Z][A-Z0-9_]*$", T[j][0])):
                    j += 1
                else:
                    break
            if best is not None:
                j, v = best
                if isinstance(v, R):
                    cols |= set(range(v.start, v.end))
                elif isinstance(v, int) and not isinstance(v, bool):
                    cols.add(v)
                i = j
                continue
        i += 1
    return sorted(cols), exact


def ctl_facts(ref):
    """stark/ola_stark.rs: per cross-table lookup, in all_cross_table_lookups() order: looked table, looking entries per table, and per
    entry the columns its data / filter functions name"""
    path = os.path.join(ref.root, "stark", "ola_stark.rs")
    src = Src.get(path)
    T = src.toks
    fns = src.functions()
    lo, hi = fns["all_cross_table_lookups"][1]
    names = [T[i][0] for i in range(lo, hi) if re.match(r"^ctl_\w+$", T[i][0]) and T[i + 1][0] == "("]

    def call_of(toks):
        """`Some(cpu_stark::ctl_filter_x(i))` / `mem_ctl_data()` / `None` -> (module or None, function, has an argument) or None"""
        names_ = [t for t, _ in toks]
        if names_ == ["None"]:
            return None
        if names_[0] == "Some":
            names_ = names_[2:-1]
        k = names_.index("(")
        path_ = [x for x in names_[:k] if x != "::"]
        if len(path_) == 1 and path_[0] in imports:
            path_ = list(imports[path_[0]])
        return (path_[-2] if len(path_) > 1 else None, path_[-1], names_[k + 1] != ")")

    # `use crate::memory::memory_stark::{self, ctl_data as mem_ctl_data, ctl_filter as mem_ctl_filter};`: local name -> (module, name there)
    imports = {}
    i = 0
    while i < len(T):
        if T[i][0] == "use":
            j = i + 1
            path_ = []
            while T[j][0] not in (";", "{"):
                if T[j][0] != "::":
                    path_.append(T[j][0])
                j += 1
            if T[j][0] == "{":
                e = src.match(j)
                for item in Walker(ref).split(T[j + 1:e]):
                    it = [t for t, _ in item]
                    if len(it) == 3 and it[1] == "as":
                        imports[it[2]] = (path_[-1], it[0])
                    elif len(it) == 1 and it[0] != "self":
                        imports[it[0]] = (path_[-1], it[0])
                j = e
            elif len(path_) >= 2:
                imports[path_[-1]] = (path_[-2], path_[-1])
            i = j
        i += 1

    def file_of(module, fn):
        cands = [f for f in ref.files if fn in Src.get(f).functions() and (module is None or os.path.basename(f) == module + ".rs")]
        cands = [f for f in cands if os.path.realpath(f).startswith(os.path.realpath(ref.root))]
        if len(cands) != 1:
            raise Unsupported(f"{path}: {module}::{fn} resolves to {cands}")
        return cands[0]

    out = []
    for name in names:
        blo, bhi = fns[name][1]
        w = Walker(ref)

        def count(a, b, argvals, looking):
            """TableWithColumns::new(Table::X ...) occurrences between a and b, with the values the enclosing `(A..B).map(|i| ..)` feeds them"""
            i = a
            while i < b:
                if T[i][0] == "(" and T[i + 1][0] not in (")",) and i + 2 < b:
                    # `(A..B).map(|i| { ... })` / `.for_each`: the closure runs once per item (the result is consumed by extend / collect)
                    e = src.match(i)
                    if e + 3 < b and T[e + 1][0] == "." and T[e + 2][0] in ("map", "for_each", "flat_map") and any(x[0] == ".." for x in T[i + 1:e]):
                        its = w.items(T[i:e + 1], src, (blo, bhi), {}, i)
                        if argvals != [None] or its.ints is None:
                            raise Unsupported(f"{path}:{T[i][1]}: nested or unevaluable map around TableWithColumns")
                        ce = src.match(e + 3)
                        count(e + 4, ce, list(its.ints), looking)
                        i = ce + 1
                        continue
                if T[i][0] == "TableWithColumns" and T[i + 1][0] == "::" and T[i + 2][0] == "new" and T[i + 3][0] == "(" and T[i + 4][0] == "Table":
                    args = w.split(T[i + 4:src.match(i + 3)])
                    looking.append({"table": T[i + 6][0], "tok": i, "args": argvals, "data": call_of(args[1]), "filter": call_of(args[2])})
                i += 1

        found = []
        count(blo, bhi, [None], found)
        # the looked table: second argument of CrossTableLookup::new
        idx = [i for i in range(blo, bhi) if T[i][0] == "CrossTableLookup" and T[i + 2][0] == "new"]
        if len(idx) != 1:
            raise Unsupported(f"{path}: {name} has {len(idx)} CrossTableLookup::new")
        a = idx[0] + 3
        args = w.split(T[a + 1:src.match(a)])
        second = args[1]
        if second[0][0] == "TableWithColumns":
            looked_tok = T.index(second[0], a)
        else:
            var = second[0][0]
            looked_tok = None
            for i in range(blo, bhi):
                if T[i][0] == "let" and T[i + 1][0] == var:
                    j = i
                    while T[j][0] != "TableWithColumns":
                        j += 1
                    looked_tok = j
            if looked_tok is None:
                raise Unsupported(f"{path}: {name}: looked table `{var}` not found")
        looked = [f for f in found if f["tok"] == looked_tok]
        if len(looked) != 1 or looked[0]["args"] != [None]:
            raise Unsupported(f"{path}: {name}: cannot identify the looked table")

        def entry(f):
            """one record per TableWithColumns the reference constructs: the columns its data and filter functions name"""
            recs = []
            for av in f["args"]:
                m, fn, has = f["data"]
                dcols, dexact = named_columns(ref, file_of(m, fn), fn, av if has else None)
                r = {"table": f["table"], "data_fn": fn + ("(%d)" % av if has and av is not None else "()"), "data_file": os.path.relpath(file_of(m, fn), ref.root),
                     "data_columns": dcols, "data_exact": dexact}
                if f["filter"] is not None:
                    m, fn, has = f["filter"]
                    fcols, fexact = named_columns(ref, file_of(m, fn), fn, av if has else None)
                    r.update({"filter_fn": fn + ("(%d)" % av if has and av is not None else "()"), "filter_file": os.path.relpath(file_of(m, fn), ref.root),
                              "filter_columns": fcols, "filter_exact": fexact})
                recs.append(r)
            return recs

        cnt = Counter()
        for f in found:
            if f["tok"] != looked_tok:
                cnt[f["table"]] += len(f["args"])
        # in the order the entries are written down in the function (every ctl_* builds its vector in that order: a `vec![..]` literal,
        # or lets followed by extend / chain in the same sequence)
        order = [r for f in found if f["tok"] != looked_tok for r in entry(f)]
        out.append({"name": name, "looked": entry(looked[0])[0], "looking": dict(sorted(cnt.items())), "looking_in_source_order": order})
    return out


def extract(reference):
    ref = Ref(reference)
    tables, problems = [], []
    for name, rel in TABLES:
        try:
            tables.append({"table": name, **table_facts(ref, name, rel)})
        except Unsupported as e:
            problems.append({"table": name, "reason": str(e)})
            tables.append({"table": name, "file": rel, "unwalked": str(e)})
    try:
        ctls = ctl_facts(ref)
    except Unsupported as e:
        problems.append({"table": "ctl", "reason": str(e)})
        ctls = []
    return {"generator": "tools/extract_air_emits.py", "source": "Sin7Y/olavm circuits/src (walked, not transcribed)", "tables": tables,
            "cross_table_lookups": ctls, "problems": problems}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=FIXTURE)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    data = extract(a.reference)
    text = json.dumps(data, indent=1) + "\n"
    for t in data["tables"]:
        if "emit_kinds" in t:
            c = Counter(t["emit_kinds"])
            named = sum(1 for x in t["emit_direct_cells"] if x)
            print("%-14s %4d emits %s  columns %s degree %s permutation pairs %s; %d emits name %d cells directly" % (
                t["table"], len(t["emit_kinds"]), dict(c), t.get("columns"), t.get("constraint_degree"), t.get("permutation_pairs"), named, sum(len(x) for x in t["emit_direct_cells"])))
        else:
            print("%-14s NOT WALKED: %s" % (t["table"], t["unwalked"]))
    print(len(data["cross_table_lookups"]), "cross-table lookups,", sum(sum(c["looking"].values()) for c in data["cross_table_lookups"]), "looking entries")
    if a.check:
        if open(a.out).read() != text:
            raise SystemExit(a.out + " is stale")
        print("fixture is up to date")
        return
    open(a.out, "w").write(text)
    print("wrote", a.out)


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
