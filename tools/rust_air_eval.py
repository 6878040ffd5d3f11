#!/usr/bin/env python3
"""The reference's twelve `eval_packed_generic` EVALUATED from their Rust source on given rows -- a small interpreter for the subset
of Rust the AIR files are written in.

tools/extract_air_emits.py pins how many constraints each table emits, in which order, of which kind, and which cells they name.
This goes the rest of the way: it RUNS the reference's constraint code.  `eval_packed_generic` of every `*_stark.rs` (and whatever it
calls: the `cpu/*.rs` opcode files, `CpuAdjacentRowWrapper::from_vars`, `lookup.rs`, `plonk_common.rs::reduce_with_powers`, the
Poseidon layers of `core/src/util/poseidon_utils.rs` with plonky2's constant tables, `OlaOpcode::binary_bit_mask`) is parsed into an
AST and interpreted with `P` = one Goldilocks element: `lv` / `nv` are the rows handed in, `yield_constr.constraint*(x)` records
(kind, value of x).  The values are then compared, constraint by constraint, with what olavm_amd/air/ola_tables.py -- the
hand transcription that the oracle, the verifier restatement and the GPU all consume -- evaluates to on the same rows
(tests/test_air_eval.py), and committed as vectors (tests/golden/air_eval_vectors.json) so that the comparison also runs where the
reference tree is absent.

The interpreter knows Rust's expression grammar (precedence, closures, blocks, if / match / for, patterns, struct literals, macros
`vec!` / `izip!`), the iterator adaptors the files use, arrays / slices / ranges / tuples, mutable locals and element assignment,
associated constants and functions, and enum values with `match`.  Types, generics, lifetimes and trait bounds are skipped.  It
does not know anything about the AIRs: every constant, column index, loop bound and formula comes out of the reference's files.
Anything it cannot parse or evaluate raises with file:line.

    python tools/rust_air_eval.py [--reference /root/reference] [--points 3] [--out tests/golden/air_eval_vectors.json] [--check]
"""
import argparse
import functools
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import extract_air_emits as X          # noqa: E402  (tokeniser, file index, table list)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "air_eval_vectors.json")
P = 0xFFFFFFFF00000001


class RustError(Exception):
    pass


@functools.lru_cache(maxsize=None)
def _rp(path):
    return os.path.realpath(path)


class Fe:
    """one Goldilocks element (what `P: PackedField`, `P::Scalar`, `F` and `FE` all are here)"""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = int(v) % P

    def __repr__(self):
        return "Fe(%d)" % self.v

    def __eq__(self, o):                    # arrays and structs of elements compare by value (`digest == cap.0[i]`)
        return isinstance(o, Fe) and o.v == self.v

    def __hash__(self):
        return hash(self.v)


class Fe2:
    """one element of the quadratic extension F[X]/(X^2 - 7) (plonky2_field extension/quadratic.rs with GoldilocksField's W = 7,
    goldilocks_extensions.rs:12): what `F::Extension` / `FE` are when the verifier runs"""
    __slots__ = ("a", "b")
    W = 7

    def __init__(self, a, b=0):
        self.a, self.b = int(a) % P, int(b) % P

    @staticmethod
    def of(x):
        return x if isinstance(x, Fe2) else Fe2(x.v, 0)

    def __repr__(self):
        return "Fe2(%d,%d)" % (self.a, self.b)

    def __eq__(self, o):
        if isinstance(o, Fe):
            o = Fe2(o.v, 0)
        return isinstance(o, Fe2) and (self.a, self.b) == (o.a, o.b)

    def __hash__(self):
        return hash((self.a, self.b))

    def add(self, o):
        return Fe2(self.a + o.a, self.b + o.b)

    def sub(self, o):
        return Fe2(self.a - o.a, self.b - o.b)

    def mul(self, o):
        return Fe2(self.a * o.a + Fe2.W * self.b * o.b, self.a * o.b + self.b * o.a)

    def inverse(self):
        # 1 / (a + bX) = (a - bX) / (a^2 - 7 b^2)
        n = (self.a * self.a - Fe2.W * self.b * self.b) % P
        if n == 0:
            raise ZeroDivisionError("inverse of zero in the extension field")
        i = pow(n, P - 2, P)
        return Fe2(self.a * i, -self.b * i)

    def pow(self, e):
        r, b = Fe2(1, 0), self
        while e:
            if e & 1:
                r = r.mul(b)
            b = b.mul(b)
            e >>= 1
        return r


class Rng:
    def __init__(self, a, b):
        self.start, self.end = a, b

    def items(self):
        return list(range(self.start, self.end))


class Enum:
    def __init__(self, ty, variant, payload=None):
        self.ty, self.variant, self.payload = ty, variant, payload

    def __eq__(self, o):
        return isinstance(o, Enum) and (self.ty, self.variant) == (o.ty, o.variant)

    def __hash__(self):
        return hash((self.ty, self.variant))


class Struct(dict):
    pass


class Closure:
    def __init__(self, params, body, env, src):
        self.params, self.body, self.env, self.src = params, body, env, src


class FnRef:
    """a path that names a function, used as a value"""

    def __init__(self, segs, env, src, line):
        self.segs, self.env, self.src, self.line = segs, env, src, line


class TInt(int):
    """an integer whose Rust type is known (a cast, a typed parameter, to_canonical_u64): only `to_le_bytes` cares"""
    bits = 64

    def __new__(cls, v, bits):
        o = int.__new__(cls, v)
        o.bits = bits
        return o


INT_BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64, "u128": 128, "usize": 64}


class Cursor:
    """std::io::Cursor<Vec<u8>> as far as serialization.rs uses it for writing"""

    def __init__(self, data=()):
        self.data = list(data)

    def write_all(self, b):
        for x in b:
            if not (isinstance(x, int) and 0 <= x < 256):
                raise RustError("write_all: not a byte")
            self.data.append(int(x))
        return None                             # Ok(())


class Opt:
    """`Some(v)` where it must be told from v itself: `cond.then(|| vec)` followed by `.map(|v| ..)` would otherwise map the vector's items.
    (A literal `Some(x)` stays x; `None` is None.)"""

    def __init__(self, v):
        self.v = v


class PtrCast:
    def __init__(self, base, group):
        self.base, self.group = base, group


class GroupView:
    """`slice::from_raw_parts(xs.as_ptr() as *const [T; 2], n)`: xs seen as n arrays of 2 -- the same memory, so later writes to xs show"""

    def __init__(self, base, group, n):
        self.base, self.group, self.n = base, group, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise IndexError(i)
        return self.base[self.group * i:self.group * (i + 1)]


class MapEntry:
    def __init__(self, d, k):
        self.d, self.k = d, k


class RayonScope:
    """the `s` of `rayon::scope(|s| ..)`: `s.spawn(f)` runs f now"""


class ChunksMut:
    """`xs.chunks_mut(n)` (also chunks_exact_mut / par_chunks_mut): the chunks are handed out as copies and written back after each use --
    a chunk is only ever changed through the closure or loop body it was handed to"""

    def __init__(self, base, n, enumerate_=False, left=None):
        self.base, self.n, self.enumerate_, self.left = base, n, enumerate_, left        # left: what `xs.zip(ys.chunks_mut(n))` pairs the chunks with

    def run(self, body):
        for k in range(0, len(self.base), self.n):
            c = self.base[k:k + self.n]
            item = c
            if self.left is not None:
                if k // self.n >= len(self.left):
                    break
                item = (self.left[k // self.n], c)
            body((k // self.n, item) if self.enumerate_ else item)
            if len(c) != min(self.n, len(self.base) - k):
                raise RustError("a chunk of chunks_mut changed its length")
            self.base[k:k + self.n] = c


class ElemRef:
    """`&mut xs[i]` for a scalar element, made where the source says so (`&mut xs` / `xs.iter_mut()` iterated or zipped): `*r = v` writes the
    array; a plain read of the variable gives the current value"""
    __slots__ = ("c", "k")

    def __init__(self, c, k):
        self.c, self.k = c, k

    def get(self):
        return self.c[self.k]

    def set(self, v):
        self.c[self.k] = v


def refs_of(lst):
    return [ElemRef(lst, i) if (x is None or isinstance(x, (Fe, Fe2, int, bool))) else x for i, x in enumerate(lst)]


def is_mut_iter(node):
    """`&mut xs`, `xs.iter_mut()` (possibly followed by adaptors that keep the items: enumerate is handled by the caller)"""
    return node[0] == "mutref" or (node[0] == "mcall" and node[2] in ("iter_mut", "par_iter_mut"))


class Params(list):
    """a function's parameter patterns; `.types[i]` = the tokens of parameter i's type (None for self); `.ret` = head of the return type"""
    types = ()
    ret = None


class Return(Exception):
    def __init__(self, v):
        self.v = v


NOT_FOUND = object()


class Consumer:
    """ConstraintConsumer<P> (constraint_consumer.rs:34-78), recording instead of accumulating"""

    def __init__(self):
        self.emits = []


# ------------------------------------------------------------------------------------------------ parser
BINOPS = [["||"], ["&&"], ["==", "!=", "<", ">", "<=", ">="], ["|"], ["^"], ["&"], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]
ASSIGN = ("=", "+=", "-=", "*=", "/=")


class Parser:
    def __init__(self, src, lo, hi):
        self.src, self.T, self.i, self.hi = src, src.toks, lo, hi

    def err(self, what):
        line = self.T[min(self.i, len(self.T) - 1)][1]
        ctx = " ".join(t for t, _ in self.T[max(0, self.i - 6):self.i + 6])
        return RustError(f"{self.src.path}:{line}: {what} (near `{ctx}`)")

    def peek(self, k=0):
        j = self.i + k
        return self.T[j][0] if j < self.hi else None

    def next(self):
        t = self.peek()
        if t is None:
            raise self.err("unexpected end")
        self.i += 1
        return t

    def eat(self, t):
        if self.peek() == t:
            self.i += 1
            return True
        return False

    def expect(self, t):
        if self.peek() != t:
            raise self.err(f"expected `{t}`, found `{self.peek()}`")
        self.i += 1

    def line(self):
        return self.T[min(self.i, len(self.T) - 1)][1]

    # ---- types are skipped
    def skip_generics(self):
        """at `<`: skip to the matching `>` (`>>` closes two)"""
        depth = 0
        while True:
            t = self.next()
            if t == "<":
                depth += 1
            elif t == ">":
                depth -= 1
            elif t == ">>":
                depth -= 2
            elif t == "<<":
                depth += 2
            elif t in ("(", "[", "{"):
                self.i = self.src.match(self.i - 1) + 1
            if depth <= 0:
                return

    def skip_type(self, stops):
        depth = 0
        while True:
            t = self.peek()
            if t is None:
                return
            if depth == 0 and t in stops:
                return
            if t in ("(", "["):
                self.i = self.src.match(self.i) + 1
                continue
            if t == "{":          # const generic `{ N }`
                self.i = self.src.match(self.i) + 1
                continue
            if t == "<":
                depth += 1
            elif t == "<<":                     # `Vec<<F as Extendable<D>>::Extension>`
                depth += 2
            elif t == ">":
                depth -= 1
            elif t == ">>":
                depth -= 2
            self.i += 1

    # ---- patterns
    def pattern(self):
        alts = [self.pattern1()]
        while self.peek() == "|" and self.peek(1) not in ("|",):
            # only inside match arms (closure parameter lists never reach here with a bare `|`)
            self.next()
            alts.append(self.pattern1())
        return alts[0] if len(alts) == 1 else ("por", alts)

    def pattern1(self):
        t = self.peek()
        if t in ("&", "&&"):
            self.next()
            self.eat("mut")
            return self.pattern1()
        if t in ("mut", "ref"):
            self.next()
            return self.pattern1()
        if t == "_":
            self.next()
            return ("pwild",)
        if t == "(":
            self.next()
            items = []
            while not self.eat(")"):
                items.append(self.pattern())
                self.eat(",")
            return ("ptuple", items)
        if t == "[":
            self.next()
            items = []
            while not self.eat("]"):
                items.append(self.pattern())
                self.eat(",")
            return ("ptuple", items)
        if t == "-" or re.match(r"^\d", t):
            neg = self.eat("-")
            v = parse_int(self.next())
            return ("plit", -v if neg else v)
        if re.match(r"^[A-Za-z_]", t):
            segs = [self.next()]
            while self.peek() == "::":
                self.next()
                segs.append(self.next())
            if self.peek() == "(":           # Some(x)
                self.next()
                inner = []
                while not self.eat(")"):
                    inner.append(self.pattern())
                    self.eat(",")
                return ("pcall", segs, inner)
            if self.peek() == "{" and re.match(r"^[A-Z]", segs[-1]):        # PermutationInstance { pair, challenge: GrandProductChallenge { beta, gamma }, .. }
                self.next()
                fields = []
                while not self.eat("}"):
                    if self.eat(".."):
                        continue
                    self.eat("ref")
                    self.eat("mut")
                    name = self.next()
                    sub = self.pattern() if self.eat(":") else ("pid", name)
                    fields.append((name, sub))
                    self.eat(",")
                return ("pstruct", segs, fields)
            if len(segs) == 1 and re.match(r"^[a-z_]", segs[0]):
                return ("pid", segs[0])
            return ("ppath", segs)
        raise self.err("pattern")

    # ---- statements
    def block(self):
        self.expect("{")
        stmts, tail = [], None
        pending = None
        while self.peek() != "}":
            if self.peek() == "#":           # attribute
                self.next()
                close_ = self.src.match(self.i)
                toks = [t for t, _ in self.src.toks[self.i + 1:close_]]
                self.i = close_ + 1
                # #[cfg(feature = "x")] / #[cfg(not(feature = "x"))] / #[cfg(test)]: the statement exists only in such a build
                if toks[:1] == ["cfg"]:
                    inner = toks[2:-1]
                    neg = inner[:1] == ["not"]
                    if neg:
                        inner = inner[2:-1]
                    if len(inner) == 3 and inner[0] == "feature" and inner[1] == "=":
                        pending = (inner[2].strip('"'), neg)
                    elif inner == ["test"]:
                        pending = ("test", neg)
                continue
            if self.eat(";"):
                continue
            s, is_expr, needs_semi = self.statement()
            if pending is not None:
                s = ("cfg", pending[0], pending[1], s)
                pending = None
            if is_expr and self.peek() == "}":
                tail = s
                break
            if needs_semi and not self.eat(";") and self.peek() != "}":
                raise self.err("expected `;`")
            stmts.append(s)
        self.expect("}")
        return ("block", stmts, tail)

    def statement(self):
        t = self.peek()
        ln = self.line()
        if t == "let":
            self.next()
            pat = self.pattern()
            if self.eat(":"):
                self.skip_type(("=", ";"))
            init = None
            if self.eat("="):
                init = self.expr()
            return ("let", pat, init, ln), False, True
        if t == "for" and self.peek(1) != "<":
            self.next()
            pat = self.pattern()
            self.expect("in")
            it = self.expr(no_struct=True)
            body = self.block()
            return ("for", pat, it, body, ln), False, False
        if t == "loop":
            ln = self.line()
            self.next()
            body = self.block()
            return ("while", ("path", ["true"], ln), body, ln), False, False
        if t == "while":
            self.next()
            cond = self.expr(no_struct=True)
            body = self.block()
            return ("while", cond, body, ln), False, False
        if t == "return":
            self.next()
            e = None if self.peek() in (";", "}") else self.expr()
            return ("return", e, ln), False, True
        if t in ("if", "match", "{", "unsafe"):
            e = self.expr()
            # block-like expressions need no semicolon; they may still be the tail
            return e, True, False
        e = self.expr()
        return e, True, True

    # ---- expressions
    def expr(self, no_struct=False):
        return self.assign(no_struct)

    def assign(self, ns):
        lhs = self.range_(ns)
        if self.peek() in (">>", "<<") and self.peek(1) == "=":
            op = self.next() + self.next()
            rhs = self.assign(ns)
            return ("assign", op, lhs, rhs, self.line())
        if self.peek() in ASSIGN:
            op = self.next()
            rhs = self.assign(ns)
            return ("assign", op, lhs, rhs, self.line())
        return lhs

    def range_(self, ns):
        if self.peek() in ("..", "..="):
            incl = self.next() == "..="
            hi = None if self.peek() in (None, ")", "]", ",", ";", "}") else self.binary(0, ns)
            return ("range", None, hi, incl)
        lo = self.binary(0, ns)
        if self.peek() in ("..", "..="):
            incl = self.next() == "..="
            stop = self.peek() in (None, ")", "]", ",", ";", "}") or (ns and self.peek() == "{")
            hi = None if stop else self.binary(0, ns)
            return ("range", lo, hi, incl)
        return lo

    def binary(self, level, ns):
        if level == len(BINOPS):
            return self.cast(ns)
        a = self.binary(level + 1, ns)
        while self.peek() in BINOPS[level] and not (self.peek() in (">>", "<<") and self.peek(1) == "="):      # `x >>= 1` is tokenised `>>` `=`
            # `|` as a binary operator never follows an expression in these files except as bit-or on integers; closures start a primary
            op = self.next()
            b = self.binary(level + 1, ns)
            a = ("bin", op, a, b, self.line())
        return a

    def cast(self, ns):
        e = self.unary(ns)
        while self.peek() == "as":
            self.next()
            if self.peek() == "*" and self.peek(1) in ("mut", "const"):      # `as *mut [F]`: a pointer to the same memory
                self.next()
                self.next()
                group = 1
                if self.peek() == "[":              # `*const [H::Hash; 2]`: the memory seen as arrays of 2
                    close_ = self.src.match(self.i)
                    inner = [t for t, _ in self.src.toks[self.i + 1:close_]]
                    if ";" in inner and len(inner) - inner.index(";") == 2 and inner[-1].isdigit():
                        group = int(inner[-1])
                self.skip_type((")", ",", ";", "}", "]"))
                if group != 1:
                    e = ("ptrcast", e, group)
                continue
            ty = self.next()                  # the type's head
            while self.peek() == "::":
                self.next()
                ty = self.next()
            if self.peek() == "<":
                self.skip_generics()
            e = ("cast", e, ty)
        return e

    def unary(self, ns):
        t = self.peek()
        if t == "-":
            self.next()
            return ("neg", self.unary(ns), self.line())
        if t == "!":
            self.next()
            return ("not", self.unary(ns))
        if t == "*":
            self.next()
            return ("deref", self.unary(ns))   # references are transparent, except as the target of `*x = ...`
        if t in ("&", "&&"):
            self.next()
            if self.eat("mut"):
                return ("mutref", self.unary(ns))      # `&mut x`: a callee may assign through it
            return self.unary(ns)
        return self.postfix(ns)

    def args(self):
        self.expect("(")
        out = []
        while not self.eat(")"):
            out.append(self.expr())
            self.eat(",")
        return out

    def postfix(self, ns):
        e = self.primary(ns)
        while True:
            t = self.peek()
            if t == "(":
                e = ("call", e, self.args(), self.line())
            elif t == "[":
                self.next()
                idx = self.expr()
                self.expect("]")
                e = ("index", e, idx, self.line())
            elif t == "?":
                self.next()
                e = ("try", e)
            elif t == ".":
                self.next()
                name = self.next()
                if re.match(r"^\d", name):
                    e = ("tfield", e, int(name))
                    continue
                if self.peek() == "::":
                    self.next()
                    self.skip_generics()
                if self.peek() == "(":
                    e = ("mcall", e, name, self.args(), self.line())
                else:
                    e = ("field", e, name, self.line())
            else:
                return e

    def closure(self):
        params = []
        if self.eat("||"):
            pass
        else:
            self.expect("|")
            while not self.eat("|"):
                params.append(self.pattern1())
                if self.eat(":"):
                    self.skip_type((",", "|"))
                self.eat(",")
        if self.peek() == "->":
            self.next()
            self.skip_type(("{",))
        body = self.expr()
        return ("closure", params, body)

    def primary(self, ns):
        t = self.peek()
        ln = self.line()
        if t is None:
            raise self.err("expression expected")
        if t == "(":
            self.next()
            items, trailing = [], False
            while not self.eat(")"):
                items.append(self.expr())
                trailing = self.eat(",")
            if len(items) == 1 and not trailing:
                return items[0]
            return ("tuple", items)
        if t == "[":
            self.next()
            if self.eat("]"):
                return ("array", [])
            first = self.expr()
            if self.eat(";"):
                n = self.expr()
                self.expect("]")
                return ("repeat", first, n)
            items = [first]
            while not self.eat("]"):
                self.expect(",")
                if self.eat("]"):
                    break
                items.append(self.expr())
            return ("array", items)
        if t == "{":
            return self.block()
        if t == "unsafe":
            self.next()
            return self.block()
        if t in ("|", "||", "move"):
            self.eat("move")
            return self.closure()
        if t == "if":
            self.next()
            if self.peek() == "let":
                self.next()
                pat = self.pattern()
                self.expect("=")
                cond = ("iflet", pat, self.expr(no_struct=True))
            else:
                cond = self.expr(no_struct=True)
            then = self.block()
            other = None
            if self.eat("else"):
                other = self.primary(ns) if self.peek() == "if" else self.block()
            return ("if", cond, then, other)
        if t == "match":
            self.next()
            subj = self.expr(no_struct=True)
            self.expect("{")
            arms = []
            while not self.eat("}"):
                pat = self.pattern()
                guard = None
                if self.eat("if"):
                    guard = self.expr()
                self.expect("=>")
                body = self.expr()
                self.eat(",")
                arms.append((pat, guard, body))
            return ("match", subj, arms, ln)
        if re.match(r"^\d", t):
            self.next()
            return ("num", parse_int(t))
        if t.startswith('"') or t.startswith('b"'):
            self.next()
            return ("str", t)
        if t.startswith("'"):
            self.next()
            return ("str", t)
        if t.endswith("!"):               # macro
            name = self.next()
            open_ = self.peek()
            close = {"(": ")", "[": "]", "{": "}"}[open_]
            end = self.src.match(self.i)
            if name in ("vec!", "izip!", "array!"):
                self.next()
                if name == "vec!":
                    first = None if self.peek() == close else self.expr()
                    if first is not None and self.eat(";"):
                        n = self.expr()
                        self.expect(close)
                        return ("repeat", first, n)
                    items = [] if first is None else [first]
                    while not self.eat(close):
                        self.expect(",")
                        if self.eat(close):
                            break
                        items.append(self.expr())
                    return ("array", items)
                items = []
                while not self.eat(close):
                    items.append(self.expr())
                    self.eat(",")
                return ("macro", name, items)
            if name in ("timed!", "batch_iter_mut!"):      # macro_rules of the reference (util/timing.rs:180, plonky2_util lib.rs:284): arguments are expressions
                ln = self.line()
                self.next()
                items = []
                while not self.eat(close):
                    items.append(self.expr())
                    self.eat(",")
                return ("macro", name, items, ln)
            if name == "cfg!":                # cfg!(feature = "parallel"): asked of the driver's feature set
                toks = [t for t, _ in self.src.toks[self.i + 1:end]]
                self.i = end + 1
                if len(toks) == 3 and toks[0] == "feature" and toks[1] == "=":
                    return ("macro", name, [], toks[2].strip('"'))
                return ("unit",)
            if name == "ensure!":             # anyhow's ensure!(cond, "..."): the condition matters, the message does not
                ln = self.line()
                self.next()
                cond = self.expr()
                self.i = end + 1
                return ("macro", name, [cond], ln)
            self.i = end + 1                  # assert! / debug_assert! / println! ...: no value
            return ("unit",)
        if re.match(r"^[A-Za-z_]", t):
            segs = [self.next()]
            while self.peek() == "::":
                self.next()
                if self.peek() == "<":
                    self.skip_generics()
                    continue
                segs.append(self.next())
            if segs[0] == "<":
                raise self.err("qualified path")
            if self.peek() == "{" and not ns and re.match(r"^[A-Z]", segs[-1]) and self.looks_like_struct_literal():
                self.next()
                fields = []
                while not self.eat("}"):
                    if self.eat(".."):
                        fields.append(("..", self.expr()))
                        continue
                    name = self.next()
                    if self.eat(":"):
                        fields.append((name, self.expr()))
                    else:
                        fields.append((name, ("path", [name], ln)))
                    self.eat(",")
                return ("struct", segs, fields)
            return ("path", segs, ln)
        if t == "<":                        # <T as Trait>::f
            self.skip_generics()
            segs = []
            while self.peek() == "::":
                self.next()
                segs.append(self.next())
            return ("path", segs, ln)
        raise self.err(f"unexpected `{t}`")

    def looks_like_struct_literal(self):
        # `Name { ident: ...` or `Name { ident, ...` or `Name { }`
        a, b = self.peek(1), self.peek(2)
        return a == "}" or (a is not None and re.match(r"^[a-z_]", a) and b in (":", ",", "}")) or a == ".."


def parse_int(t):
    t = re.sub(r"_?(?:[ui](?:8|16|32|64|128|size))$", "", t).replace("_", "")
    if t.startswith("0x"):
        return int(t, 16)
    if t.startswith("0b"):
        return int(t, 2)
    return int(t)


# ------------------------------------------------------------------------------------------------ interpreter
FIELD_CTORS = {"from_canonical_u64", "from_canonical_usize", "from_canonical_u32", "from_canonical_u16", "from_canonical_u8",
               "from_noncanonical_u64", "from_canonical_i64", "from_noncanonical_u128", "from_basefield", "from_canonical_u128"}
FIELD_CONSTS = {"ONES": 1, "ZEROS": 0, "ONE": 1, "ZERO": 0, "TWO": 2, "NEG_ONE": P - 1}
INT_TYPES = {"u8", "u16", "u32", "u64", "u128", "usize", "i32", "i64", "isize"}


class Interp:
    def __init__(self, ref):
        self.ref = ref
        self.fn_cache = {}
        self.const_cache = {}
        self.find_cache = {}
        self.impl_cache = {}
        self.enum_cache = {}
        self.import_cache = {}
        self.assoc_cache = {}
        self.pending_writeback = []
        self.near_cache = {}
        self.field_consts = {}              # TWO_ADICITY, POWER_OF_TWO_GENERATOR, MULTIPLICATIVE_GROUP_GENERATOR of goldilocks_field.rs (set by a driver)
        self.extension = False              # True while the verifier runs: FE = F::Extension is the quadratic extension (D = 2), not F itself
        self.find_any_hint = None           # see the Rng `find_any` method
        self.packing_width = 0              # P::WIDTH where the reference's code is generic over a packing (a driver sets 1)
        self.features = set()               # cargo features `cfg!(feature = "..")` sees: the reference builds with "parallel" (a driver sets it)
        self.num_threads = 8                # what maybe_rayon::current_num_threads() answers (results must not depend on it)
        self.fn_hooks = {}                  # free function name -> python function of the argument list
        self.assoc_hooks = {}               # (type, fn) -> python function of the argument list: the reference's calls into crates outside its tree
        self.extra_files = []               # files outside the AIR tree whose impl blocks a driver needs (plonky2's fri/, iop/challenger.rs)
        self.generics = {}                  # generic parameter -> the types tried for `H::f(..)`: {"H": ["PoseidonHash", "Hasher"]} (set by a driver)
        self.permutation_hook = None        # what `H::Permutation::permute` runs (the driver installs the interpreted poseidon_naive)
        self.depth = 0

    # ---- lookup of functions and constants in the reference tree
    def fn_ast(self, path, name):
        key = (_rp(path), name)
        if key not in self.fn_cache:
            src = X.Src.get(path)
            fns = src.functions()
            if name not in fns:
                raise RustError(f"{path}: fn {name} not found")
            (plo, phi), (blo, bhi) = self.impl_index(path).get(("", name), fns[name])       # a free function before a method of the same name
            params = self.parse_params(src, plo, phi)
            params.ret = self.ret_head(src, phi, blo)
            body = Parser(src, blo, bhi + 1).block()
            self.fn_cache[key] = (params, body, src)
        return self.fn_cache[key]

    def impl_index(self, path):
        """(type, fn) -> (params range, body range) for the functions inside `impl .. Type { }` and `trait Name { }` blocks of a file
        (trait blocks are indexed under the trait's name: default methods)"""
        key = _rp(path)
        if key in self.impl_cache:
            return self.impl_cache[key]
        src = X.Src.get(path)
        T = src.toks
        out = {}
        i = 0
        while i < len(T):
            t = T[i][0]
            if t in ("impl", "trait") and (i == 0 or T[i - 1][0] not in ("::", ".")):
                j = i + 1
                if T[j][0] == "<":
                    depth = 0
                    while True:
                        depth += {"<": 1, ">": -1, "<<": 2, ">>": -2}.get(T[j][0], 0)
                        j += 1
                        if depth <= 0:
                            break
                names, depth = [], 0
                while T[j][0] not in ("{", ";") or depth > 0:
                    x = T[j][0]
                    if x == "where" and depth == 0:
                        break
                    depth += {"<": 1, ">": -1, "<<": 2, ">>": -2}.get(x, 0)
                    if depth == 0 and re.match(r"^[A-Za-z_]\w*$", x) and x not in ("for", "dyn", "const", "mut") and T[j - 1][0] not in ("'",):
                        names.append(x)
                    if x == "for" and depth == 0:
                        names = []
                    j += 1
                while T[j][0] not in ("{", ";"):
                    j = src.match(j) + 1 if T[j][0] in ("(", "[") else j + 1
                if T[j][0] == "{" and names:
                    ty = names[-1] if t == "impl" else names[0]
                    end = src.match(j)
                    k = j + 1
                    while k < end:
                        if T[k][0] == "fn":
                            name = T[k + 1][0]
                            q = k + 2
                            if T[q][0] == "<":
                                depth = 0
                                while True:
                                    depth += {"<": 1, ">": -1, "<<": 2, ">>": -2}.get(T[q][0], 0)
                                    q += 1
                                    if depth <= 0:
                                        break
                            while T[q][0] != "(":
                                q += 1
                            pe = src.match(q)
                            b = pe + 1
                            while T[b][0] not in ("{", ";"):
                                b = src.match(b) + 1 if T[b][0] in ("[", "(") else b + 1
                            if T[b][0] == "{":
                                out.setdefault((ty, name), ((q, pe), (b, src.match(b))))
                                k = src.match(b)
                        elif T[k][0] == "{":
                            k = src.match(k)
                        k += 1
                    i = end
            elif t == "fn":                     # a free function: indexed under the empty type name
                name = T[i + 1][0]
                q = i + 2
                if T[q][0] == "<":
                    depth = 0
                    while True:
                        depth += {"<": 1, ">": -1, "<<": 2, ">>": -2}.get(T[q][0], 0)
                        q += 1
                        if depth <= 0:
                            break
                while T[q][0] != "(":
                    q += 1
                pe = src.match(q)
                b = pe + 1
                while T[b][0] not in ("{", ";"):
                    b = src.match(b) + 1 if T[b][0] in ("[", "(") else b + 1
                if T[b][0] == "{":
                    out.setdefault(("", name), ((q, pe), (b, src.match(b))))
                    i = src.match(b)
            i += 1
        self.impl_cache[key] = out
        return out

    def default_struct(self, name, src):
        """`#[derive(Default)]` of a struct with named fields: Vec -> [], Option -> None, integers -> 0, bool -> false, F -> 0 (found in the
        file at hand, then anywhere in the tree); a struct that is not found stays an empty record"""
        st = Struct({"__name__": name})
        for f in [src.path] + self.ref.files + self.extra_files:
            T = X.Src.get(f).toks
            for i in range(len(T) - 2):
                if T[i][0] == "struct" and T[i + 1][0] == name:
                    j = i + 2
                    while T[j][0] not in ("{", ";", "("):
                        j += 1
                    if T[j][0] != "{":
                        return st
                    end = X.Src.get(f).match(j)
                    k = j + 1
                    while k < end:
                        while T[k][0] == "#":
                            k = X.Src.get(f).match(k + 1) + 1
                        if T[k][0] == "pub":
                            k += 1
                            if T[k][0] == "(":
                                k = X.Src.get(f).match(k) + 1
                        field = T[k][0]
                        ty0 = T[k + 2][0]
                        st[field] = [] if ty0 == "Vec" else 0 if ty0 in INT_TYPES else False if ty0 == "bool" else Fe(0) if ty0 in ("F", "P") else None
                        depth = 0
                        k += 2
                        while k < end and not (T[k][0] == "," and depth == 0):
                            depth += {"<": 1, ">": -1, ">>": -2, "(": 1, ")": -1, "[": 1, "]": -1}.get(T[k][0], 0)
                            k += 1
                        k += 1
                    return st
        return st

    def call_free(self, path, name, args):
        """a free function of one file (`call_fn` takes the first `fn` of that name, which may be a method)"""
        v = self.call_assoc("", name, args, path)
        if v is NOT_FOUND:
            raise RustError(f"{path}: no free fn {name}")
        return v

    def find_assoc(self, ty, name, here):
        """the file and ranges of `ty::name`, nearest file first"""
        key = (ty, name, here)
        if key not in self.assoc_cache:
            found = None
            for f in [here] + self.ref.files + self.extra_files:
                r = self.impl_index(f).get((ty, name))
                if r is not None:
                    found = (f, r)
                    break
            self.assoc_cache[key] = found
        return self.assoc_cache[key]

    def fn_ast_at(self, path, ranges):
        key = (_rp(path), ranges[1][0])
        if key not in self.fn_cache:
            src = X.Src.get(path)
            (plo, phi), (blo, bhi) = ranges
            params = self.parse_params(src, plo, phi)
            params.ret = self.ret_head(src, phi, blo)
            self.fn_cache[key] = (params, Parser(src, blo, bhi + 1).block(), src)
        return self.fn_cache[key]

    @staticmethod
    def ret_head(src, phi, blo):
        """the head of a function's declared return type (`-> PolynomialValues<F>` gives "PolynomialValues"), or None"""
        toks = [t for t, _ in src.toks[phi + 1:blo]]
        if toks[:1] != ["->"]:
            return None
        toks = [t for t in toks[1:] if t not in ("&", "mut") and not t.startswith("'")]
        return toks[0] if toks else None

    @staticmethod
    def coerce_return(v, params):
        """`res.into()` as the tail of a function declared to return PolynomialValues<F> / PolynomialCoeffs<F>: the From<Vec<F>> impl"""
        ret = getattr(params, "ret", None)
        if isinstance(v, list) and ret in ("PolynomialValues", "PolynomialCoeffs"):
            return Struct({"__name__": ret, ("values" if ret == "PolynomialValues" else "coeffs"): v})
        return v

    @staticmethod
    def parse_params(src, plo, phi):
        pp = Parser(src, plo + 1, phi)
        params, types = Params(), []
        while pp.peek() is not None:
            if pp.peek() in ("&", "mut") or pp.peek().startswith("'"):
                pp.next()
                continue
            if pp.peek() == "self":
                pp.next()
                params.append(("pid", "self"))
                types.append(None)
                pp.eat(",")
                continue
            pat = pp.pattern1()
            pp.expect(":")
            t0 = pp.i
            pp.skip_type((",",))
            types.append([t for t, _ in src.toks[t0:pp.i] if t not in ("&", "mut") and not t.startswith("'")])
            pp.eat(",")
            params.append(pat)
        params.types = types
        return params

    @staticmethod
    def bind_generics(params, shift, args, env):
        """`stark: S` / `stark: &S`: inside the function `S::COLUMNS` is the associated constant of that argument's type"""
        for i, a in enumerate(args):                        # `x: u32`: the width travels with the value
            j = i + shift
            if j < len(params.types) and params.types[j] and len(params.types[j]) == 1 and params.types[j][0] in INT_BITS \
                    and isinstance(a, int) and not isinstance(a, bool) and params[j][0] == "pid":
                bits = INT_BITS[params.types[j][0]]
                if not 0 <= a < 1 << bits:
                    raise RustError(f"argument {a} does not fit {params.types[j][0]}")
                env[params[j][1]] = TInt(a, bits)
        g = None
        for i, a in enumerate(args):
            j = i + shift
            if j < len(params.types) and params.types[j] and len(params.types[j]) == 1 and re.match(r"^[A-Z]\w?$", params.types[j][0]) and isinstance(a, Struct):
                g = g if g is not None else dict(env.get("__generic__", {}))
                g[params.types[j][0]] = a
        if g is not None:
            env["__generic__"] = g

    def call_assoc(self, ty, name, args, here, self_val=None, has_self=False):
        if (ty, name) in self.assoc_hooks:
            return self.assoc_hooks[(ty, name)](([self_val] + list(args)) if has_self else args)
        found = self.find_assoc(ty, name, here)
        if found is None:
            return NOT_FOUND
        path, ranges = found
        params, body, src = self.fn_ast_at(path, ranges)
        env = {"__src__": src, "__impl__": ty}
        ps = list(params)
        if ps and ps[0] == ("pid", "self"):
            if not has_self:
                self_val, args = args[0], args[1:]
            env["self"] = self_val
            ps = ps[1:]
        elif has_self:
            return NOT_FOUND
        if len(ps) != len(args):
            raise RustError(f"{path}: {ty}::{name} takes {len(ps)} arguments, {len(args)} given")
        for p_, a in zip(ps, args):
            self.bind(p_, a, env, src)
        if isinstance(params, Params):
            self.bind_generics(params, len(params) - len(ps), args, env)
        wb, self.pending_writeback = self.pending_writeback, []
        shift = 1 if (params and params[0] == ("pid", "self") and not has_self) else 0
        self.depth += 1
        if self.depth > 200:
            raise RustError(f"{path}: call depth")
        try:
            v = self.ev(body, env, src)
            if isinstance(v, list) and body[0] == "block" and body[2] is not None and body[2][0] == "field" and body[2][1] == ("path", ["self"], body[2][1][2]):
                v = list(v)          # `fn compact(&mut self) -> [F; W] { ..; self.sponge_state }`: an array leaves a borrowed struct by copy
            return self.coerce_return(v, params)
        except Return as r:
            return self.coerce_return(r.v, params)
        finally:
            self.depth -= 1
            for i, cenv, cname in wb:
                j = i - shift
                if 0 <= j < len(ps) and ps[j][0] == "pid" and ps[j][1] in env:
                    cenv[cname] = env[ps[j][1]]
                    cenv.setdefault("__assigned__", set()).add(cname)

    def enum_value(self, e, src):
        """discriminant of an enum value (`Table::Memory as usize`)"""
        key = e.ty
        if key not in self.enum_cache:
            table = None
            for f in self.ref.files:
                T = X.Src.get(f).toks
                for i in range(len(T) - 2):
                    if T[i][0] == "enum" and T[i + 1][0] == e.ty and T[i + 2][0] == "{":
                        end = X.Src.get(f).match(i + 2)
                        table, nxt, k = {}, 0, i + 3
                        while k < end:
                            if T[k][0] == "#":
                                k = X.Src.get(f).match(k + 1) + 1
                                continue
                            name = T[k][0]
                            k += 1
                            if T[k][0] == "=":
                                nxt = parse_int(T[k + 1][0])
                                k += 2
                            table[name] = nxt
                            nxt += 1
                            if T[k][0] in ("(", "{"):
                                k = X.Src.get(f).match(k) + 1
                            if T[k][0] == ",":
                                k += 1
                        break
                if table is not None:
                    break
            if table is None:
                raise RustError(f"{src.path}: enum {e.ty} not found")
            self.enum_cache[key] = table
        return self.enum_cache[key][e.variant]

    def imports(self, src):
        """`use a::b::{self, x as y, z};` of a file: local name -> (module, name there)"""
        key = _rp(src.path)
        if key in self.import_cache:
            return self.import_cache[key]
        T, out, i = src.toks, {}, 0
        while i < len(T):
            if T[i][0] == "use":
                j, path_ = i + 1, []
                while T[j][0] not in (";", "{"):
                    if T[j][0] != "::":
                        path_.append(T[j][0])
                    j += 1
                if T[j][0] == "{" and path_:
                    e = src.match(j)
                    k, item = j + 1, []
                    depth = 0
                    while k <= e:
                        t = T[k][0]
                        if (t == "," and depth == 0) or k == e:
                            if len(item) == 3 and item[1] == "as":
                                out[item[2]] = (path_[-1], item[0])
                            elif len(item) == 1 and item[0] != "self":
                                out[item[0]] = (path_[-1], item[0])
                            item = []
                        else:
                            depth += t == "{"
                            depth -= t == "}"
                            if depth == 0 and t not in ("{", "}"):
                                item.append(t)
                        k += 1
                    j = e
                elif len(path_) >= 2:
                    if len(path_) >= 4 and path_[-2] == "as":
                        out[path_[-1]] = (path_[-4], path_[-3])
                    else:
                        out[path_[-1]] = (path_[-2], path_[-1])
                i = j
            i += 1
        self.import_cache[key] = out
        return out

    def find_fn_file(self, name, here, module=None):
        key = (name, _rp(here), module)
        if key not in self.find_cache:
            f = self.ref.find_fn(name, here, module)
            if f is None and module is not None:
                for x in self.extra_files:          # `serial::permute` from cfft/mod.rs: the sibling file of that name
                    if os.path.basename(x) == module + ".rs" and os.path.dirname(x) == os.path.dirname(here) and ("", name) in self.impl_index(x):
                        f = x
                        break
            if f is None and module in ("super", "self", "crate") and ("", name) in self.impl_index(here):
                f = here
            if f is None and module == "super":
                for x in self.extra_files:
                    if os.path.basename(x) == "mod.rs" and os.path.dirname(x) == os.path.dirname(here) and ("", name) in self.impl_index(x):
                        f = x
                        break
            if f is None and module is not None:
                f = self.ref.find_fn(name, here, None)
            if f is None:
                for x in self.extra_files:          # plonky2's fri/, util/: free functions only
                    if ("", name) in self.impl_index(x):
                        f = x
                        break
            self.find_cache[key] = f
        return self.find_cache[key]

    def const_value(self, name, src):
        nk = (name, src.path)
        if nk not in self.near_cache:
            for x in self.extra_files:
                self.ref.index_consts(x)
            cands = self.ref.consts.get(name)
            c = None
            if cands:
                c = self.ref.near(cands, src.path)
                if c is None:
                    c = cands[0]
            self.near_cache[nk] = c
        c = self.near_cache[nk]
        if c is None:
            return None
        key = (name, c[0])
        if key not in self.const_cache:
            csrc = X.Src.get(c[0])
            toks = c[1]
            # the expression's token span inside csrc
            lo = csrc.toks.index(toks[0]) if toks else 0
            # (token tuples are unique objects per position only by identity: find by identity)
            for k in range(len(csrc.toks)):
                if csrc.toks[k] is toks[0]:
                    lo = k
                    break
            p = Parser(csrc, lo, lo + len(toks))
            self.const_cache[key] = self.ev(p.expr(), {"__src__": csrc}, csrc)
        return self.const_cache[key]

    # ---- helpers
    def err(self, src, line, what):
        return RustError(f"{src.path}:{line}: {what}")

    @staticmethod
    def truthy(v):
        return bool(v)

    def bind(self, pat, v, env, src):
        k = pat[0]
        if k == "pid":
            env[pat[1]] = v
        elif k == "pwild":
            pass
        elif k == "ptuple":
            v = list(v)
            if len(v) != len(pat[1]):
                raise RustError(f"{src.path}: tuple pattern of {len(pat[1])} against {len(v)} values")
            for p, x in zip(pat[1], v):
                self.bind(p, x, env, src)
        elif k == "pcall":
            self.bind(pat[2][0], v.v if isinstance(v, Opt) else v, env, src)            # Some(x)
        elif k == "pstruct":
            for name, sub in pat[2]:
                if name not in v:
                    raise RustError(f"{src.path}: no field `{name}` in {v.get('__name__')}")
                self.bind(sub, v[name], env, src)
        else:
            raise RustError(f"{src.path}: pattern {k} cannot bind")

    def matches(self, pat, v, env, src):
        k = pat[0]
        if k == "pwild":
            return True
        if k == "pid":
            env[pat[1]] = v
            return True
        if k == "plit":
            return (v.v if isinstance(v, Fe) else v) == pat[1]
        if k == "por":
            return any(self.matches(p, v, env, src) for p in pat[1])
        if k == "ppath":
            segs = pat[1]
            if segs[-1] == "None":
                return v is None
            if isinstance(v, Enum):
                return v.variant == segs[-1]
            c = self.const_value(segs[-1], src)
            return c is not None and c == v
        if k == "pcall":
            if pat[1][-1] == "Some":
                if v is None:
                    return False
                return self.matches(pat[2][0], v.v if isinstance(v, Opt) else v, env, src)
            if isinstance(v, Enum) and v.variant == pat[1][-1] and v.payload is not None and len(v.payload) == len(pat[2]):
                return all(self.matches(p_, x, env, src) for p_, x in zip(pat[2], v.payload))
            return False
        if k == "ptuple":
            return len(v) == len(pat[1]) and all(self.matches(p, x, env, src) for p, x in zip(pat[1], v))
        if k == "pstruct":
            return isinstance(v, dict) and all(name in v and self.matches(sub, v[name], env, src) for name, sub in pat[2])
        raise RustError(f"{src.path}: pattern {k}")

    def powers(self, pw, n, src, line):
        out, w = [], Fe(1)
        for _ in range(n):
            v = w
            for f in pw.maps:
                v = self.call_closure(f, [v])
            out.append(v)
            w = self.binop("*", w, pw.base, src, line)
        return out

    def call_closure(self, c, args):
        if isinstance(c, FnRef):
            return self.call(("call", ("path", c.segs, c.line), [("value", a) for a in args], c.line), c.env, c.src)
        env = dict(c.env)
        if len(c.params) == 1 and len(args) != 1:
            args = [tuple(args)]
        if len(c.params) != len(args):
            if len(args) == 1 and isinstance(args[0], (tuple, list)) and len(args[0]) == len(c.params):
                args = list(args[0])
            else:
                raise RustError(f"{c.src.path}: closure of {len(c.params)} parameters called with {len(args)}")
        env["__declared__"] = set(n_ for p in c.params for n_ in pattern_names(p))
        env["__assigned__"] = set()
        for p, a in zip(c.params, args):
            self.bind(p, a, env, c.src)
        try:
            return self.ev(c.body, env, c.src)
        finally:
            self.write_back(c.env, env)          # a closure that assigns a captured variable (`pair_index >>= 1` inside `.map(|i| ..)`)

    def call_fn(self, path, name, args, self_val=None):
        params, body, src = self.fn_ast(path, name)
        env = {"__src__": src}
        ps = list(params)
        if ps and ps[0] == ("pid", "self"):
            env["self"] = self_val
            ps = ps[1:]
        if len(ps) != len(args):
            raise RustError(f"{path}: {name} takes {len(ps)} arguments, {len(args)} given")
        for p, a in zip(ps, args):
            self.bind(p, a, env, src)
        if isinstance(params, Params):
            self.bind_generics(params, len(params) - len(ps), args, env)
        wb, self.pending_writeback = self.pending_writeback, []
        self.depth += 1
        if self.depth > 200:
            raise RustError(f"{path}: call depth")
        try:
            return self.coerce_return(self.ev(body, env, src), params)
        except Return as r:
            return self.coerce_return(r.v, params)
        finally:
            self.depth -= 1
            for i, cenv, cname in wb:
                if i < len(ps) and ps[i][0] == "pid" and ps[i][1] in env:
                    cenv[cname] = env[ps[i][1]]
                    cenv.setdefault("__assigned__", set()).add(cname)

    def iterate(self, v, src, line=0):
        if isinstance(v, Rng):
            return v.items()
        if isinstance(v, (list, tuple)):
            return list(v)
        raise self.err(src, line, f"cannot iterate {type(v).__name__}")

    # ---- arithmetic
    OP_TRAIT = {"+": "add", "-": "sub", "*": "mul", "/": "div"}

    def binop(self, op, a, b, src, line):
        if isinstance(a, Struct) and "__name__" in a and op in self.OP_TRAIT:        # `&acc + &p`: impl Add for &PolynomialCoeffs<F>
            r = self.call_assoc(a["__name__"], self.OP_TRAIT[op], [a, b], src.path)
            if r is not NOT_FOUND:
                return r
        if isinstance(a, (ZeroSum, OneProduct)) and isinstance(b, (Fe, Fe2)):
            a = Fe(int(a))
        if isinstance(b, (ZeroSum, OneProduct)) and isinstance(a, (Fe, Fe2)):
            b = Fe(int(b))
        if isinstance(a, Fe2) or isinstance(b, Fe2):
            if not (isinstance(a, (Fe, Fe2)) and isinstance(b, (Fe, Fe2))):
                raise self.err(src, line, f"extension element {op} {type(b).__name__ if isinstance(a, Fe2) else type(a).__name__}")
            a, b = Fe2.of(a), Fe2.of(b)
            if op == "+":
                return a.add(b)
            if op == "-":
                return a.sub(b)
            if op == "*":
                return a.mul(b)
            if op == "/":
                return a.mul(b.inverse())
            if op == "==":
                return a == b
            if op == "!=":
                return not (a == b)
            raise self.err(src, line, f"extension element operator {op}")
        if isinstance(a, Fe) or isinstance(b, Fe):
            if not (isinstance(a, Fe) and isinstance(b, Fe)):
                # `P * u64` does not type-check in Rust: a sign that the interpreter mis-typed a value
                raise self.err(src, line, f"field element {op} {type(b).__name__ if isinstance(a, Fe) else type(a).__name__}")
            if op == "+":
                return Fe(a.v + b.v)
            if op == "-":
                return Fe(a.v - b.v)
            if op == "*":
                return Fe(a.v * b.v)
            if op == "==":
                return a.v == b.v
            if op == "!=":
                return a.v != b.v
            if op == "/":
                return Fe(a.v * pow(b.v, P - 2, P))
            raise self.err(src, line, f"field element operator {op}")
        if isinstance(a, Enum) or isinstance(b, Enum):
            if op == "==":
                return a == b
            if op == "!=":
                return a != b
        f = {"+": lambda: a + b, "-": lambda: a - b, "*": lambda: a * b, "/": lambda: a // b, "%": lambda: a % b, "==": lambda: a == b,
             "!=": lambda: a != b, "<": lambda: a < b, ">": lambda: a > b, "<=": lambda: a <= b, ">=": lambda: a >= b, "&&": lambda: a and b,
             "||": lambda: a or b, "<<": lambda: a << b, ">>": lambda: a >> b, "&": lambda: a & b, "|": lambda: a | b, "^": lambda: a ^ b}.get(op)
        if f is None:
            raise self.err(src, line, f"operator {op}")
        return f()

    # ---- evaluation
    def ev(self, n, env, src):
        k = n[0]
        if k == "num":
            return n[1]
        if k == "path":
            return self.path_value(n[1], env, src, n[2])
        if k == "bin":
            op = n[1]
            if op == "&&":
                return self.truthy(self.ev(n[2], env, src)) and self.truthy(self.ev(n[3], env, src))
            if op == "||":
                return self.truthy(self.ev(n[2], env, src)) or self.truthy(self.ev(n[3], env, src))
            return self.binop(op, self.ev(n[2], env, src), self.ev(n[3], env, src), src, n[4])
        if k == "neg":
            v = self.ev(n[1], env, src)
            return Fe(-v.v) if isinstance(v, Fe) else -v
        if k == "not":
            return not self.ev(n[1], env, src)
        if k == "cast":
            v = self.ev(n[1], env, src)
            if isinstance(v, Enum):
                return self.enum_value(v, src)
            v = v.v if isinstance(v, Fe) else (int(v) if isinstance(v, bool) else v)
            bits = INT_BITS.get(n[2] if len(n) > 2 else None)
            if bits is not None and isinstance(v, int):
                v = TInt(v & ((1 << bits) - 1), bits)          # `sum as u64`: truncation
            return v
        if k == "index":
            base = self.ev(n[1], env, src)
            idx = self.ev(n[2], env, src)
            return self.index(base, idx, src, n[3])
        if k == "field":
            base = self.ev(n[1], env, src)
            return self.field(base, n[2], src, n[3])
        if k == "tfield":
            return self.ev(n[1], env, src)[n[2]]
        if k == "tuple":
            return tuple(self.ev(x, env, src) for x in n[1])
        if k == "array":
            return [self.ev(x, env, src) for x in n[1]]
        if k == "repeat":
            v, cnt = self.ev(n[1], env, src), self.ev(n[2], env, src)
            return [clone(v) for _ in range(cnt)]
        if k == "range":
            a = None if n[1] is None else self.ev(n[1], env, src)
            b = None if n[2] is None else self.ev(n[2], env, src)
            if n[3] and b is not None:
                b += 1
            return Rng(a, b)
        if k in ("mutref", "deref"):
            return self.ev(n[1], env, src)
        if k == "ptrcast":
            v = self.ev(n[1], env, src)
            return PtrCast(v, n[2]) if isinstance(v, list) else v
        if k == "try":                           # `expr?`: an Err leaves the function (Ok(x) and Some(x) are x here)
            v = self.ev(n[1], env, src)
            if isinstance(v, Enum) and v.variant == "Err":
                raise Return(v)
            return v
        if k == "closure":
            return Closure(n[1], n[2], env, src)
        if k == "block":
            return self.block(n, env, src)
        if k == "if":
            cond = n[1]
            inner = dict(env)
            if cond[0] == "iflet":
                ok = self.matches(cond[1], self.ev(cond[2], env, src), inner, src)
            else:
                ok = self.truthy(self.ev(cond, env, src))
            if ok:
                return self.scoped(n[2], env, src, inner if cond[0] == "iflet" else None)
            return self.scoped(n[3], env, src) if n[3] is not None else None
        if k == "match":
            v = self.ev(n[1], env, src)
            for pat, guard, body in n[2]:
                inner = dict(env)
                if self.matches(pat, v, inner, src) and (guard is None or self.truthy(self.ev(guard, inner, src))):
                    r = self.ev(body, inner, src)
                    self.write_back(env, inner)
                    return r
            raise self.err(src, n[3], f"no match arm for {v!r}")
        if k == "struct":
            s = Struct()
            s["__name__"] = env.get("__impl__", "Self") if n[1][-1] == "Self" else n[1][-1]
            for name, e in n[2]:
                if name == "..":
                    s.update(self.ev(e, env, src))
                else:
                    s[name] = self.ev(e, env, src)
            return s
        if k == "macro":
            if n[1] == "izip!":
                cols = [self.iterate(self.ev(x, env, src), src) for x in n[2]]
                return [tuple(t) for t in zip(*cols)]
            if n[1] == "cfg!":
                return n[3] in self.features
            if n[1] == "timed!":                     # push / pop on the timing tree around the last argument: its value
                return self.ev(n[2][-1], env, src)
            if n[1] == "batch_iter_mut!":
                # plonky2_util lib.rs:284-319 with feature "parallel": batches of len / threads.next_power_of_two() items, or one call c(e, 0) when
                # that is below the minimum
                vals = [self.ev(x, env, src) for x in n[2]]
                e, c = vals[0], vals[-1]
                min_batch = vals[1] if len(vals) == 3 else 1
                threads = self.num_threads
                batch = len(e) // (1 if threads <= 1 else 1 << (threads - 1).bit_length()) if "parallel" in self.features else 0
                if batch < min_batch or batch < 1:
                    self.call_closure(c, [e, 0])
                else:
                    ChunksMut(e, batch, True).run(lambda item: self.call_closure(c, [item[1], item[0] * batch]))
                return None
            if n[1] == "ensure!":
                if not self.truthy(self.ev(n[2][0], env, src)):
                    raise Return(Enum("Result", "Err", [f"{os.path.basename(src.path)}:{n[3]}"]))
                return None
            raise RustError(f"{src.path}: macro {n[1]}")
        if k == "call":
            return self.call(n, env, src)
        if k == "mcall":
            return self.mcall(n, env, src)
        if k == "assign":
            return self.assign(n, env, src)
        if k == "value":
            return n[1]
        if k == "cfg":
            if (n[1] in self.features) != n[2]:
                return self.stmt(n[3], env, src) if n[3][0] in ("let", "for", "while", "return") else self.ev(n[3], env, src)
            return None
        if k == "unit":
            return None
        if k == "str":
            return n[1]
        if k in ("let", "for", "while", "return"):
            return self.stmt(n, env, src)
        raise RustError(f"{src.path}: cannot evaluate node {k}")

    def scoped(self, node, env, src, inner=None):
        if inner is None:
            inner = dict(env)
            inner["__declared__"] = set()       # a fresh scope: what the enclosing block declared is OUTER here
            inner["__assigned__"] = set()
        r = self.ev(node, inner, src)
        self.write_back(env, inner)
        return r

    @staticmethod
    def write_back(env, inner):
        """assignments to OUTER variables made inside a nested scope (a fresh dict per scope keeps shadowing `let`s local)"""
        for name in inner.get("__assigned__", ()):
            if name in env and name not in inner.get("__declared__", ()):
                env[name] = inner[name]
                env.setdefault("__assigned__", set()).add(name)

    def block(self, n, env, src):
        inner = dict(env)
        inner["__declared__"] = set()
        inner["__assigned__"] = set()
        try:
            for s in n[1]:
                self.stmt(s, inner, src)
            return self.ev(n[2], inner, src) if n[2] is not None else None
        finally:
            self.write_back(env, inner)

    def stmt(self, s, env, src):
        k = s[0]
        if k == "let":
            v = self.ev(s[2], env, src) if s[2] is not None else None
            if s[2] is not None and s[2][0] in ("path", "field") and s[1][0] == "pid" and isinstance(v, list) and len(v) <= 4096 \
                    and all(x is None or isinstance(x, (Fe, Fe2, int, bool)) for x in v):
                v = list(v)          # `let mut state = input;`: an array is copied, a Vec is moved (the source is dead afterwards): a copy is right for both
            names = pattern_names(s[1])
            env.setdefault("__declared__", set()).update(names)
            self.bind(s[1], v, env, src)
            return None
        if k == "for":
            seq = self.ev(s[2], env, src)
            if isinstance(seq, ChunksMut):
                def body(item):
                    inner = dict(env)
                    inner["__declared__"] = set(pattern_names(s[1]))
                    inner["__assigned__"] = set()
                    self.bind(s[1], item, inner, src)
                    self.ev(s[3], inner, src)
                    self.write_back(env, inner)
                seq.run(body)
                return None
            if is_mut_iter(s[2]) and isinstance(seq, list):
                seq = refs_of(seq)
            for item in self.iterate(seq, src, s[4]):
                inner = dict(env)
                inner["__declared__"] = set(pattern_names(s[1]))
                inner["__assigned__"] = set()
                self.bind(s[1], item, inner, src)
                self.ev(s[3], inner, src)
                self.write_back(env, inner)
            return None
        if k == "while":
            guard = 0
            while self.truthy(self.ev(s[1], env, src)):
                self.scoped(s[2], env, src)
                guard += 1
                if guard > 1 << 20:
                    raise self.err(src, s[3], "while loop does not end")
            return None
        if k == "return":
            raise Return(self.ev(s[1], env, src) if s[1] is not None else None)
        return self.ev(s, env, src)

    def assign(self, n, env, src):
        op, lhs, rhs, line = n[1], n[2], self.ev(n[3], env, src), n[4]
        if op != "=" and op[:-1] in self.OP_TRAIT:
            target = lhs[1] if lhs[0] in ("deref", "paren") else lhs
            cur = self.ev(target, env, src) if target[0] in ("path", "field", "mcall", "index") else None
            if isinstance(cur, Struct) and "__name__" in cur:              # `final_poly += quotient`: impl AddAssign for PolynomialCoeffs<F>
                r = self.call_assoc(cur["__name__"], self.OP_TRAIT[op[:-1]] + "_assign", [cur, rhs], src.path)
                if r is not NOT_FOUND:
                    return None
        if lhs[0] == "deref" and lhs[1][0] == "path" and len(lhs[1][1]) == 1 and isinstance(env.get(lhs[1][1][0]), ElemRef):
            r = env[lhs[1][1][0]]
            r.set(rhs if op == "=" else self.binop(op[:-1], r.get(), rhs, src, line))
            return None
        if lhs[0] == "deref":
            lhs = lhs[1]
            if op == "=" and isinstance(rhs, list):
                cur = self.ev(lhs, env, src)
                if isinstance(cur, list):              # `*state = new_array`: the caller's array changes
                    cur[:] = rhs
                    return None
        if lhs[0] == "path" and len(lhs[1]) == 1:
            name = lhs[1][0]
            if name not in env:
                raise self.err(src, line, f"assignment to unknown `{name}`")
            env[name] = rhs if op == "=" else self.binop(op[:-1], env[name], rhs, src, line)
            env.setdefault("__assigned__", set()).add(name)
            return None
        if lhs[0] == "index":
            base = self.ev(lhs[1], env, src)
            idx = self.ev(lhs[2], env, src)
            base[idx] = rhs if op == "=" else self.binop(op[:-1], base[idx], rhs, src, line)
            return None
        if lhs[0] == "field":
            base = self.ev(lhs[1], env, src)
            base[lhs[2]] = rhs if op == "=" else self.binop(op[:-1], base[lhs[2]], rhs, src, line)
            return None
        raise self.err(src, line, "assignment target")

    def index(self, base, idx, src, line):
        if isinstance(base, GroupView):
            return base[idx]
        if isinstance(idx, Rng):
            a = idx.start or 0
            b = len(base) if idx.end is None else idx.end
            if not (0 <= a <= b <= len(base)):
                raise self.err(src, line, f"slice {a}..{b} of {len(base)}")
            if idx.start is None and idx.end is None and isinstance(base, list):
                return base                        # `v[..]` is the whole slice itself (concurrent.rs:82 takes a second &mut to it)
            return list(base[a:b])
        if isinstance(idx, Fe):
            raise self.err(src, line, "field element as index")
        if not (0 <= idx < len(base)):
            raise self.err(src, line, f"index {idx} of {len(base)}")
        return base[idx]

    def field(self, base, name, src, line):
        if isinstance(base, Rng) and name in ("start", "end"):
            return getattr(base, name)
        if isinstance(base, (Struct, dict)):
            if name in base:
                return base[name]
            raise self.err(src, line, f"no field `{name}` in {base.get('__name__')}")
        raise self.err(src, line, f"field `{name}` of {type(base).__name__}")

    def path_value(self, segs, env, src, line):
        name = segs[-1]
        if len(segs) == 1:
            if name in env:
                v = env[name]
                return v.get() if isinstance(v, ElemRef) else v
            if name == "None":
                return None
            if name in ("true", "false"):
                return name == "true"
        if len(segs) >= 2 and name in FIELD_CONSTS and segs[-2] in ("P", "F", "FE", "Scalar", "GoldilocksField", "Self", "Extension"):
            return Fe2(FIELD_CONSTS[name], 0) if segs[-2] == "Extension" else Fe(FIELD_CONSTS[name])
        if len(segs) >= 2 and name in self.field_consts and segs[-2] in ("F", "Self", "GoldilocksField"):
            return Fe(self.field_consts[name]) if name != "TWO_ADICITY" else self.field_consts[name]
        if len(segs) == 2 and segs[0] == "P" and name == "WIDTH" and self.packing_width:
            return self.packing_width
        if len(segs) == 2 and segs[0] in INT_TYPES and name == "BITS":
            return {"u8": 8, "u16": 16, "u32": 32, "u64": 64, "u128": 128, "usize": 64, "i32": 32, "i64": 64, "isize": 64}[segs[0]]
        if len(segs) == 2 and segs[0] in INT_TYPES and name in ("MAX", "MIN"):
            bits = {"u8": 8, "u16": 16, "u32": 32, "u64": 64, "u128": 128, "usize": 64}.get(segs[0])
            return (1 << bits) - 1 if name == "MAX" else 0
        if re.match(r"^[A-Z][A-Z0-9_]*$", name):
            if len(segs) == 2:
                owner = env.get("self") if segs[0] == "Self" else env.get("__generic__", {}).get(segs[0])
                if isinstance(owner, Struct) and owner.get("__file__"):
                    osrc = X.Src.get(owner["__file__"])
                    if any(c[0] == owner["__file__"] for c in self.ref.consts.get(name, ())):
                        return self.const_value(name, osrc)           # `const COLUMNS` of that type's impl block (its own file)
            v = self.const_value(name, src)
            if v is not None:
                return v
        if len(segs) >= 2 and re.match(r"^[A-Z]", segs[-2]) and re.match(r"^[A-Z]", name):
            return Enum(segs[-2], name)                   # OlaOpcode::ADD
        if re.match(r"^[a-z_]", name) and len(segs) >= 2:
            return FnRef(segs, env, src, line)            # `.map(F::Extension::from_basefield)`: a function named, not called
        raise self.err(src, line, f"unknown name `{'::'.join(segs)}`")

    def call(self, n, env, src):
        try:
            return self._call(n, env, src)
        finally:
            self.pending_writeback = []         # a built-in took the call: nothing to copy back

    def _call(self, n, env, src):
        callee, line = n[1], n[3]
        args = [self.ev(a, env, src) for a in n[2]]
        # `f(&mut counter)`: scalars are values here, so what the callee left in that parameter is copied back afterwards
        self.pending_writeback = [(i, env, a[1][1][0]) for i, a in enumerate(n[2])
                                  if a[0] == "mutref" and a[1][0] == "path" and len(a[1][1]) == 1 and not isinstance(env.get(a[1][1][0]), (list, dict))]
        if callee[0] != "path":
            f = self.ev(callee, env, src)
            if isinstance(f, Closure):
                return self.call_closure(f, args)
            raise self.err(src, line, "call of a non-function")
        segs = callee[1]
        name = segs[-1]
        if name in self.fn_hooks and (len(segs) == 1 or not re.match(r"^[A-Z]", segs[-2])):
            return self.fn_hooks[name](args)
        if len(segs) == 1 and name in env and isinstance(env[name], Closure):
            return self.call_closure(env[name], args)
        if name == "from_basefield" and isinstance(args[0], (Fe, Fe2)) and (self.extension or (len(segs) >= 2 and segs[-2] == "Extension")):
            return Fe2.of(args[0])          # `F::Extension::from_basefield` always; `FE::from_basefield` where FE is the extension (the verifier)
        if name in FIELD_CTORS:
            v = args[0]
            return v if isinstance(v, Fe) else Fe(v)
        if name in ("Some", "Ok", "Box", "Reverse"):
            return args[0]
        if name in ("new", "with_capacity") and len(segs) >= 2 and segs[-2] in ("Vec", "String"):
            return []
        if name == "from_noncanonical_u96":
            return Fe(args[0][0] + (args[0][1] << 64))
        if name in ("from_basefield_array",):
            return Fe2(args[0][0].v, args[0][1].v) if len(args[0]) == 2 else tuple(args[0])
        if name == "default" and len(segs) >= 2 and segs[-2] == "Default":
            return None
        if name == "permute" and len(segs) >= 2 and segs[-2] in ("Permutation", "P", "PoseidonPermutation") and self.permutation_hook is not None:
            return self.permutation_hook(args[0], segs)
        # ---- the field's own functions (plonky2_field types.rs), with the constants a driver read from goldilocks_field.rs
        if name == "primitive_root_of_unity" and self.field_consts:
            if not 0 <= args[0] <= self.field_consts["TWO_ADICITY"]:
                raise self.err(src, line, "primitive_root_of_unity: n_log out of range")
            return Fe(pow(self.field_consts["POWER_OF_TWO_GENERATOR"], 1 << (self.field_consts["TWO_ADICITY"] - args[0]), P))
        if name == "order" and self.field_consts and len(args) == 0:
            return P
        if name == "coset_shift" and self.field_consts:
            return Fe(self.field_consts["MULTIPLICATIVE_GROUP_GENERATOR"])
        if name == "batch_multiplicative_inverse" and len(args) == 1:
            return [self.method(x, "inverse", [], src, line) for x in args[0]]
        if name in ("reverse_index_bits_in_place", "reverse_index_bits") and len(args) == 1 and isinstance(args[0], list):
            # plonky2_util's cache-blocked, `unsafe` permutation (util/src/lib.rs:80-215); what it computes is its doc line: "Bit-reverse the order
            # of elements in arr"
            a = args[0]
            n_ = len(a)
            if n_ & (n_ - 1) or n_ == 0:
                raise self.err(src, line, f"{name}: length {n_} is not a power of two")
            bits = n_.bit_length() - 1
            out = [a[int(format(i, "0%db" % bits)[::-1], 2) if bits else 0] for i in range(n_)]
            if name == "reverse_index_bits":
                return out
            a[:] = out
            return None
        if name == "from_le_bytes" and len(segs) == 2 and segs[0] in INT_BITS and isinstance(args[0], list):
            if len(args[0]) * 8 != INT_BITS[segs[0]] or not all(isinstance(x, int) and 0 <= x < 256 for x in args[0]):
                raise self.err(src, line, f"{segs[0]}::from_le_bytes of {len(args[0])} items")
            return TInt(sum(int(x) << (8 * i) for i, x in enumerate(args[0])), INT_BITS[segs[0]])
        if name == "from_raw_parts" and len(args) == 2 and isinstance(args[0], PtrCast):
            if args[0].group * args[1] > len(args[0].base):
                raise self.err(src, line, "from_raw_parts beyond the allocation")
            return GroupView(args[0].base, args[0].group, args[1])
        if name == "new" and len(segs) >= 2 and segs[-2] in ("BTreeMap", "HashMap"):
            return {}
        if name == "from_slice" and len(segs) == 2 and segs[0] == "P" and self.packing_width == 1 and isinstance(args[0], list) and len(args[0]) == 1:
            return args[0][0]                       # P::from_slice(&xs[i..i + 1]) for a packing of width 1
        if name in ("max", "min") and len(args) == 2 and all(isinstance(x, int) for x in args) and (len(segs) == 1 or segs[-2] == "cmp"):
            return max(args) if name == "max" else min(args)
        if name == "current_num_threads" and not args:
            return self.num_threads
        if name == "scope" and len(segs) >= 2 and segs[-2] in ("maybe_rayon", "rayon") and len(args) == 1:
            return self.call_closure(args[0], [RayonScope()])          # the spawned closures run at once, in order
        if name == "from_fn" and "array" in segs:
            # std::array::from_fn: the length is the array type's, which only inference knows; every use in the reference is [_; NUM_TABLES]
            n_ = self.const_value("NUM_TABLES", src)
            return [self.call_closure(args[0], [i]) for i in range(n_)]
        if name == "init_gpu":
            return None
        if name == "once" and "iter" in segs:
            return [args[0]]
        if name == "repeat" and ("iter" in segs or (len(segs) == 1 and self.find_fn_file(name, src.path) is None)):
            return RepeatForever(args[0])
        if name == "from" and len(segs) >= 2 and segs[-2] in INT_TYPES | {"F", "P", "FE"}:
            return args[0]
        if name == "default" and len(segs) >= 2:
            return self.default_struct(segs[-2], src)
        if len(segs) >= 2 and segs[-2] in self.generics:
            for ty in self.generics[segs[-2]]:
                r = self.call_assoc(ty, name, args, src.path)
                if r is not NOT_FOUND:
                    return r
        if len(segs) >= 2 and (segs[-2] == "Self" or (re.match(r"^[A-Z]", segs[-2]) and segs[-2] not in ("P", "F", "FE", "C", "S", "D", "T", "H"))):
            ty = env.get("__impl__") if segs[-2] == "Self" else segs[-2]
            if ty is not None:
                r = self.call_assoc(ty, name, args, src.path)
                if r is not NOT_FOUND:
                    return r
        module = None
        if len(segs) >= 2 and segs[-2] not in ("Self",) and re.match(r"^[a-z_]", segs[-2]):
            module = segs[-2]
        if len(segs) == 1 and name not in X.Src.get(src.path).functions() and name in self.imports(src):
            module, name = self.imports(src)[name]          # `use ..::{ctl_data as mem_ctl_data}`
        target = None
        cur = src.path
        if (module is None) and name in X.Src.get(cur).functions():
            target = cur
        if target is None:
            target = self.find_fn_file(name, cur, module)
            if target is not None and len(segs) == 1 and ("", name) not in self.impl_index(target):
                # `flatten(evals)` names a free function; the first `fn flatten` found was a method (MerkleCap::flatten)
                for x in self.ref.files + self.extra_files:
                    if ("", name) in self.impl_index(x):
                        target = x
                        break
        if target is None and len(segs) >= 2 and re.match(r"^[A-Z]", segs[-2]) and re.match(r"^[A-Z]", name):
            return Enum(segs[-2], name, list(args))         # FriReductionStrategy::ConstantArityBits(4, 5)
        if target is None and len(segs) == 1 and re.match(r"^[A-Z][a-z]", name):
            st = Struct({"__name__": env.get("__impl__", name) if name == "Self" else name})       # a tuple struct: MerkleCap(cap)
            for i_, a in enumerate(args):
                st[i_] = a
            return st
        if target is None:
            raise self.err(src, line, f"function `{'::'.join(segs)}` not found")
        params, _, _ = self.fn_ast(target, name)
        if params and params[0] == ("pid", "self"):
            return self.call_fn(target, name, args[1:], args[0] if args else None)
        return self.call_fn(target, name, args)

    def mcall(self, n, env, src):
        recv_node, name, line = n[1], n[2], n[4]
        # the consumer
        recv = self.ev(recv_node, env, src)
        if isinstance(recv, Consumer):
            if name not in X.KINDS:
                raise self.err(src, line, f"consumer method {name}")
            v = self.ev(n[3][0], env, src)
            if not isinstance(v, (Fe, Fe2)):
                raise self.err(src, line, "constraint argument is not a field element")
            recv.emits.append((X.KINDS[name], v.v if isinstance(v, Fe) else (v.a, v.b), os.path.relpath(src.path, self.ref.root), line))
            return None
        if name in ("iter_mut", "par_iter_mut") and not n[3] and recv_node[0] == "index":
            base, idx = self.ev(recv_node[1], env, src), self.ev(recv_node[2], env, src)
            if isinstance(idx, Rng) and isinstance(base, list):       # `nodes[n..].par_iter_mut()`: references into nodes itself
                a = idx.start or 0
                b = len(base) if idx.end is None else idx.end
                return [ElemRef(base, i) for i in range(a, b)]
        if name in ("iter_mut", "par_iter_mut") and isinstance(recv, list) and not n[3]:
            return refs_of(recv)                    # an iterator of `&mut` items: adaptors (skip, zip, enumerate ..) pass the references on
        args = [self.ev(a, env, src) for a in n[3]]
        if name == "zip" and n[3] and is_mut_iter(n[3][0]) and isinstance(args[0], list):
            args[0] = refs_of(args[0])
        if name in ("zip", "enumerate", "for_each") and is_mut_iter(recv_node) and isinstance(recv, list):
            recv = refs_of(recv)
        if name in ("copy_from_slice", "clone_from_slice", "fill") and recv_node[0] in ("index", "mutref", "deref"):
            # `state[..k].copy_from_slice(chunk)`: a slice of a list is a copy here, so the write goes to the indexed array itself
            node = recv_node
            while node[0] in ("mutref", "deref"):
                node = node[1]
            if node[0] == "index":
                base, idx = self.ev(node[1], env, src), self.ev(node[2], env, src)
                if isinstance(idx, Rng) and isinstance(base, list):
                    a = idx.start or 0
                    b = len(base) if idx.end is None else idx.end
                    new = list(args[0]) if name != "fill" else [args[0]] * (b - a)
                    if len(new) != b - a or not (0 <= a <= b <= len(base)):
                        raise self.err(src, line, f"{name}: {len(new)} elements into {a}..{b} of {len(base)}")
                    base[a:b] = new
                    return None
        return self.method(recv, name, args, src, line)

    def method(self, r, name, args, src, line):
        if isinstance(r, Opt) or (r is None and name in ("map", "as_ref", "as_mut", "unwrap_or", "and_then", "cloned", "copied")):
            some = isinstance(r, Opt)
            if name in ("as_ref", "as_mut", "cloned", "copied"):
                return r
            if name == "clone":
                return Opt(clone(r.v))
            if name == "map":
                return Opt(self.call_closure(args[0], [r.v])) if some else None
            if name == "and_then":
                return self.call_closure(args[0], [r.v]) if some else None
            if name in ("unwrap", "expect"):
                return r.v
            if name == "unwrap_or":
                return r.v if some else args[0]
            if name == "unwrap_or_default":
                return r.v
            if name in ("is_some", "is_none"):
                return some == (name == "is_some")
            if name in ("iter", "into_iter"):
                return [r.v]
            raise self.err(src, line, f"Option::{name}")
        if isinstance(r, GroupView):
            if name in ("par_iter", "iter"):
                return [r[i] for i in range(len(r))]
            if name == "len":
                return len(r)
            raise self.err(src, line, f"raw-parts view .{name}")
        if isinstance(r, MapEntry):
            if name == "or_insert_with":
                if r.k not in r.d:
                    r.d[r.k] = self.call_closure(args[0], [])
                return r.d[r.k]
            raise self.err(src, line, f"map entry .{name}")
        if isinstance(r, dict) and not isinstance(r, Struct) and name == "entry":
            return MapEntry(r, args[0])
        if isinstance(r, dict) and not isinstance(r, Struct) and name in ("insert", "get", "contains_key", "len", "is_empty", "remove"):
            key_ = (args[0].v if isinstance(args[0], Fe) else args[0]) if args else None
            if name == "insert":
                old_ = r.get(key_)
                r[key_] = args[1]
                return old_
            if name == "get":
                return r.get(key_)
            if name == "contains_key":
                return key_ in r
            if name == "remove":
                return r.pop(key_, None)
            return len(r) if name == "len" else not r
        if isinstance(r, RayonScope):
            if name == "spawn":
                self.call_closure(args[0], [r])
                return None
            raise self.err(src, line, f"rayon scope .{name}")
        if isinstance(r, ChunksMut):
            if name == "enumerate":
                return ChunksMut(r.base, r.n, True, r.left)
            if name == "for_each":
                r.run(lambda item: self.call_closure(args[0], [item]))
                return None
            if name == "len":
                return (len(r.base) + r.n - 1) // r.n
            raise self.err(src, line, f"chunks_mut().{name}")
        if isinstance(r, Cursor):
            if name == "write_all":
                return r.write_all(args[0])
            if name == "get_ref":
                return r.data
            raise self.err(src, line, f"Cursor::{name}")
        if name == "to_vec" and isinstance(r, Struct) and "__name__" in r:         # HashOut::to_vec (hash_types.rs:80), not the slice adaptor
            v = self.call_assoc(r["__name__"], name, args, r.get("__file__", src.path), self_val=r, has_self=True)
            if v is not NOT_FOUND:
                return v
        # ---- adaptors that do nothing here
        if name in ("iter", "into_iter", "iter_mut", "copied", "cloned", "collect", "collect_vec", "to_vec", "try_into", "unwrap", "expect", "by_ref",
                    "as_ref", "as_mut", "borrow", "borrow_mut", "as_mut_slice", "as_ptr", "as_mut_ptr", "into", "to_owned", "as_slice_of_cells", "peekable", "into_par_iter", "par_iter", "par_iter_mut", "unwrap_or_else", "unwrap_or_default"):
            if name in ("unwrap", "expect") and r is None:
                raise self.err(src, line, "unwrap of None")
            if name == "to_vec" or name == "collect" or name == "collect_vec":
                return list(self.iterate(r, src, line)) if isinstance(r, (list, tuple, Rng)) else r
            if isinstance(r, Rng) and (r.end - r.start) > 1 << 24:
                return r                            # 0..=p-1: only ever searched (`find_any`), never collected
            if isinstance(r, Rng) and name in ("iter", "into_iter"):
                return r.items()
            if isinstance(r, list) and name in ("iter", "into_iter"):
                return list(r)                      # an iterator is its own object: `.next()` on it must not eat the vector it came from
            return r
        if name == "clone":
            return clone(r)
        if isinstance(r, Fe):
            if name == "square":
                return Fe(r.v * r.v)
            if name == "cube":
                return Fe(r.v * r.v * r.v)
            if name == "double":
                return Fe(2 * r.v)
            if name == "is_zero":
                return r.v == 0
            if name == "is_one":
                return r.v == 1
            if name == "as_slice":
                return [r]
            if name in ("to_canonical_u64", "to_noncanonical_u64"):
                return TInt(r.v, 64)
            if name == "to_canonical":
                return r
            if name in ("exp_u64", "exp_power_of_2"):
                return Fe(pow(r.v, args[0] if name == "exp_u64" else 1 << args[0], P))
            if name == "inverse":
                if r.v == 0:
                    raise self.err(src, line, "inverse of zero")
                return Fe(pow(r.v, P - 2, P))
            if name == "try_inverse":
                return None if r.v == 0 else Fe(pow(r.v, P - 2, P))
            if name in ("mul", "add", "sub", "div"):
                return self.binop({"mul": "*", "add": "+", "sub": "-", "div": "/"}[name], r, args[0], src, line)
            if name == "scalar_mul":
                return Fe(r.v * args[0].v)
            if name == "powers":
                return Powers(r)
            if name == "add_canonical_u64":
                return Fe(r.v + args[0])
            if name == "to_basefield_array":
                return [r, Fe(0)] if self.extension else [r]
        if isinstance(r, Fe2):
            if name == "square":
                return r.mul(r)
            if name == "double":
                return r.add(r)
            if name == "cube":
                return r.mul(r).mul(r)
            if name == "is_zero":
                return r.a == 0 and r.b == 0
            if name in ("exp_u64", "exp_power_of_2"):
                return r.pow(args[0] if name == "exp_u64" else 1 << args[0])
            if name == "inverse":
                return r.inverse()
            if name == "try_inverse":
                return None if (r.a == 0 and r.b == 0) else r.inverse()
            if name in ("mul", "add", "sub", "div"):
                return self.binop({"mul": "*", "add": "+", "sub": "-", "div": "/"}[name], r, args[0], src, line)
            if name == "scalar_mul":
                return Fe2(r.a * args[0].v, r.b * args[0].v)
            if name == "to_basefield_array":
                return [Fe(r.a), Fe(r.b)]
            if name == "powers":
                return Powers(r)
            if name == "as_slice":
                return [r]
        if isinstance(r, bool):
            if name == "then":
                return Opt(self.call_closure(args[0], [])) if r else None
            if name == "then_some":
                return Opt(args[0]) if r else None
        if isinstance(r, int) and not isinstance(r, bool):
            if name == "pow":
                return r ** args[0]
            if name in ("min", "max"):
                return min(r, args[0]) if name == "min" else max(r, args[0])
            if name in ("add", "sub", "mul", "div"):
                return self.binop({"add": "+", "sub": "-", "mul": "*", "div": "/"}[name], r, args[0], src, line)
            if name in ("wrapping_add", "saturating_sub", "checked_sub"):
                return r + args[0] if name == "wrapping_add" else max(0, r - args[0])
            if name == "to_le_bytes":
                if not isinstance(r, TInt):
                    raise self.err(src, line, "to_le_bytes of an integer whose type the interpreter does not know")
                return [TInt((r >> (8 * i)) & 0xFF, 8) for i in range(r.bits // 8)]
            if name == "leading_zeros":
                if not 0 <= r < 1 << 64:
                    raise self.err(src, line, "leading_zeros of a value that is not a u64")
                return 64 - r.bit_length()
            if name == "bits":
                return r.bit_length()
            if name == "reverse_bits":              # of a usize (64 bits)
                return int(format(r, "064b")[::-1], 2)
            if name == "overflowing_shr":           # (value shifted by the amount mod 64, whether the amount was >= 64)
                return ((r >> (args[0] % 64)) & ((1 << 64) - 1), args[0] >= 64)
            if name == "count_ones":
                return bin(r).count("1")
            if name == "count_zeros":               # of a usize / u64 unless the value carries its type
                return (r.bits if isinstance(r, TInt) else 64) - bin(r).count("1")
            if name == "trailing_zeros":
                return (r & -r).bit_length() - 1
            if name in ("cmp", "partial_cmp") and isinstance(args[0], int):
                return Enum("Ordering", "Less" if r < args[0] else "Greater" if r > args[0] else "Equal")
            if name == "next_power_of_two":
                return 1 if r <= 1 else 1 << (r - 1).bit_length()
            if name == "is_power_of_two":
                return r > 0 and r & (r - 1) == 0
        if r is None or (not isinstance(r, (list, tuple, Rng, Struct, Enum, Fe, Fe2, int, RepeatForever))):
            if name in ("is_some", "is_none"):
                return (r is not None) == (name == "is_some")
        if name in ("is_some", "is_none"):
            return (r is not None) == (name == "is_some")
        if isinstance(r, tuple) and name == "to_basefield_array":
            return list(r)
        if isinstance(r, Rng) and name in ("find_any", "find_first", "find"):
            # rayon's find_any may return any match; the first one is the choice this repository's provers make (the minimal witness)
            if self.find_any_hint is not None:
                # a quick re-run: the hinted value must satisfy the closure and a spread of smaller ones must not (the full search from 0
                # was done when the record was made)
                w = self.find_any_hint
                if not (r.start <= w < r.end) or not self.truthy(self.call_closure(args[0], [w])):
                    raise self.err(src, line, "the hinted witness does not satisfy the search")
                for j in range(1, 65):
                    c = r.start + (w - r.start) * j // 65
                    if c < w and self.truthy(self.call_closure(args[0], [c])):
                        raise self.err(src, line, "a smaller witness than the hinted one exists")
                return Opt(TInt(w, 64))
            for i_ in range(r.start, r.end):
                if self.truthy(self.call_closure(args[0], [i_])):
                    return Opt(TInt(i_, 64))
            return None
        if isinstance(r, Rng) and name in ("contains",):
            return r.start <= args[0] < r.end
        if isinstance(r, Powers):
            if name == "map":
                return Powers(r.base, r.maps + (args[0],))
            if name == "take":
                return self.powers(r, args[0], src, line)
            if name == "zip":
                other = self.iterate(args[0], src, line)
                return [(w, o) for w, o in zip(self.powers(r, len(other), src, line), other)]
            raise self.err(src, line, f"powers().{name}")
        if isinstance(r, (list, tuple, Rng, RepeatForever)):
            if isinstance(r, RepeatForever):
                if name == "take":
                    return [clone(r.v) for _ in range(args[0])]
                if name == "zip":
                    other = self.iterate(args[0], src, line)
                    return [(clone(r.v), o) for o in other]
                raise self.err(src, line, f"iter::repeat().{name}")
            items = self.iterate(r, src, line)
            if name == "len":
                return len(items)
            if name == "rev":
                return items[::-1]
            if name == "enumerate":
                return [(i, x) for i, x in enumerate(items)]
            if name in ("sorted_unstable_by_key", "sorted_by_key", "sorted_by_cached_key"):
                def key_(x):
                    k_ = self.call_closure(args[0], [x])
                    return k_.v if isinstance(k_, Fe) else k_
                return sorted(items, key=key_)
            if name == "zip_eq":                    # itertools: zip that panics on different lengths
                other = self.iterate(args[0], src, line)
                if len(other) != len(items):
                    raise self.err(src, line, f"zip_eq of {len(items)} and {len(other)} items")
                return [tuple(t) for t in zip(items, other)]
            if name == "zip":
                o = args[0]
                if isinstance(o, ChunksMut) and o.left is None and not o.enumerate_:
                    return ChunksMut(o.base, o.n, False, items)
                if isinstance(o, RepeatForever):
                    return [(x, clone(o.v)) for x in items]
                if isinstance(o, Powers):
                    return [(x, w) for x, w in zip(items, self.powers(o, len(items), src, line))]
                return [tuple(t) for t in zip(items, self.iterate(o, src, line))]
            if name == "chain":
                o = args[0]
                if o is None or isinstance(o, (Fe, Fe2, int, Struct)):      # an Option is an iterator of zero or one item (`Some(x)` is x here)
                    return items + ([] if o is None else [o])
                return items + self.iterate(o, src, line)
            if name == "skip":
                return items[args[0]:]
            if name == "take":
                return items[:args[0]]
            if name == "step_by":
                return items[::args[0]]
            if name == "map":
                return [self.call_closure(args[0], [x]) for x in items]
            if name == "for_each":
                for x in items:
                    self.call_closure(args[0], [x])
                return None
            if name == "filter":
                return [x for x in items if self.truthy(self.call_closure(args[0], [x]))]
            if name in ("flat_map", "flat_map_iter"):
                out = []
                for x in items:
                    out += self.iterate(self.call_closure(args[0], [x]), src, line)
                return out
            if name == "flatten":
                out = []
                for x in items:
                    out += self.iterate(x, src, line)
                return out
            if name == "all":
                return all(self.truthy(self.call_closure(args[0], [x])) for x in items)
            if name == "any":
                return any(self.truthy(self.call_closure(args[0], [x])) for x in items)
            if name == "sum":
                if items and isinstance(items[0], Struct) and "__name__" in items[0]:
                    r_ = self.call_assoc(items[0]["__name__"], "sum", [items], src.path)       # impl Sum for PolynomialCoeffs<F>
                    if r_ is not NOT_FOUND:
                        return r_
                if not items:
                    return ZeroSum(0)
                acc = items[0]
                for x in items[1:]:
                    acc = self.binop("+", acc, x, src, line)
                return acc
            if name == "product":
                if not items:
                    return OneProduct(1)
                acc = items[0]
                for x in items[1:]:
                    acc = self.binop("*", acc, x, src, line)
                return acc
            if name == "scan":
                # .scan(init, |state, x| { *state = ...; Some(y) }): the closure's first parameter is the running state
                c, state, out = args[1], args[0], []
                for x in items:
                    env = dict(c.env)
                    self.bind(c.params[0], state, env, c.src)
                    self.bind(c.params[1], x, env, c.src)
                    y = self.ev(c.body, env, c.src)
                    state = env[pattern_names(c.params[0])[0]]
                    if y is None:
                        break
                    out.append(y)
                return out
            if name == "fold":
                acc = args[0]
                for x in items:
                    acc = self.call_closure(args[1], [acc, x])
                return acc
            if name == "reduce":
                acc = items[0]
                for x in items[1:]:
                    acc = self.call_closure(args[0], [acc, x])
                return acc
            if name == "next" and isinstance(r, list):
                return r.pop(0) if r else None
            if name == "unzip":
                return [x[0] for x in items], [x[1] for x in items]
            if name == "cartesian_product":
                other = self.iterate(args[0], src, line)
                return [(x, y) for x in items for y in other]
            if name == "count":
                return len(items)
            if name == "unique":
                out = []
                for x in items:
                    if not any(same(x, y) for y in out):
                        out.append(x)
                return out
            if name in ("first", "last"):
                return (items[0] if name == "first" else items[-1]) if items else None
            if name in ("chunks", "chunks_exact", "par_chunks", "par_chunks_exact"):
                return [items[i:i + args[0]] for i in range(0, len(items), args[0])]
            if name in ("chunks_mut", "chunks_exact_mut", "par_chunks_mut", "par_chunks_exact_mut") and isinstance(r, list):
                return ChunksMut(r, args[0])
            if name == "reverse" and isinstance(r, list):
                r.reverse()
                return None
            if name == "swap" and isinstance(r, list):
                r[args[0]], r[args[1]] = r[args[1]], r[args[0]]
                return None
            if name == "set_len" and isinstance(r, list):          # after Vec::with_capacity: uninitialised slots
                if len(r) > args[0]:
                    del r[args[0]:]
                r.extend([None] * (args[0] - len(r)))
                return None
            if name == "resize" and isinstance(r, list):
                if len(r) > args[0]:
                    del r[args[0]:]
                r.extend([clone(args[1]) for _ in range(args[0] - len(r))])
                return None
            if name == "insert" and isinstance(r, list):
                r.insert(args[0], args[1])
                return None
            if name == "windows":
                return [items[i:i + args[0]] for i in range(0, len(items) - args[0] + 1)]
            if isinstance(r, list) and name in ("clear", "pop", "drain", "extend_from_slice", "shrink_to_fit", "copy_from_slice", "truncate", "reserve", "reserve_exact"):
                if name == "clear":
                    del r[:]
                    return None
                if name == "pop":
                    return r.pop() if r else None
                if name == "drain":
                    out = list(r)
                    del r[:]
                    return out
                if name == "extend_from_slice":
                    r.extend(self.iterate(args[0], src, line))
                    return None
                if name == "copy_from_slice":
                    r[:] = self.iterate(args[0], src, line)
                    return None
                if name == "truncate":
                    del r[args[0]:]
                    return None
                return None
            if name == "push" and isinstance(r, list):
                r.append(args[0])
                return None
            if name == "extend" and isinstance(r, list):
                r.extend(self.iterate(args[0], src, line))
                return None
            if name == "contains":
                return any(same(x, args[0]) for x in items)
            if name == "concat":
                out = []
                for x in items:
                    out += list(x)
                return out
            if name == "as_slice":
                return items
            if name == "is_empty":
                return not items
        if isinstance(r, Struct) and "__name__" in r:
            for ty in (r["__name__"], r.get("__trait__")):
                if ty:
                    v = self.call_assoc(ty, name, args, r.get("__file__", src.path), self_val=r, has_self=True)
                    if v is not NOT_FOUND:
                        return v
        if isinstance(r, Enum):
            v = self.call_assoc(r.ty, name, args, src.path, self_val=r, has_self=True)
            if v is not NOT_FOUND:
                return v
        # ---- a method of the reference (on `self` structs, enums): found by name
        target = self.find_fn_file(name, src.path)
        if target is not None:
            params, _, _ = self.fn_ast(target, name)
            if params and params[0] == ("pid", "self"):
                return self.call_fn(target, name, args, r)
        raise self.err(src, line, f"method `.{name}()` on {type(r).__name__}")


class RepeatForever:
    def __init__(self, v):
        self.v = v


class Powers:
    """F::powers(): 1, b, b^2, ... -- infinite, so `.map(f)` is kept pending until a finite iterator is zipped on or `.take(n)` cuts it"""

    def __init__(self, base, maps=()):
        self.base, self.maps = base, tuple(maps)


class ZeroSum(int):
    """the sum of an empty iterator: 0 of whatever type the context wants"""


class OneProduct(int):
    """the product of an empty iterator"""


def clone(v):
    if isinstance(v, Opt):
        return Opt(clone(v.v))
    if isinstance(v, list):
        return [clone(x) for x in v]
    if isinstance(v, Struct):
        s = Struct()
        for k, x in v.items():
            s[k] = clone(x)
        return s
    return v


def same(a, b):
    if isinstance(a, Fe) and isinstance(b, Fe):
        return a.v == b.v
    return a == b


def pattern_names(pat):
    k = pat[0]
    if k == "pid":
        return [pat[1]]
    if k in ("ptuple",):
        return [n for p in pat[1] for n in pattern_names(p)]
    if k == "pcall":
        return [n for p in pat[2] for n in pattern_names(p)]
    if k == "pstruct":
        return [n for _, p in pat[2] for n in pattern_names(p)]
    return []


# ------------------------------------------------------------------------------------------------ driving it
def make_ref(reference):
    ref = X.Ref(reference)
    for rel in ("plonky2/plonky2/src/plonk/plonk_common.rs", "plonky2/plonky2/src/util/reducing.rs", "plonky2/util/src/lib.rs"):
        extra = os.path.join(reference, rel)
        if os.path.exists(extra) and extra not in ref.files:
            ref.files.append(extra)
    # constants of the added file
    return ref


def rows_for(seed, table, ncols):
    """deterministic rows: the fixture stores the recipe, not the rows.  seed = int: pseudo-random (splitmix64, reduced mod p);
    seed = {"fill": v}: every cell of both rows is v (0, 1, p - 1, and the reference's ADDR_HEAP_PTR: the rows on which the memory
    table's `is_zero` branch, memory_stark.rs:292-300, takes its other side)"""
    if isinstance(seed, dict):
        return [seed["fill"] % P] * ncols, [seed["fill"] % P] * ncols
    x = (seed * 0x9E3779B97F4A7C15 + table * 0xD1342543DE82EF95 + 0x1234567) & (2**64 - 1)
    out = []
    for _ in range(2 * ncols):
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        z ^= z >> 31
        out.append(z % P)
    return out[:ncols], out[ncols:]


PARAM = 0x0123456789ABCDEF % P          # the compress challenge handed to the bitwise and program tables


def eval_table(it, ref, index, rel, lv, nv):
    """-> [(kind, value, file, line)] of the reference's eval_packed_generic of table `index` on rows (lv, nv)"""
    path = os.path.join(ref.root, rel)
    src = X.Src.get(path)
    T = src.toks
    # the trait impl's eval_packed_generic: the one that takes &self
    fn = None
    for i, (t, _) in enumerate(T):
        if t == "fn" and T[i + 1][0] == "eval_packed_generic":
            j = i + 2
            while T[j][0] != "(":
                j += 1
            if T[j + 1][0] == "&" and T[j + 2][0] == "self":
                fn = (j, src.match(j))
                break
    if fn is None:
        raise RustError(f"{path}: no eval_packed_generic(&self, ...)")
    k = fn[1] + 1
    while T[k][0] != "{":
        k += 1
    body = Parser(src, k, src.match(k) + 1).block()
    cons = Consumer()
    lift = (lambda x: x) if (lv and isinstance(lv[0], Fe2)) else Fe
    vars_ = Struct({"__name__": "StarkEvaluationVars", "local_values": [lift(x) for x in lv], "next_values": [lift(x) for x in nv], "public_inputs": []})
    me = Struct({"__name__": "Stark", "compress_challenge": Fe(PARAM), "_phantom": None})
    env = {"__src__": src, "self": me, "vars": vars_, "yield_constr": cons}
    # the consumer's parameter may have another name
    pp = [t for t, _ in T[fn[0] + 1:fn[1]]]
    for a in range(len(pp)):
        if pp[a] == ":" and "ConstraintConsumer" in pp[a:a + 8] and a >= 1:
            env[pp[a - 1]] = cons
        if pp[a] == ":" and "StarkEvaluationVars" in pp[a:a + 4] and a >= 1:
            env[pp[a - 1]] = vars_
    try:
        it.ev(body, env, src)
    except Return:
        pass
    return cons.emits


def eval_ctl_entry(it, ref, entry, row):
    """One TableWithColumns of stark/ola_stark.rs: its `ctl_data_*` / `ctl_filter_*` functions are CALLED (they build `Column`s with
    the reference's constructors, cross_table_lookup.rs:34-98), and every column is evaluated on `row` with the reference's own
    `Column::eval` (:100-109) -> ([values of the data columns], value of the filter or None)"""
    ctl_rs = os.path.join(ref.root, "stark", "cross_table_lookup.rs")

    def call(fn_str, rel):
        m = re.match(r"^(\w+)\((\d*)\)$", fn_str)
        args = [int(m.group(2))] if m.group(2) else []
        return it.call_fn(os.path.join(ref.root, rel), m.group(1), args)

    vals = [Fe(x) for x in row]
    cols = it.iterate(call(entry["data_fn"], entry["data_file"]), X.Src.get(ctl_rs))
    data = [it.call_fn(ctl_rs, "eval", [vals], c).v for c in cols]
    filt = None
    if "filter_fn" in entry:
        filt = it.call_fn(ctl_rs, "eval", [vals], call(entry["filter_fn"], entry["filter_file"])).v
    return data, filt


def stream_for(seed, salt, count):
    """`count` deterministic field elements (splitmix64 mod p) -- Z values and challenges of the vanishing-polynomial vectors"""
    m = 2**64 - 1
    x = (seed * 0x9E3779B97F4A7C15 + salt * 0xD1342543DE82EF95 + 0x7654321) & m
    out = []
    for _ in range(count):
        x = (x + 0x9E3779B97F4A7C15) & m
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        z ^= z >> 31
        out.append(z % P)
    return out


NUM_CHALLENGES = 2
Z_PER_TABLE = 256          # Z openings handed to from_proofs per table: more than any table has (it takes what it needs, in its order)


def stark_struct(it, ref, rel):
    """a value standing for `&S` (the table's Stark impl): methods resolve through the impl blocks of the table's file, then the trait's defaults"""
    path = os.path.join(ref.root, rel)
    ty = [t for (t, n) in it.impl_index(path) if n == "eval_packed_generic"][0]
    return Struct({"__name__": ty, "__trait__": "Stark", "__file__": path, "compress_challenge": Fe(PARAM), "_phantom": None})


def eval_vanishing(it, ref, seed, golden):
    """`eval_vanishing_poly` (vanishing_poly.rs:20-45) of every table on one set of inputs: the table's AIR, then `eval_permutation_checks`
    (permutation.rs:302-360), then `eval_cross_table_lookup_checks` (cross_table_lookup.rs:380-421), with the per-table `CtlCheckVars` built by
    the reference's own `CtlCheckVars::from_proofs` (:330-378) from `all_cross_table_lookups()` (ola_stark.rs:122-560) -- so the ORDER in which a
    table's lookup Z columns are consumed is the reference's, too.  Inputs: rows_for(seed, t, columns); Z openings stream_for(seed, 100 + t,
    2 * Z_PER_TABLE) (local then next); lookup challenges stream_for(seed, 99, 4) = (beta, gamma) x 2; permutation challenge sets of table t
    stream_for(seed, 200 + t, ...) = [batch slot][challenge](beta, gamma).  -> per table [(kind, value)]"""
    stark_dir = os.path.join(ref.root, "stark")
    ola = os.path.join(stark_dir, "ola_stark.rs")
    config = Struct({"__name__": "StarkConfig", "num_challenges": NUM_CHALLENGES, "security_bits": 100})
    starks = [stark_struct(it, ref, rel) for _, rel in X.TABLES]
    ctls = it.call_fn(ola, "all_cross_table_lookups", [])
    cc = stream_for(seed, 99, 2 * NUM_CHALLENGES)
    ctl_challenges = Struct({"__name__": "GrandProductChallengeSet", "challenges": [
        Struct({"__name__": "GrandProductChallenge", "beta": Fe(cc[2 * c]), "gamma": Fe(cc[2 * c + 1])}) for c in range(NUM_CHALLENGES)]})
    nperm = [it.method(st, "num_permutation_batches", [config], X.Src.get(ola), 0) for st in starks]
    proofs, zs = [], []
    for t in range(len(starks)):
        z = stream_for(seed, 100 + t, 2 * Z_PER_TABLE)
        zs.append((z[:Z_PER_TABLE], z[Z_PER_TABLE:]))
        proofs.append(Struct({"__name__": "StarkProof", "openings": Struct({"__name__": "StarkOpeningSet", "permutation_ctl_zs": [Fe(x) for x in z[:Z_PER_TABLE]],
                                                                            "permutation_ctl_zs_next": [Fe(x) for x in z[Z_PER_TABLE:]]})}))
    r = it.call_assoc("CtlCheckVars", "from_proofs", [proofs, ctls, ctl_challenges, nperm], os.path.join(stark_dir, "cross_table_lookup.rs"))
    if r is NOT_FOUND:
        raise RustError("CtlCheckVars::from_proofs not found")
    out = []
    for t, (st, (_, rel)) in enumerate(zip(starks, X.TABLES)):
        ncols = golden["tables"][t]["columns"]
        lv, nv = rows_for(seed, t, ncols)
        vars_ = Struct({"__name__": "StarkEvaluationVars", "local_values": [Fe(x) for x in lv], "next_values": [Fe(x) for x in nv], "public_inputs": []})
        pv = None
        if it.method(st, "uses_permutation_args", [], X.Src.get(ola), 0):
            bs = it.method(st, "permutation_batch_size", [], X.Src.get(ola), 0)
            pc = stream_for(seed, 200 + t, 2 * bs * NUM_CHALLENGES)
            sets = [Struct({"__name__": "GrandProductChallengeSet", "challenges": [
                Struct({"__name__": "GrandProductChallenge", "beta": Fe(pc[2 * (i * NUM_CHALLENGES + c)]), "gamma": Fe(pc[2 * (i * NUM_CHALLENGES + c) + 1])})
                for c in range(NUM_CHALLENGES)]}) for i in range(bs)]
            pv = Struct({"__name__": "PermutationCheckVars", "local_zs": [Fe(x) for x in zs[t][0][:nperm[t]]], "next_zs": [Fe(x) for x in zs[t][1][:nperm[t]]],
                         "permutation_challenge_sets": sets})
        cons = Consumer()
        it.call_fn(os.path.join(stark_dir, "vanishing_poly.rs"), "eval_vanishing_poly", [st, config, vars_, pv, r[t], cons])
        out.append({"kinds": [e[0] for e in cons.emits], "values": [e[1] for e in cons.emits], "num_permutation_zs": nperm[t], "num_ctl_zs": len(r[t])})
    return out


def extract(reference, points):
    ref = make_ref(reference)
    it = Interp(ref)
    golden = json.load(open(X.FIXTURE))
    heap_ptr = it.const_value("ADDR_HEAP_PTR", X.Src.get(os.path.join(ref.root, "memory", "memory_stark.rs")))
    seeds = list(range(points)) + [{"fill": 0}, {"fill": 1}, {"fill": P - 1}, {"fill": int(heap_ptr)}]
    tables, problems = [], []
    for index, (name, rel) in enumerate(X.TABLES):
        ncols = golden["tables"][index]["columns"]
        rec = {"table": name, "file": rel, "columns": ncols, "points": []}
        try:
            for s in seeds:
                lv, nv = rows_for(s, index, ncols)
                emits = eval_table(it, ref, index, rel, lv, nv)
                rec["points"].append({"seed": s, "kinds": [e[0] for e in emits], "values": [e[1] for e in emits]})
            rec["emit_sites"] = ["%s:%d" % (e[2], e[3]) for e in emits]
        except (RustError, RecursionError, IndexError, KeyError, TypeError, AttributeError, AssertionError) as e:
            problems.append({"table": name, "reason": "%s: %s" % (type(e).__name__, e)})
            rec = {"table": name, "file": rel, "columns": ncols, "unevaluated": "%s: %s" % (type(e).__name__, e)}
        tables.append(rec)
    index_of = {name: i for i, (name, _) in enumerate(X.TABLES)}
    ctls = []
    for c in golden["cross_table_lookups"]:
        rec = {"name": c["name"], "entries": []}
        try:
            for entry in [c["looked"]] + c["looking_in_source_order"]:
                t = index_of[entry["table"]]
                pts = []
                for sd in list(range(points)) + [{"fill": 1}]:
                    lv, _ = rows_for(sd, t, golden["tables"][t]["columns"])
                    data, filt = eval_ctl_entry(it, ref, entry, lv)
                    pts.append({"seed": sd, "data": data, "filter": filt})
                rec["entries"].append({"table": entry["table"], "data_fn": entry["data_fn"], "filter_fn": entry.get("filter_fn"), "points": pts})
        except (RustError, RecursionError, IndexError, KeyError, TypeError, AttributeError, AssertionError) as e:
            problems.append({"table": c["name"], "reason": "%s: %s" % (type(e).__name__, e)})
            rec = {"name": c["name"], "unevaluated": "%s: %s" % (type(e).__name__, e)}
        ctls.append(rec)
    vanishing = []
    try:
        for sd in range(points):
            vanishing.append({"seed": sd, "tables": eval_vanishing(it, ref, sd, golden)})
    except (RustError, RecursionError, IndexError, KeyError, TypeError, AttributeError, AssertionError) as e:
        problems.append({"table": "eval_vanishing_poly", "reason": "%s: %s" % (type(e).__name__, e)})
    return {"generator": "tools/rust_air_eval.py", "cross_table_lookups": ctls, "vanishing_poly": vanishing, "what": "values of the constraints the reference's eval_packed_generic emits on pseudo-random rows "
            "(rows_for(seed, table, columns): splitmix64 mod p; compress challenge PARAM), obtained by interpreting the reference's Rust source",
            "param": PARAM, "tables": tables, "problems": problems}


PRIMITIVES_FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_primitive_vectors.json")


def plonky2_interp(reference):
    """an interpreter whose `H` is PoseidonHash and whose permutation is the interpreted `Poseidon::poseidon_naive` (hash/poseidon.rs:617)"""
    it = Interp(make_ref(reference))
    base = os.path.join(reference, "plonky2", "plonky2", "src")
    it.plonky2 = base
    pos = os.path.join(base, "hash", "poseidon.rs")
    it.extra_files = [os.path.join(base, "fri", "mod.rs"), os.path.join(base, "fri", "reduction_strategies.rs"), os.path.join(base, "iop", "challenger.rs"),
                      os.path.join(base, "hash", "merkle_proofs.rs")]
    it.generics = {"H": ["PoseidonHash", "Hasher"], "OH": ["PoseidonHash", "Hasher"]}
    it.permutation_hook = lambda st, segs=None: it.call_assoc("Poseidon", "poseidon_naive", [list(st)], pos)
    return it


def hash_out(d):
    return Struct({"__name__": "HashOut", "elements": [Fe(int(x)) for x in d]})


def verify_merkle_proof_to_cap(it, leaf, index, cap, siblings):
    """hash/merkle_proofs.rs:46 interpreted; True when it returns Ok(())"""
    r = it.call_free(os.path.join(it.plonky2, "hash", "merkle_proofs.rs"), "verify_merkle_proof_to_cap",
                     [[Fe(int(x)) for x in leaf], int(index), Struct({"__name__": "MerkleCap", 0: [hash_out(c) for c in cap]}),
                      Struct({"__name__": "MerkleProof", "siblings": [hash_out(x) for x in siblings]})])
    return not isinstance(r, Enum)


def primitives(reference):
    """Outputs of the reference's own hashing, transcript and FRI-parameter code, interpreted: the vectors the oracle and the product's host code
    are compared with (tests/test_ref_primitives.py)."""
    it = plonky2_interp(reference)
    base = it.plonky2
    pos = os.path.join(base, "hash", "poseidon.rs")
    hashing = os.path.join(base, "hash", "hashing.rs")
    chal = os.path.join(base, "iop", "challenger.rs")
    out = {"generated_by": "tools/rust_air_eval.py --primitives", "sources": {}}

    # ---- the permutation (poseidon.rs:617 poseidon_naive: constant_layer, sbox_layer, mds_layer as written there)
    inputs = [[0] * 12, [P - 1] * 12, list(range(12))] + [stream_for(900 + k, 1, 12) for k in range(5)]
    out["poseidon"] = [{"input": v, "output": [x.v for x in it.call_assoc("Poseidon", "poseidon_naive", [[Fe(x) for x in v]], pos)]} for v in inputs]
    out["sources"]["poseidon"] = "plonky2/plonky2/src/hash/poseidon.rs: Poseidon::poseidon_naive"

    # ---- sponge hashing (hashing.rs:93 hash_n_to_m_no_pad, :117 hash_n_to_hash_no_pad, :77 compress) through PoseidonHash (poseidon.rs:640)
    out["hash_no_pad"] = []
    for n in (0, 1, 4, 5, 7, 8, 9, 15, 16, 17, 29, 76, 135):
        v = stream_for(1000 + n, 2, n)
        h = it.call_assoc("PoseidonHash", "hash_no_pad", [[Fe(x) for x in v]], pos)
        out["hash_no_pad"].append({"input": v, "digest": [x.v for x in h["elements"]]})
    out["two_to_one"] = []
    for k in range(4):
        l, r = stream_for(1100 + k, 3, 4), stream_for(1100 + k, 4, 4)
        h = it.call_assoc("PoseidonHash", "two_to_one", [hash_out(l), hash_out(r)], pos)
        out["two_to_one"].append({"left": l, "right": r, "digest": [x.v for x in h["elements"]]})
    out["sources"]["hash_no_pad"] = "plonky2/plonky2/src/hash/poseidon.rs: PoseidonHash::hash_no_pad -> hashing.rs: hash_n_to_hash_no_pad"
    out["sources"]["two_to_one"] = "plonky2/plonky2/src/hash/poseidon.rs: PoseidonHash::two_to_one -> hashing.rs: compress"

    # ---- the transcript (iop/challenger.rs:36-162): scripts of observe / observe_cap / get / compact
    scripts = [
        [["get", 1]],
        [["observe", 1], ["get", 1]],
        [["observe", 8], ["get", 3], ["get", 6], ["observe", 3], ["get", 2]],
        [["observe", 7], ["observe", 1], ["observe", 1], ["get", 9], ["compact"], ["get", 1]],
        [["observe_cap", 16], ["get", 2], ["compact"], ["observe_cap", 16], ["get", 2], ["observe", 20], ["get", 2], ["compact"], ["observe", 5], ["compact"], ["get", 4]],
        [["observe", 23], ["compact"], ["compact"], ["get", 8], ["get", 1], ["observe", 8], ["observe", 8], ["get", 1]],
    ]
    out["challenger"] = []
    for k, script in enumerate(scripts):
        c = it.call_assoc("Challenger", "new", [], chal)
        ops, outputs = [], []
        for j, op in enumerate(script):
            if op[0] == "observe":
                v = stream_for(1200 + k, j, op[1])
                it.call_assoc("Challenger", "observe_elements", [c, [Fe(x) for x in v]], chal)
                ops.append(["observe", v])
            elif op[0] == "observe_cap":
                v = stream_for(1200 + k, j, 4 * op[1])
                cap = Struct({"__name__": "MerkleCap", 0: [hash_out(v[4 * i:4 * i + 4]) for i in range(op[1])]})
                it.call_assoc("Challenger", "observe_cap", [c, cap], chal)
                ops.append(["observe_cap", v])
            elif op[0] == "get":
                got = it.call_assoc("Challenger", "get_n_challenges", [c, op[1]], chal)
                ops.append(["get", op[1]])
                outputs.append([x.v for x in got])
            else:
                st = it.call_assoc("Challenger", "compact", [c], chal)
                ops.append(["compact"])
                outputs.append([x.v for x in st])
        out["challenger"].append({"ops": ops, "outputs": outputs, "state": [x.v for x in c["sponge_state"]]})
    out["sources"]["challenger"] = "plonky2/plonky2/src/iop/challenger.rs: Challenger::{new, observe_elements, observe_cap, get_n_challenges, compact}"

    # ---- the lookup argument's column generator (circuits/src/stark/lookup.rs:68-132 permuted_cols)
    lookup = os.path.join(reference, "circuits", "src", "stark", "lookup.rs")
    out["permuted_cols"] = []
    for n in (1, 2, 3, 8, 37, 64):
        r = stream_for(1300 + n, 0, 4 * n)
        fixed = list(range(n))
        cases = [("lookup", [x % n for x in r[:n]], fixed),
                 ("few-distinct", [x % max(1, n // 8) for x in r[:n]], fixed),
                 ("all-max", [n - 1] * n, fixed), ("all-min", [0] * n, fixed),
                 ("dup-table", [x % (2 * n + 3) for x in r[:n]], sorted(3 * (x % max(2, n // 2)) for x in r[n:2 * n])),
                 ("inputs-above", [5 * n + x % (n + 1) for x in r[:n]], fixed), ("inputs-below", [x % 3 for x in r[:n]], [x + 10 for x in fixed]),
                 ("blocks", [4 * (i // 4) for i in range(n)], fixed),
                 ("wide", [r[2 * n + (x % n)] for x in r[:n]], r[2 * n:3 * n][::-1])]
        for name, inputs, table in cases:
            got = it.call_free(lookup, "permuted_cols", [[Fe(x) for x in inputs], [Fe(x) for x in table]])
            out["permuted_cols"].append({"case": "%s n=%d" % (name, n), "inputs": inputs, "table": table,
                                         "permuted_inputs": [x.v for x in got[0]], "permuted_table": [x.v for x in got[1]]})
    out["sources"]["permuted_cols"] = "circuits/src/stark/lookup.rs: permuted_cols"

    # ---- rows of the Poseidon table (core/src/crypto/poseidon_trace.rs:79 calculate_poseidon_and_generate_intermediate_trace -- the fast
    #      partial rounds of core/src/util/poseidon_utils.rs -- then circuits/src/generation/poseidon.rs:5 generate_poseidon_trace with its padding)
    ptrace = os.path.join(reference, "core", "src", "crypto", "poseidon_trace.rs")
    gen = os.path.join(reference, "circuits", "src", "generation", "poseidon.rs")
    flags = ("filter_looked_normal", "filter_looked_treekey", "filter_looked_storage", "filter_looked_storage_branch")
    cells, meta = [], []
    for k, filt in enumerate([(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)]):
        inp = [0] * 12 if k == 0 else [P - 1] * 12 if k == 1 else stream_for(1500 + k, 0, 12)
        cell = it.call_free(ptrace, "calculate_poseidon_and_generate_intermediate_trace", [[Fe(x) for x in inp]])
        for name, f in zip(flags, filt):
            cell[name] = bool(f)
        cells.append(cell)
        meta.append({"input": inp, "filters": list(filt)})
    cols = it.call_free(gen, "generate_poseidon_trace", [cells])          # 5 rows -> padded to 8
    out["poseidon_table"] = {"columns": len(cols), "rows": len(cols[0]), "live": meta, "trace_rows": [[c[i].v for c in cols] for i in range(len(cols[0]))]}
    out["sources"]["poseidon_table"] = "core/src/crypto/poseidon_trace.rs: calculate_poseidon_and_generate_intermediate_trace; circuits/src/generation/poseidon.rs: generate_poseidon_trace"

    # ---- the proving configuration and the FRI reduction plan (circuits/src/stark/config.rs:18 standard_fast_config, :32 fri_params;
    #      plonky2 fri/mod.rs:36 FriConfig::fri_params; fri/reduction_strategies.rs:30 reduction_arity_bits)
    cfgf = os.path.join(reference, "circuits", "src", "stark", "config.rs")
    cfg = it.call_assoc("StarkConfig", "standard_fast_config", [], cfgf)
    fc = cfg["fri_config"]
    strat = fc["reduction_strategy"]
    out["stark_config"] = {"security_bits": cfg["security_bits"], "num_challenges": cfg["num_challenges"], "rate_bits": fc["rate_bits"], "cap_height": fc["cap_height"],
                           "proof_of_work_bits": fc["proof_of_work_bits"], "num_query_rounds": fc["num_query_rounds"],
                           "reduction_strategy": [strat.variant] + list(strat.payload)}
    out["fri_params"] = []
    for db in range(0, 31):
        fp = it.call_assoc("StarkConfig", "fri_params", [cfg, db], cfgf)
        out["fri_params"].append({"degree_bits": db, "hiding": fp["hiding"], "reduction_arity_bits": list(fp["reduction_arity_bits"])})
    out["sources"]["stark_config"] = "circuits/src/stark/config.rs: StarkConfig::standard_fast_config"
    out["sources"]["fri_params"] = "circuits/src/stark/config.rs: StarkConfig::fri_params -> plonky2 fri/mod.rs: FriConfig::fri_params -> reduction_strategies.rs"
    return out


NTT_FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_ntt_vectors.json")
NTT_OPS = ("evaluate_poly", "interpolate_poly", "evaluate_poly_with_offset", "interpolate_poly_with_offset")


def ntt_interp(reference):
    """the reference's transforms (plonky2/field/src/cfft/{mod,serial,concurrent}.rs) as the real build runs them: feature "parallel" on, so
    mod.rs dispatches sizes >= MIN_CONCURRENT_SIZE = 1024 to concurrent.rs (the four-step split_radix_fft) and smaller ones to serial.rs"""
    it = plonky2_interp(reference)
    cf = os.path.join(reference, "plonky2", "field", "src", "cfft")
    it.cfft = os.path.join(cf, "mod.rs")
    it.extra_files += [it.cfft, os.path.join(cf, "serial.rs"), os.path.join(cf, "concurrent.rs"), os.path.join(reference, "plonky2", "util", "src", "lib.rs")]
    it.features = {"parallel"}
    gf = open(os.path.join(reference, "plonky2", "field", "src", "goldilocks_field.rs")).read()
    for name in ("TWO_ADICITY", "POWER_OF_TWO_GENERATOR", "MULTIPLICATIVE_GROUP_GENERATOR"):
        it.field_consts[name] = int(re.search(r"const %s: \w+ = (?:Self\()?(\d+)\)?;" % name, gf).group(1))
    return it


def ntt_input(log_n, op):
    return stream_for(4000 + log_n, NTT_OPS.index(op), 1 << log_n)


def ntt_run(it, op, log_n, threads=8):
    """one transform of the reference on ntt_input(log_n, op) -> output words.  `with_offset`: domain offset = F::coset_shift(), blowup 8 (what
    PolynomialBatch::lde_values asks for, fri/oracle.rs:120)"""
    n = 1 << log_n
    it.num_threads = threads
    data = [Fe(x) for x in ntt_input(log_n, op)]
    shift = it.call(("call", ("path", ["F", "coset_shift"], 0), [], 0), {}, X.Src.get(it.cfft))
    if op in ("evaluate_poly", "evaluate_poly_with_offset"):
        tw = it.call_free(it.cfft, "get_twiddles", [n])
    else:
        tw = it.call_free(it.cfft, "get_inv_twiddles", [n])
    if op == "evaluate_poly_with_offset":
        out = it.call_free(it.cfft, op, [data, tw, shift, 8])
    elif op == "interpolate_poly_with_offset":
        it.call_free(it.cfft, op, [data, tw, shift])
        out = data
    else:
        it.call_free(it.cfft, op, [data, tw])
        out = data
    return [x.v for x in out]


def ntt_record(words):
    import hashlib
    import struct
    return {"sha256": hashlib.sha256(struct.pack("<%dQ" % len(words), *words)).hexdigest(), "len": len(words), "head": words[:4], "tail": words[-2:]}


def ntt_vectors(reference, sizes=tuple(range(1, 17))):
    """outputs of the reference's four transforms, sizes 2^1 .. 2^16 (serial.rs below 2^10; concurrent.rs from 2^10 on: square splits at even,
    2:1 splits at odd exponents), as digests.  2^14 and up are the sizes the GPU runs on its T-form pass kernels.  About 40 minutes."""
    it = ntt_interp(reference)
    out = {"generated_by": "tools/rust_air_eval.py --ntt", "input": "ntt_input(log_n, op) = stream_for(4000 + log_n, index of op, 2^log_n)",
           "source": "plonky2/field/src/cfft/mod.rs :22 :65 :128 :180 -> serial.rs / concurrent.rs (feature parallel: n >= 1024)", "vectors": []}
    for log_n in sizes:
        for op in NTT_OPS:
            out["vectors"].append({"op": op, "log_n": log_n, "path": "concurrent.rs" if log_n >= 10 else "serial.rs", **ntt_record(ntt_run(it, op, log_n))})
    # the split into rayon batches must not show: the same outputs with another thread count
    for op in NTT_OPS:
        assert ntt_record(ntt_run(it, op, 10, threads=3)) == {k: v for k, v in [x for x in out["vectors"] if x["op"] == op and x["log_n"] == 10][0].items()
                                                                if k in ("sha256", "len", "head", "tail")}
    return out


TRACEGEN_FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_tracegen_vectors.json")
CMP_ROWS = [(5, 3), (3, 5), (7, 7), (0xFFFFFFFF, 0), (0x12345678, 0x9ABCDEF0)]
RC_ROWS = [(123456789, 1, 0, 0, 0), (65535, 0, 1, 0, 0), (4294967295, 0, 0, 1, 0), (65536, 0, 0, 0, 1), (0, 1, 0, 0, 0), (70000, 0, 0, 0, 1), (70000, 1, 0, 0, 0)]


def trace_digest(cols):
    import hashlib
    import struct
    h = hashlib.sha256()
    for c in cols:
        h.update(struct.pack("<%dQ" % len(c), *[x.v for x in c]))
    return {"columns": len(cols), "rows": len(cols[0]), "sha256": h.hexdigest()}


def tracegen_vectors(reference, heavy=True):
    """What the reference's trace generators (circuits/src/generation/*.rs) return -- column-major, as u64 words, digested: the comparison
    table with live rows, the range-check table (2^16 rows: the fixed column, the looked-up values' limbs and both `permuted_cols` pairs) with
    one row for each looking table, and the padding-only output of the generators that take no rows here (an execution that never touches
    the table)"""
    it = plonky2_interp(reference)
    g = os.path.join(reference, "circuits", "src", "generation")
    out = {"generated_by": "tools/rust_air_eval.py --tracegen", "sources": "circuits/src/generation/{builtin,cpu,memory,poseidon,tape,sccall}.rs"}
    cells = []
    for a, b in CMP_ROWS:
        d = abs(a - b)
        cells.append(Struct({"__name__": "CmpRow", "op0": Fe(a), "op1": Fe(b), "gte": Fe(int(a >= b)), "abs_diff": Fe(d),
                             "abs_diff_inv": Fe(pow(d, P - 2, P) if d else 0), "filter_looking_rc": Fe(1)}))
    cmp_ = it.call_free(os.path.join(g, "builtin.rs"), "generate_cmp_trace", [cells])
    out["cmp"] = {"rows_in": [list(r) for r in CMP_ROWS], **trace_digest(cmp_), "trace": [[x.v for x in c] for c in cmp_]}
    for name, file, fn in (("cpu", "cpu.rs", "generate_cpu_trace"), ("memory", "memory.rs", "generate_memory_trace"), ("tape", "tape.rs", "generate_tape_trace"),
                           ("sccall", "sccall.rs", "generate_sccall_trace"), ("poseidon", "poseidon.rs", "generate_poseidon_trace")):
        cols = it.call_free(os.path.join(g, file), fn, [[]])
        out[name + "_no_rows"] = {**trace_digest(cols), "trace": [[x.v for x in c] for c in cols]}
    if heavy:
        cells = [Struct({"__name__": "RangeCheckRow", "val": Fe(v), "limb_lo": Fe(v % 65536), "limb_hi": Fe(v // 65536), "filter_looked_for_cpu": Fe(a),
                         "filter_looked_for_mem_sort": Fe(b), "filter_looked_for_mem_region": Fe(c), "filter_looked_for_comparison": Fe(d),
                         "filter_looked_for_storage": Fe(0)}) for v, a, b, c, d in RC_ROWS]
        rc = it.call_free(os.path.join(g, "builtin.rs"), "generate_rc_trace", [cells])
        out["rangecheck"] = {"rows_in": [list(r) for r in RC_ROWS], **trace_digest(rc), "head": [[x.v for x in c[:8]] for c in rc]}
    return out


BITWISE_FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_tracegen_bitwise.json")
BITWISE_OPS = [("AND", 0xADBEEF, 0x345678), ("OR", 0xFFFFFF, 0x800001), ("XOR", 0x00FFFF, 0xA5A5A5)]


def tracegen_bitwise(reference, ops=BITWISE_OPS):
    """`generate_bitwise_trace` (generation/builtin.rs:35-205) interpreted: the 2^18-row table (fixed AND / OR / XOR table, the operations'
    rows, the compress challenge drawn from a transcript over the twelve limb columns, sixteen `permuted_cols` pairs) -- about 40 minutes.
    Operands below 2^24: the reference writes the fourth limb of op0 / op1 / res to `OP0_LIMBS.end` etc., i.e. one column too far, where the
    next write overwrites it (builtin.rs:66, 71, 76), so its limb-3 columns stay zero whatever the operand."""
    it = plonky2_interp(reference)
    pos = FastPoseidonHook(it)
    it.permutation_hook = pos
    fn = {"AND": lambda x, y: x & y, "OR": lambda x, y: x | y, "XOR": lambda x, y: x ^ y}
    src = X.Src.get(os.path.join(reference, "circuits", "src", "generation", "builtin.rs"))
    cells = []
    for name, x, y in ops:
        z = fn[name](x, y)
        d = {"__name__": "BitwiseCombinedRow", "opcode": it.method(Enum("OlaOpcode", name), "binary_bit_mask", [], src, 0), "op0": Fe(x), "op1": Fe(y),
             "res": Fe(z)}
        for i in range(4):
            d["op0_%d" % i], d["op1_%d" % i], d["res_%d" % i] = Fe((x >> (8 * i)) & 255), Fe((y >> (8 * i)) & 255), Fe((z >> (8 * i)) & 255)
        cells.append(Struct(d))
    tr, beta = it.call_free(src.path, "generate_bitwise_trace", [cells])
    return {"generated_by": "tools/rust_air_eval.py --tracegen-bitwise", "ops": [list(o) for o in ops], "beta": beta.v, **trace_digest(tr),
            "rows_head": [[c[i].v for c in tr] for i in range(len(ops))]}


class FastPoseidonHook:
    """the permutation for the 393 216 sponge calls of the bitwise generator's transcript: built from the reference's constant tables and checked
    against the interpreted poseidon_naive before use (as tools/ref_verifier.py does)"""

    def __init__(self, it):
        src = X.Src.get(os.path.join(it.plonky2, "hash", "poseidon.rs"))
        gsrc = X.Src.get(os.path.join(it.plonky2, "hash", "poseidon_goldilocks.rs"))
        circ = [int(x) for x in it.const_value("MDS_MATRIX_CIRC", gsrc)]
        diag = [int(x) for x in it.const_value("MDS_MATRIX_DIAG", gsrc)]
        self.rc = [int(x) for x in it.const_value("ALL_ROUND_CONSTANTS", src)]
        self.rows = [[circ[(j - r) % 12] + (diag[r] if j == r else 0) for j in range(12)] for r in range(12)]
        naive = it.permutation_hook
        for k in range(4):
            v = [Fe(x) for x in stream_for(8800 + k, 3, 12)]
            if [x.v for x in naive(list(v))] != [x.v for x in self(v)]:
                raise SystemExit("direct Poseidon permutation disagrees with the interpreted poseidon_naive")

    def __call__(self, state, segs=None):
        s = [x.v for x in state]
        r = 0
        for full, rounds in ((True, 4), (False, 22), (True, 4)):
            for _ in range(rounds):
                k = 12 * r
                if full:
                    s = [pow(s[i] + self.rc[k + i], 7, P) for i in range(12)]
                else:
                    s = [s[i] + self.rc[k + i] for i in range(12)]
                    s[0] = pow(s[0], 7, P)
                s = [sum(m[j] * s[j] for j in range(12)) % P for m in self.rows]
                r += 1
        return [Fe(x) for x in s]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--points", type=int, default=2)
    ap.add_argument("--out", default=FIXTURE)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--primitives", action="store_true", help="the hashing / transcript / FRI-parameter vectors instead of the AIR vectors")
    ap.add_argument("--ntt", action="store_true", help="the transform vectors (cfft) instead of the AIR vectors")
    ap.add_argument("--tracegen", action="store_true", help="the trace generators' outputs (generation/*.rs) instead of the AIR vectors")
    ap.add_argument("--tracegen-bitwise", action="store_true", help="the bitwise table's generator (2^18 rows: about 40 minutes)")
    a = ap.parse_args()
    sys.setrecursionlimit(20000)
    if a.tracegen_bitwise:
        data = tracegen_bitwise(a.reference)
        open(BITWISE_FIXTURE, "w").write(json.dumps(data, separators=(",", ":")) + "\n")
        print("wrote", BITWISE_FIXTURE)
        return
    if a.tracegen:
        out = TRACEGEN_FIXTURE if a.out == FIXTURE else a.out
        data = tracegen_vectors(a.reference)
        text = json.dumps(data, separators=(",", ":")) + "\n"
        if a.check:
            if open(out).read() != text:
                raise SystemExit(out + " is stale")
            print("fixture is up to date")
            return
        open(out, "w").write(text)
        print("wrote", out, "(%d bytes)" % len(text))
        return
    if a.ntt:
        out = NTT_FIXTURE if a.out == FIXTURE else a.out
        data = ntt_vectors(a.reference)
        text = json.dumps(data, indent=0) + "\n"
        print("%d transforms" % len(data["vectors"]))
        if a.check:
            if open(out).read() != text:
                raise SystemExit(out + " is stale")
            print("fixture is up to date")
            return
        open(out, "w").write(text)
        print("wrote", out, "(%d bytes)" % len(text))
        return
    if a.primitives:
        out = PRIMITIVES_FIXTURE if a.out == FIXTURE else a.out
        data = primitives(a.reference)
        text = json.dumps(data, separators=(",", ":")) + "\n"
        print("%d permutations, %d sponge hashes, %d compressions, %d transcripts, %d lookup cases, %d FRI plans" % (
            len(data["poseidon"]), len(data["hash_no_pad"]), len(data["two_to_one"]), len(data["challenger"]), len(data["permuted_cols"]),
            len(data["fri_params"])))
        if a.check:
            if open(out).read() != text:
                raise SystemExit(out + " is stale")
            print("fixture is up to date")
            return
        open(out, "w").write(text)
        print("wrote", out, "(%d bytes)" % len(text))
        return
    data = extract(a.reference, a.points)
    for t in data["tables"]:
        if "points" in t:
            print("%-14s %4d constraints evaluated at %d points" % (t["table"], len(t["points"][0]["values"]), len(t["points"])))
        else:
            print("%-14s NOT EVALUATED: %s" % (t["table"], t["unevaluated"]))
    bad = [c for c in data["cross_table_lookups"] if "unevaluated" in c]
    print("%d cross-table lookups: %d lookup entries evaluated%s" % (len(data["cross_table_lookups"]), sum(len(c.get("entries", ())) for c in data["cross_table_lookups"]),
                                                                      "".join("\n  NOT EVALUATED %s: %s" % (c["name"], c["unevaluated"]) for c in bad)))
    for pr in data["problems"]:
        print("PROBLEM", pr["table"], pr["reason"][:300])
    if data["vanishing_poly"]:
        print("eval_vanishing_poly: %d constraints over the 12 tables, %d input sets" % (sum(len(t["values"]) for t in data["vanishing_poly"][0]["tables"]), len(data["vanishing_poly"])))
    text = json.dumps(data, separators=(",", ":")) + "\n"
    if a.check:
        if open(a.out).read() != text:
            raise SystemExit(a.out + " is stale")
        print("fixture is up to date")
        return
    open(a.out, "w").write(text)
    print("wrote", a.out, "(%d bytes)" % len(text))


if __name__ == "__main__":
    main()
