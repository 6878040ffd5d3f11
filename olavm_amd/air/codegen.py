"""Ahead-of-time specialisation of the constraint-quotient kernel.

The reference compiles every table's `eval_packed_generic` into the prover binary (e.g. cpu/cpu_stark.rs:325-1000).  The
MI355X counterpart: for each table of an AirSet this module prints one straight-line HIP kernel (constraint program,
permutation checks and cross-table-lookup checks unrolled, column indices and constants as immediates), which
`olavm_amd/csrc/stark.hip` includes and selects at run time by the signature of the AIR-set blob it is handed
(`AirSet.signature`).  A blob without a matching kernel runs on the generic interpreter kernel instead -- same results,
fewer points per second.

The random-linear-combination of constraints is accumulated lazily: sum_i alpha^(K-1-i) * c_i is the value the reference's
Horner recurrence `acc = acc*alpha + c_i` (constraint_consumer.rs:57-64) produces, so each emitted constraint costs two
multiply-accumulates into lazy sums (one per challenge; round 6: three 64-bit sums of 22-bit-limb products, airq.cuh Acc3 --
160-bit sums of 128-bit products before) and the reductions mod p happen once per point.
Transition constraints share their `z_last` factor the same way.

Run as a script to (re)generate olavm_amd/csrc/gen/air_kernels.inc for the OlaStark table sets the tests and benches use.
"""
import os

from .dsl import (OP_LOCAL, OP_NEXT, OP_CONST, OP_PARAM, OP_ADD, OP_SUB, OP_MUL, OP_ISZERO,
                  KIND_ALL, KIND_TRANSITION, KIND_FIRST, KIND_LAST, P)

D_ZH = 0          # D[0..8): 1/Z_H per coset
D_W = 8           # D[8 .. 8+2K): alpha weights, challenge-major


def _lit(v):
    return "0x%Xull" % (int(v) % P)


LIMB_LOADS_AHEAD = int(os.environ.get("OLA_AIRQ_LIMB_LOADS_AHEAD", "2"))    # uses of limb slots between a slot's load and its own use
W_MARK, USE_MARK = "/*W*/", "/*U*/"     # lines that load limb slots / lines that use them (hoist_limb_loads)


def hoist_limb_loads(lines):
    """The limb forms of the multipliers stream through the scalar cache (40 KB per point for the CPU table: no cache holds them),
    and a scalar load issued where it is used is waited for on the spot.  Every AIRQ_W line is therefore moved up to just behind
    the PREVIOUS use of limb slots (or segment barrier) -- one use ahead, twelve SGPRs -- as far as C++ scope allows: the
    declaration must stay in a block that is still open where it is used."""
    def delta(ln):
        return ln.count("{") - ln.count("}")
    out = list(lines)
    i = 0
    while i < len(out):
        if not out[i].rstrip().endswith(W_MARK):
            i += 1
            continue
        # depth before each line, relative to the W line (0); walk upwards
        depth, lowest, best, uses = 0, 0, i, 0
        j = i - 1
        while j >= 0:
            ln = out[j]
            if "AIRQ_SEGMENT_BARRIER" in ln or "AIRQ_LIMBS_AT" in ln:
                break
            if ln.rstrip().endswith(USE_MARK):
                uses += 1
                if uses >= LIMB_LOADS_AHEAD:
                    break
            depth -= delta(ln)          # depth before line j
            if depth <= lowest:
                lowest = depth
                best = j                # inserting before line j keeps the declaration in an enclosing, still open block
            j -= 1
        if best != i:
            out.insert(best, out.pop(i))
        i += 1
    return out


ACC_REFOLD_EVERY = 480   # multiply-accumulates into one lazy sum before it is folded and restarted (airq.cuh Acc3: 512 at most)


class _Emitter:
    def __init__(self, K):
        self.lines, self.K, self.idx = [], K, 0
        self.n_all = self.n_trans = 0
        self.limb_src = []      # per limb slot (three descriptor words each, in the order the kernel uses them): the u64 word it is the limb form of

    def slot(self, *words):
        """consecutive limb slots for the given u64 descriptor words -> index of the first"""
        s0 = len(self.limb_src)
        self.limb_src.extend(words)
        return s0

    def _count(self, trans):
        if trans:
            self.n_trans += 1
            if self.n_trans % ACC_REFOLD_EVERY == 0:
                self.add("AIRQ_REFOLD_TRANS;")
        else:
            self.n_all += 1
            if self.n_all % ACC_REFOLD_EVERY == 0:
                self.add("AIRQ_REFOLD_ALL;")

    def add(self, s):
        self.lines.append("    " + s)

    def emit(self, kind, expr, index=None):
        """index: the constraint's position in the reference's emission order (its alpha power); evaluation order is free"""
        if index is None:
            i = self.idx
            self.idx += 1
        else:
            i = index
        sl = self.slot(D_W + i, D_W + self.K + i)     # the two challenges' weights, side by side in the limb area
        self.add("AIRQ_W(%d); AIRQ_W(%d); %s" % (sl, sl + 1, W_MARK))
        if kind == KIND_ALL:
            self.add("AIRQ_EMIT_ALL(%d, %d, %d, %s); %s" % (i, sl, sl + 1, expr, USE_MARK))
        elif kind == KIND_TRANSITION:
            self.add("AIRQ_EMIT_TRANS(%d, %d, %d, %s); %s" % (i, sl, sl + 1, expr, USE_MARK))
        elif kind == KIND_FIRST:
            self.add("AIRQ_EMIT_ALL(%d, %d, %d, gl_mul(%s, lag_first)); %s" % (i, sl, sl + 1, expr, USE_MARK))
        else:
            self.add("AIRQ_EMIT_ALL(%d, %d, %d, gl_mul(%s, lag_last)); %s" % (i, sl, sl + 1, expr, USE_MARK))
        self._count(kind == KIND_TRANSITION)


def _lincol(col, row, cell=None):
    """C expression of a Col (linear combination of trace columns) on the local ('L') or next ('N') row."""
    acc = None
    for c, f in col.terms:
        v = cell(row, c) if cell else "%sC(%d)" % (row, c)
        if f == 1:
            term, neg = v, False
        elif f == P - 1:
            term, neg = v, True
        else:
            term, neg = "gl_mul(%s, %s)" % (v, _lit(f)), False
        if acc is None:
            acc = ("gl_neg(%s)" % term) if neg else term
        else:
            acc = ("gl_sub(%s, %s)" if neg else "gl_add(%s, %s)") % (acc, term)
    if acc is None:
        return _lit(col.constant)
    if col.constant:
        acc = "gl_add(%s, %s)" % (acc, _lit(col.constant))
    return acc


def num_emits(airset, t, num_challenges=2):
    tab = airset.tables[t]
    return len(tab.emits) + 2 * tab.num_permutation_batches(num_challenges) + 2 * len(airset.ctl_jobs(t, num_challenges))


SEGMENT_OPS = int(os.environ.get("OLA_AIRQ_SEGMENT_OPS", "64"))     # field operations between two code-motion barriers
# Re-loads of trace cells are what the large tables' kernels wait for (the CPU table: 1 247 cell loads in the source for 331
# distinct cells, none of them answered by a cache -- profiles/r04_proof_pmc_*.txt).  Two places a cell can stay instead:
CTL_REG_CELLS = int(os.environ.get("OLA_AIRQ_CTL_REG_CELLS", "12"))   # cells most lookups of the table read (clk, the selectors, ...): in registers for the whole lookup section
CTL_REG_MIN_USES = 6
LDS_SLOTS = int(os.environ.get("OLA_AIRQ_LDS_SLOTS", "20"))       # lane-private LDS slots per thread (20 x 2 KB per workgroup: four workgroups per CU still fit in 160 KB)
MIN_WAVES = int(os.environ.get("OLA_AIRQ_MIN_WAVES", "4"))        # waves per SIMD the kernels are compiled for (__launch_bounds__: 4 -> 128 VGPRs, 3 -> 168)
ORDER_WINDOWS = (0, 2, 3, 5, 8)   # candidate evaluation orders of the constraint program: 0 = the reference's, W = greedy with a window of W constraints


def _emit_cells(tab):
    """per emitted constraint: the trace cells under its expression"""
    nodes, out = tab.nodes, []
    memo = {}

    def cells(j):
        if j in memo:
            return memo[j]
        op, a, b = nodes[j]
        if op == OP_LOCAL:
            r = frozenset([("L", a)])
        elif op == OP_NEXT:
            r = frozenset([("N", a)])
        elif op in (OP_CONST, OP_PARAM):
            r = frozenset()
        elif op == OP_ISZERO:
            r = cells(a)
        else:
            r = cells(a) | cells(b)
        memo[j] = r
        return r

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 20000))
    try:
        for _, j in tab.emits:
            out.append(cells(j))
    finally:
        sys.setrecursionlimit(old)
    return out


def emit_order(tab, window):
    """Evaluation order of the table's constraints (emission indices -- the alpha powers -- stay attached to them): window 0 is the
    reference's order; otherwise greedy -- the next constraint is the one that shares the most trace cells with the last `window`
    constraints and brings the fewest new ones, so that the cells a stretch of the kernel touches fit the LDS slots."""
    n = len(tab.emits)
    if window == 0 or n < 3:
        return list(range(n))
    full = _emit_cells(tab)
    left = set(range(1, n))
    order, recent = [0], [full[0]]
    while left:
        win = frozenset().union(*recent[-window:])
        best = max(left, key=lambda i: (2 * len(full[i] & win) - len(full[i] - win), -i))
        order.append(best)
        left.discard(best)
        recent.append(full[best])
    return order


def _schedule(tab, order):
    """dsl.AirTable.schedule with the emits visited in `order`: ('node', id) / ('emit', kind, id, emission index)"""
    nodes, emits = tab.nodes, tab.emits
    out, done = [], set()
    for i in order:
        kind, root = emits[i]
        stack = [root]
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
            out.append(("node", j))
            stack.pop()
        out.append(("emit", kind, root, i))
    return out


def plan_lds(segcells, capacity):
    """Which trace cells live in the lane-private LDS slots while which segment runs.  The kernel is straight-line code, so every
    future use is known: replacement is Belady's (keep what is used again soonest), decided segment by segment.  A cell enters a
    slot in the segment that loads it from global memory (AIRQ_CACHE_PUT right after the load), into a slot that is free when that
    segment starts -- a slot whose cell is still read in the segment is handed over one segment later.
    -> per segment: {cell: slot} served from LDS, {cell: slot} loaded and parked; and the number of global loads."""
    nseg = len(segcells)
    INF = 1 << 30
    uses = {}
    for i, sc in enumerate(segcells):
        for c in sc:
            uses.setdefault(c, []).append(i)

    def next_use(c, i):
        for k in uses[c]:
            if k > i:
                return k
        return INF

    cache = {}            # cell -> slot
    hits, puts, loads = [], [], 0
    for i, sc in enumerate(segcells):
        hit = {c: cache[c] for c in sc if c in cache}
        miss = [c for c in sc if c not in cache]
        loads += len(miss)
        cand = sorted(set(cache) | set(miss), key=lambda c: (next_use(c, i), c))
        keep = set(c for c in cand[:capacity] if next_use(c, i) < INF)
        for c in [c for c in cache if c not in keep and c not in sc]:
            del cache[c]                       # not read in this segment and not worth keeping: the slot is free right away
        free = [sl for sl in range(capacity) if sl not in cache.values()]
        put = {}
        for c in sorted((c for c in miss if c in keep), key=lambda c: (next_use(c, i), c)):
            if not free:
                break
            put[c] = free.pop(0)
        hits.append(hit)
        puts.append(put)
        for c in [c for c in cache if c not in keep]:
            del cache[c]                       # read in this segment for the last time: the slot frees for the next segment
        cache.update(put)
    return hits, puts, loads


def check_plan(plan, seen):
    """The slot plan was made from a COUNTING print; the final print must cut the program into the same segments and touch the
    same cells in each (a CL(slot) read of a differently segmented program would return another cell's value, and only the GPU
    byte comparisons would notice).  Also replays the slots: a cell served from a slot was parked there by an earlier segment and
    nothing has been parked over it since; a segment never parks into a slot it still reads."""
    hits, puts, segs = plan
    assert len(seen) == len(segs) == len(hits) == len(puts), (len(seen), len(segs), len(hits), len(puts))
    for i, (a, b) in enumerate(zip(seen, segs)):
        assert a == b, ("segment %d reads other cells than the plan assumed" % i, sorted(a ^ b))
    slot = {}
    for i, (hit, put) in enumerate(zip(hits, puts)):
        for c, sl in hit.items():
            assert slot.get(sl) == c, ("segment %d: slot %d holds %r, not %r" % (i, sl, slot.get(sl), c))
            assert c in segs[i]
        assert not set(hit.values()) & set(put.values()), ("segment %d parks into a slot it reads" % i)
        assert len(set(put.values())) == len(put)
        for c, sl in put.items():
            assert c in segs[i] and c not in hit
            slot[sl] = c


def table_kernel(airset, t, name, num_challenges=2, with_limbs=False):
    """-> (source text of the kernel, K).  Printed several times: a counting print per candidate evaluation order finds the cells
    the lookup section keeps in registers and the cells every segment touches; plan_lds decides from that which cells sit in the
    LDS slots when; the order with the fewest loads from global memory is printed for good."""
    import re
    tab = airset.tables[t]
    best = None
    for window in ORDER_WINDOWS:
        order = emit_order(tab, window)
        for ctl_order in (False, True):
            uses = {"ctl": {}, "segs": []}
            _table_kernel(airset, t, name, num_challenges, uses=uses, order=order, ctl_order=ctl_order)
            ctl = sorted(((n, k) for k, n in uses["ctl"].items() if n >= CTL_REG_MIN_USES), key=lambda x: (-x[0], x[1]))[:CTL_REG_CELLS]
            regs = [k for _, k in ctl]
            uses = {"ctl": {}, "segs": []}
            _table_kernel(airset, t, name, num_challenges, uses=uses, ctl_regs=regs, order=order, ctl_order=ctl_order)
            hits, puts, loads = plan_lds(uses["segs"], LDS_SLOTS)
            if best is None or loads < best[0]:
                best = (loads, order, regs, (hits, puts, [set(x) for x in uses["segs"]]), ctl_order)
        if len(tab.emits) < 3:
            break
    loads, order, regs, plan, ctl_order = best
    src, K, nl = _table_kernel(airset, t, name, num_challenges, ctl_regs=regs, plan=plan, order=order, ctl_order=ctl_order)
    return (src, K, nl) if with_limbs else (src, K)


def _table_kernel(airset, t, name, num_challenges=2, uses=None, ctl_regs=(), plan=None, order=None, ctl_order=False):
    """One print of the kernel.  uses: dict to fill with the cells every segment reads (counting print); plan: plan_lds's result;
    order: evaluation order of the constraint program (emit_order).

    Register pressure is what limits these kernels (the CPU table keeps ~90 trace cells and ~65 shared subexpressions
    alive if every value is computed once), so the code is cut into segments separated by compiler barriers: trace
    cells, constants and one-operation combinations of them ("cheap" nodes) are re-loaded / recomputed in every segment
    that uses them instead of being kept in registers; only multi-operation subexpressions stay live across segments."""
    tab = airset.tables[t]
    jobs = airset.ctl_jobs(t, num_challenges)
    nperm = tab.num_permutation_batches(num_challenges)
    bs = tab.quotient_degree_factor
    K = num_emits(airset, t, num_challenges)
    d_params = D_W + 2 * K
    d_perm = d_params + tab.n_params
    d_ctl = d_perm + 2 * nperm * bs
    e = _Emitter(K)
    e.add("AIRQ_PROLOGUE(%d)" % K)
    nodes = tab.nodes
    LEAF = (OP_LOCAL, OP_NEXT, OP_CONST, OP_PARAM)
    use_lds = plan is not None and any(plan[0]) or plan is not None and any(plan[1])
    section = {"name": "program"}
    BARRIER = "AIRQ_SEGMENT_BARRIER_C;" if use_lds else "AIRQ_SEGMENT_BARRIER;"
    state = {"seg": 0, "ops": 0, "local": {}, "cells": {}}      # seg: running index over ALL segments (program, permutation, lookups)
    if uses is not None:
        uses["segs"].append(set())
    seen = [set()]        # the cells this print touches per segment (checked against the plan's segments at the end)

    def cell(row, c):
        """Expression of trace cell (row 'L' / 'N', column c) at this place of the kernel."""
        k = (row, c)
        if section["name"] == "ctl" and k in ctl_regs:
            if uses is not None:
                uses["ctl"][k] = uses["ctl"].get(k, 0) + 1
            return "h%s%d" % (row, c)
        if uses is not None:
            uses["segs"][state["seg"]].add(k)
            if section["name"] == "ctl":
                uses["ctl"][k] = uses["ctl"].get(k, 0) + 1
        seen[state["seg"]].add(k)
        if plan is not None:
            hits, puts = plan[0], plan[1]
            if k in hits[state["seg"]]:
                return "CL(%d)" % hits[state["seg"]][k]
            if k in puts[state["seg"]]:
                if k not in state["cells"]:
                    nm = "g%d_%s%d" % (state["seg"], row, c)
                    e.add("const u64 %s = %sC(%d); AIRQ_CACHE_PUT(%d, %s);" % (nm, row, c, puts[state["seg"]][k], nm))
                    state["cells"][k] = nm
                return state["cells"][k]
        return "%sC(%d)" % (row, c)

    def next_segment():
        e.add(BARRIER)
        state["seg"] += 1
        state["ops"] = 0
        state["local"] = {}
        state["cells"] = {}
        seen.append(set())
        if uses is not None:
            uses["segs"].append(set())

    if use_lds:
        e.add("AIRQ_CACHE_DECL(%d);" % LDS_SLOTS)

    def is_cheap(j):
        op, a, b = nodes[j]
        if op in LEAF:
            return True
        return op in (OP_ADD, OP_SUB) and nodes[a][0] in LEAF and nodes[b][0] in LEAF

    barrier = next_segment

    def leaf_expr(j):
        op, a, _ = nodes[j]
        if op == OP_LOCAL:
            return cell("L", a)
        if op == OP_NEXT:
            return cell("N", a)
        if op == OP_CONST:
            return _lit(a)
        return "D[%d]" % (d_params + a)

    def ref(j):
        """Name of node j's value, materialising cheap nodes in the current segment."""
        if not is_cheap(j):
            return "t%d" % j
        op, a, b = nodes[j]
        if op == OP_CONST:
            return _lit(a)
        if j in state["local"]:
            return state["local"][j]
        nm = "s%d_%d" % (state["seg"], j)
        if op in LEAF:
            e.add("const u64 %s = %s;" % (nm, leaf_expr(j)))
        else:
            e.add("const u64 %s = %s(%s, %s);" % (nm, "gl_add" if op == OP_ADD else "gl_sub", ref(a), ref(b)))
            state["ops"] += 1
        state["local"][j] = nm
        return nm

    # ---- the table's constraint program ----
    for it in _schedule(tab, order if order is not None else list(range(len(tab.emits)))):
        if it[0] == "emit":
            e.emit(it[1], ref(it[2]), it[3])
            state["ops"] += 2
            if state["ops"] >= SEGMENT_OPS:
                barrier()
            continue
        j = it[1]
        if is_cheap(j):
            continue            # materialised where it is used
        op, a, b = nodes[j]
        if op == OP_ADD:
            rhs = "gl_add(%s, %s)" % (ref(a), ref(b))
        elif op == OP_SUB:
            rhs = "gl_sub(%s, %s)" % (ref(a), ref(b))
        elif op == OP_MUL:
            rhs = "gl_mul(%s, %s)" % (ref(a), ref(b))
        elif op == OP_ISZERO:
            rhs = "(%s == 0 ? 1ull : 0ull)" % ref(a)
        else:
            raise ValueError(op)
        e.add("const u64 t%d = %s;" % (j, rhs))
        state["ops"] += 1
    barrier()
    e.idx = len(tab.emits)
    # ---- permutation checks (permutation.rs:302-360) ----
    section["name"] = "perm"
    for b in range(nperm):
        e.emit(KIND_FIRST, "gl_sub(ZL(%d), 1)" % b)
    total = len(tab.permutation_pairs) * num_challenges
    inst = 0
    for b in range(nperm):
        e.add("{")
        e.add("    u64 pl = 1, pr = 1;")
        for i in range(bs):
            if inst >= total:
                break
            pair = tab.permutation_pairs[inst // num_challenges]
            slot = d_perm + 2 * (b * bs + i)
            ls = [cell("L", l) for l, _ in pair]          # (a cell that is parked in LDS here is declared before the block opens)
            rs = [cell("L", r) for _, r in pair]
            e.add("    { const u64 beta = D[%d], gamma = D[%d];" % (slot, slot + 1))
            e.add("      u64 l = %s, r = %s;" % (ls[-1], rs[-1]))
            for k in range(len(pair) - 2, -1, -1):
                e.add("      l = gl_add(gl_mul(l, beta), %s); r = gl_add(gl_mul(r, beta), %s);" % (ls[k], rs[k]))
            e.add("      pl = gl_mul(pl, gl_add(l, gamma)); pr = gl_mul(pr, gl_add(r, gamma)); }")
            inst += 1
        e.emit(KIND_ALL, "gl_sub(gl_mul(ZN(%d), pr), gl_mul(ZL(%d), pl))" % (b, b))
        e.add("}")
        next_segment()
    # ---- cross-table lookup checks (cross_table_lookup.rs:380-421) ----
    # Emits are indexed, so evaluation order is free: the Z columns that look at the same columns (one per challenge)
    # are evaluated together, streaming the column values through one Horner step per challenge.
    first_idx = e.idx
    groups = {}
    for i, twc in enumerate(jobs):
        groups.setdefault(id(twc), (twc, []))[1].append(i)
    # descriptor slice of CTL job i: [gamma_i, beta_i^0 .. beta_i^(ncol-1)]; the combination sum_k beta^k e_k
    # (= reduce_with_powers, plonk_common.rs:116-128) is accumulated un-reduced like the constraint sum
    d_job, off = {}, d_ctl
    for i, twc in enumerate(jobs):
        d_job[i] = off
        off += 1 + len(twc.columns)
    e.lines.insert(1, "    AIRQ_LIMBS_AT(%d);" % off)      # the limb area follows the u64 words
    section["name"] = "ctl_head"
    if ctl_regs:
        e.add("const u64 " + ", ".join("h%s%d = %s" % (r, c, cell(r, c)) for r, c in ctl_regs) + ";")
    section["name"] = "ctl"

    def twc_cells(twc):
        cols = list(twc.columns) + ([twc.filter_column] if twc.filter_column is not None else [])
        return frozenset((r, c) for col in cols for c, _ in col.terms for r in "LN" if (r, c) not in ctl_regs)

    # the lookups in an order in which neighbours read the same columns (their Z columns keep their emission indices)
    glist = list(groups.values())
    if ctl_order and len(glist) > 2:
        left, chain = glist[1:], [glist[0]]
        while left:
            last = twc_cells(chain[-1][0])
            nxt = max(left, key=lambda g: (2 * len(twc_cells(g[0]) & last) - len(twc_cells(g[0]) - last), -g[1][0]))
            chain.append(nxt)
            left.remove(nxt)
        glist = chain
    for twc, idxs in glist:
        e.add("{")
        e.add("  const u64 el0 = %s, en0 = %s;" % (_lincol(twc.columns[0], "L", cell), _lincol(twc.columns[0], "N", cell)))   # beta^0 term
        e.add("  AIRQ_ACC " + ", ".join("al%d = AIRQ_ACC_INIT(el0), an%d = AIRQ_ACC_INIT(en0)" % (i, i) for i in idxs) + ";")
        for k, col in enumerate(twc.columns):
            if k == 0:
                continue
            e.add("  { const u64 el = %s, en = %s;" % (_lincol(col, "L", cell), _lincol(col, "N", cell)))
            sl = e.slot(*[d_job[i] + 1 + k for i in idxs])
            e.lines.insert(len(e.lines) - 1, "      " + " ".join("AIRQ_W(%d);" % (sl + n_) for n_ in range(len(idxs))) + " " + W_MARK)    # before the block of this column
            for n_, i in enumerate(idxs):
                e.add("    AIRQ_MAD(al%d, el, %d, %d); AIRQ_MAD(an%d, en, %d, %d);%s" % (i, d_job[i] + 1 + k, sl + n_, i, d_job[i] + 1 + k, sl + n_,
                                                                                         " " + USE_MARK if n_ == len(idxs) - 1 else ""))
            e.add("  }")
        if twc.filter_column is not None:
            e.add("  const u64 fl = %s, fn = %s;" % (_lincol(twc.filter_column, "L", cell), _lincol(twc.filter_column, "N", cell)))
        for i in idxs:
            e.add("  { u64 cl = gl_add(AIRQ_ACC_REDUCE(al%d), D[%d]), cn = gl_add(AIRQ_ACC_REDUCE(an%d), D[%d]);" % (i, d_job[i], i, d_job[i]))
            if twc.filter_column is not None:   # select(f, x) = f*x + 1 - f
                e.add("    cl = gl_sub(gl_add(gl_mul(fl, cl), 1), fl); cn = gl_sub(gl_add(gl_mul(fn, cn), 1), fn);")
            e.add("    const u64 zl = ZL(%d), zn = ZN(%d);" % (nperm + i, nperm + i))
            e.idx = first_idx + 2 * i
            e.emit(KIND_FIRST, "gl_sub(zl, cl)")
            e.emit(KIND_TRANSITION, "gl_sub(zn, gl_mul(zl, cn))")
            e.add("  }")
        e.add("}")
        next_segment()
    e.idx = first_idx + 2 * len(jobs)
    assert e.idx == K, (e.idx, K)
    if plan is not None:
        check_plan(plan, seen)
    e.add("AIRQ_EPILOGUE")
    head = "// table %d (%s): %d columns, %d constraints, %d permutation Zs, %d CTL Zs, K = %d\n" % (
        t, tab.name, tab.ncols, len(tab.emits), nperm, len(jobs), K)
    src = head + "__global__ __launch_bounds__(AIRQ_THREADS, %d) void %s(QuotParams P) {\n%s\n}\n" % (MIN_WAVES, name, "\n".join(hoist_limb_loads(e.lines)))
    src += "// limb slot -> the descriptor word it is the limb form of (the host appends the limb area in this order: the kernel reads it front to back)\n"
    src += "static const int %s_limb_src[%d] = {%s};\n" % (name, max(1, len(e.limb_src)), ", ".join(str(x) for x in e.limb_src) or "0")
    return src, K, len(e.limb_src)


def generate(airsets):
    """-> {file name: source}: one translation unit per distinct table signature of the given AirSets (compiled in
    parallel by build()) plus air_registry.inc, the list stark.hip searches."""
    files, entries = {}, []
    for s in airsets:
        for t in range(len(s.tables)):
            sig = s.signature(t)
            name = "airq_%016x" % sig
            if name + ".hip" in files:
                continue
            src, K, nl = table_kernel(s, t, name, with_limbs=True)
            tab = s.tables[t]
            files[name + ".hip"] = (
                "// GENERATED by olavm_amd/air/codegen.py -- do not edit; regenerated by __graft_entry__.build().\n"
                "#define AIRQ_GENERATED_TU 1\n#include \"../airq.cuh\"\nnamespace ola {\n" + src +
                "extern const AirKernelEntry %s_entry;\nconst AirKernelEntry %s_entry = {0x%016Xull, %s, %d, %d, %d, \"%s\", %s_limb_src, %d};\n}  // namespace ola\n"
                % (name, name, sig, name, K, tab.n_params, tab.num_permutation_batches(), tab.name, name, nl))
            entries.append(name)
    reg = "// GENERATED by olavm_amd/air/codegen.py -- do not edit.\n"
    reg += "".join("extern const AirKernelEntry %s_entry;\n" % n for n in entries)
    reg += "static const AirKernelEntry* const AIR_KERNELS[] = {\n%s\n};\n" % "\n".join("    &%s_entry," % n for n in entries)
    files["air_registry.inc"] = reg
    return files


def default_airsets():
    """The table sets with ahead-of-time kernels: the reference's OlaStark and the miniature variants the tests use."""
    from . import ola_tables as T
    return [T.ola_stark(), T.ola_stark(range_bits=8, limb_bits=8), T.ola_stark(range_bits=4, limb_bits=2)]


def write_default(gen_dir):
    """Writes the generated sources into gen_dir (only files whose content changed are touched, stale ones are removed);
    returns the sorted list of .hip translation units."""
    files = generate(default_airsets())
    os.makedirs(gen_dir, exist_ok=True)
    for f in os.listdir(gen_dir):
        if f not in files and not f.endswith(".o"):
            os.remove(os.path.join(gen_dir, f))
    for f, src in files.items():
        path = os.path.join(gen_dir, f)
        if not os.path.exists(path) or open(path).read() != src:
            with open(path, "w") as fh:
                fh.write(src)
    return sorted(os.path.join(gen_dir, f) for f in files if f.endswith(".hip"))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for f in write_default(os.path.join(here, "..", "csrc", "gen")):
        print(f)
