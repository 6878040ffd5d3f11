"""Ahead-of-time specialisation of the constraint-quotient kernel.

The reference compiles every table's `eval_packed_generic` into the prover binary (e.g. cpu/cpu_stark.rs:325-1000).  The
MI355X counterpart: for each table of an AirSet this module prints one straight-line HIP kernel (constraint program,
permutation checks and cross-table-lookup checks unrolled, column indices and constants as immediates), which
`olavm_amd/csrc/stark.hip` includes and selects at run time by the signature of the AIR-set blob it is handed
(`AirSet.signature`).  A blob without a matching kernel runs on the generic interpreter kernel instead -- same results,
fewer points per second.

The random-linear-combination of constraints is accumulated lazily: sum_i alpha^(K-1-i) * c_i is the value the reference's
Horner recurrence `acc = acc*alpha + c_i` (constraint_consumer.rs:57-64) produces, so each emitted constraint costs two
64x64->128 multiply-accumulates into 160-bit sums (one per challenge) and the reductions mod p happen once per point.
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


class _Emitter:
    def __init__(self, K):
        self.lines, self.K, self.idx = [], K, 0

    def add(self, s):
        self.lines.append("    " + s)

    def emit(self, kind, expr):
        i = self.idx
        self.idx += 1
        if kind == KIND_ALL:
            self.add("AIRQ_EMIT_ALL(%d, %s);" % (i, expr))
        elif kind == KIND_TRANSITION:
            self.add("AIRQ_EMIT_TRANS(%d, %s);" % (i, expr))
        elif kind == KIND_FIRST:
            self.add("AIRQ_EMIT_ALL(%d, gl_mul(%s, lag_first));" % (i, expr))
        else:
            self.add("AIRQ_EMIT_ALL(%d, gl_mul(%s, lag_last));" % (i, expr))


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


SEGMENT_OPS = 64     # field operations between two code-motion barriers
# Re-loads of trace cells are what the large tables' kernels wait for (the CPU table: 1 247 cell loads in the source for 331
# distinct cells, none of them answered by a cache -- profiles/r04_proof_pmc_*.txt).  Two places a cell can stay instead:
CTL_REG_CELLS = 12   # cells most lookups of the table read (clk, the selectors, ...): in registers for the whole lookup section
CTL_REG_MIN_USES = 6
LDS_SLOTS = 20       # lane-private LDS slots per thread (20 x 2 KB per workgroup: four workgroups per CU still fit in 160 KB)
LDS_MIN_USES = 2


def table_kernel(airset, t, name, num_challenges=2):
    """-> (source text of the kernel, K).  Printed twice per candidate phase length: the first print counts how often each
    trace cell is loaded in each phase of the kernel, the second keeps the most re-loaded ones in registers (lookup section)
    or in LDS (AIRQ_CACHE_*; the slots are re-assigned at every phase boundary).  The phase length with the fewest loads from
    global memory wins (small tables: one phase for the whole constraint program)."""
    import re
    best = None
    for phase_segments in PHASE_SEGMENT_CHOICES:
        uses = {"phase": {}, "ctl": {}}
        _table_kernel(airset, t, name, num_challenges, uses=uses, phase_segments=phase_segments)
        ctl = sorted(((n, k) for k, n in uses["ctl"].items() if n >= CTL_REG_MIN_USES), key=lambda x: (-x[0], x[1]))[:CTL_REG_CELLS]
        regs = [k for _, k in ctl]
        lds_sets = {}
        for phase, cnt in uses["phase"].items():
            cnt = dict(cnt)
            if phase == "tail":        # a register cell is loaded once, at the head of the lookup section: that load may come from LDS
                for k in regs:
                    cnt[k] = cnt.get(k, 0) - uses["ctl"][k] + 1
            top = sorted(((n, k) for k, n in cnt.items() if n >= LDS_MIN_USES), key=lambda x: (-x[0], x[1]))[:LDS_SLOTS]
            lds_sets[phase] = [k for _, k in top]
        src, K = _table_kernel(airset, t, name, num_challenges, ctl_regs=regs, lds_sets=lds_sets, phase_segments=phase_segments)
        loads = len(re.findall(r"\b[LN]C\(\d+\)", src))
        if best is None or loads < best[0]:
            best = (loads, src, K)
    return best[1], best[2]


PHASE_SEGMENT_CHOICES = (7, 1 << 30)   # the constraint program is cut into phases of this many segments; each phase (and the
                                       # permutation + lookup tail) parks ITS most re-loaded cells in the LDS slots


def _table_kernel(airset, t, name, num_challenges=2, uses=None, ctl_regs=(), lds_sets=None, phase_segments=7):
    """One print of the kernel.  uses: dict to fill with the loads per phase and cell (counting print); lds_sets: {phase: cells}.

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
    lds_sets = lds_sets or {}
    use_lds = any(lds_sets.values())
    lds_slot = {}                        # cell -> slot, for the phase being printed
    section = {"name": "program", "phase": None}
    BARRIER = "AIRQ_SEGMENT_BARRIER_C;" if use_lds else "AIRQ_SEGMENT_BARRIER;"

    def cell(row, c):
        """Expression of trace cell (row 'L' / 'N', column c) at this place of the kernel."""
        k = (row, c)
        if uses is not None:
            ph = uses["phase"].setdefault(section["phase"], {})
            ph[k] = ph.get(k, 0) + 1
            if section["name"] == "ctl":
                uses["ctl"][k] = uses["ctl"].get(k, 0) + 1
        if section["name"] == "ctl" and k in ctl_regs:
            return "h%s%d" % (row, c)
        if k in lds_slot:
            return "CL(%d)" % lds_slot[k]
        return "%sC(%d)" % (row, c)

    def phase_begin(name):
        """The LDS slots change hands: cells that stay keep their slot, the others are loaded into the freed ones."""
        section["phase"] = name
        want = lds_sets.get(name, [])
        for k in [k for k in lds_slot if k not in want]:
            del lds_slot[k]
        free = [i for i in range(LDS_SLOTS) if i not in lds_slot.values()]
        fresh = [k for k in want if k not in lds_slot]
        for k in fresh:
            lds_slot[k] = free.pop(0)
            e.add("AIRQ_CACHE_PUT(%d, %sC(%d));" % (lds_slot[k], k[0], k[1]))
        if fresh:
            e.add(BARRIER)

    if use_lds:
        e.add("AIRQ_CACHE_DECL(%d);" % LDS_SLOTS)
    phase_begin("p0")

    def is_cheap(j):
        op, a, b = nodes[j]
        if op in LEAF:
            return True
        return op in (OP_ADD, OP_SUB) and nodes[a][0] in LEAF and nodes[b][0] in LEAF

    state = {"seg": 0, "ops": 0, "local": {}}

    def barrier():
        e.add(BARRIER)
        state["seg"] += 1
        state["ops"] = 0
        state["local"] = {}
        if section["name"] == "program" and state["seg"] % phase_segments == 0:
            phase_begin("p%d" % (state["seg"] // phase_segments))

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
    for it in tab.schedule():
        if it[0] == "emit":
            e.emit(it[1], ref(it[2]))
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
    # ---- permutation checks (permutation.rs:302-360) ----
    section["name"] = "perm"
    phase_begin("tail")
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
            e.add("    { const u64 beta = D[%d], gamma = D[%d];" % (slot, slot + 1))
            ls = [cell("L", l) for l, _ in pair]
            rs = [cell("L", r) for _, r in pair]
            e.add("      u64 l = %s, r = %s;" % (ls[-1], rs[-1]))
            for k in range(len(pair) - 2, -1, -1):
                e.add("      l = gl_add(gl_mul(l, beta), %s); r = gl_add(gl_mul(r, beta), %s);" % (ls[k], rs[k]))
            e.add("      pl = gl_mul(pl, gl_add(l, gamma)); pr = gl_mul(pr, gl_add(r, gamma)); }")
            inst += 1
        e.emit(KIND_ALL, "gl_sub(gl_mul(ZN(%d), pr), gl_mul(ZL(%d), pl))" % (b, b))
        e.add("}")
        e.add(BARRIER)
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
    section["name"] = "ctl_head"
    if ctl_regs:
        e.add("const u64 " + ", ".join("h%s%d = %s" % (r, c, cell(r, c)) for r, c in ctl_regs) + ";")
    section["name"] = "ctl"
    for twc, idxs in groups.values():
        e.add("{")
        e.add("  const u64 el0 = %s, en0 = %s;" % (_lincol(twc.columns[0], "L", cell), _lincol(twc.columns[0], "N", cell)))   # beta^0 term
        e.add("  Acc160 " + ", ".join("al%d = {el0, 0, 0}, an%d = {en0, 0, 0}" % (i, i) for i in idxs) + ";")
        for k, col in enumerate(twc.columns):
            if k == 0:
                continue
            e.add("  { const u64 el = %s, en = %s;" % (_lincol(col, "L", cell), _lincol(col, "N", cell)))
            for i in idxs:
                e.add("    acc_mad(al%d, el, D[%d]); acc_mad(an%d, en, D[%d]);" % (i, d_job[i] + 1 + k, i, d_job[i] + 1 + k))
            e.add("  }")
        if twc.filter_column is not None:
            e.add("  const u64 fl = %s, fn = %s;" % (_lincol(twc.filter_column, "L", cell), _lincol(twc.filter_column, "N", cell)))
        for i in idxs:
            e.add("  { u64 cl = gl_add(acc_reduce(al%d), D[%d]), cn = gl_add(acc_reduce(an%d), D[%d]);" % (i, d_job[i], i, d_job[i]))
            if twc.filter_column is not None:   # select(f, x) = f*x + 1 - f
                e.add("    cl = gl_sub(gl_add(gl_mul(fl, cl), 1), fl); cn = gl_sub(gl_add(gl_mul(fn, cn), 1), fn);")
            e.add("    const u64 zl = ZL(%d), zn = ZN(%d);" % (nperm + i, nperm + i))
            e.idx = first_idx + 2 * i
            e.emit(KIND_FIRST, "gl_sub(zl, cl)")
            e.emit(KIND_TRANSITION, "gl_sub(zn, gl_mul(zl, cn))")
            e.add("  }")
        e.add("}")
        e.add(BARRIER)
    e.idx = first_idx + 2 * len(jobs)
    assert e.idx == K, (e.idx, K)
    e.add("AIRQ_EPILOGUE")
    head = "// table %d (%s): %d columns, %d constraints, %d permutation Zs, %d CTL Zs, K = %d\n" % (
        t, tab.name, tab.ncols, len(tab.emits), nperm, len(jobs), K)
    src = head + "__global__ __launch_bounds__(AIRQ_THREADS) void %s(QuotParams P) {\n%s\n}\n" % (name, "\n".join(e.lines))
    return src, K


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
            src, K = table_kernel(s, t, name)
            tab = s.tables[t]
            files[name + ".hip"] = (
                "// GENERATED by olavm_amd/air/codegen.py -- do not edit; regenerated by __graft_entry__.build().\n"
                "#define AIRQ_GENERATED_TU 1\n#include \"../airq.cuh\"\nnamespace ola {\n" + src +
                "extern const AirKernelEntry %s_entry;\nconst AirKernelEntry %s_entry = {0x%016Xull, %s, %d, %d, %d, \"%s\"};\n}  // namespace ola\n"
                % (name, name, sig, name, K, tab.n_params, tab.num_permutation_batches(), tab.name))
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
