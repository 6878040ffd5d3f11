"""The AIR arithmetic against the reference's SOURCE, by evaluation (PARITY.md "AIR arithmetic by evaluation").

tools/rust_air_eval.py interprets the subset of Rust the reference's AIR files are written in and RUNS them: `eval_packed_generic`
of the twelve `*_stark.rs` files with everything they call -- `cpu/*.rs`, `CpuAdjacentRowWrapper::from_vars`, `lookup.rs`,
`reduce_with_powers`, the Poseidon layers of `core/src/util/poseidon_utils.rs` with plonky2's constant tables,
`OlaOpcode::binary_bit_mask` -- on pseudo-random rows and on constant rows (0, 1, p - 1, ADDR_HEAP_PTR: the memory table's
`is_zero` branch), recording the value of every emitted constraint; and every `ctl_data_*` / `ctl_filter_*` function of
stark/ola_stark.rs's 88 lookup entries, whose `Column`s are built with the reference's constructors and evaluated with the
reference's `Column::eval`.  tests/golden/air_eval_vectors.json holds the values.

Here the hand transcription (olavm_amd/air/ola_tables.py = include/ola_airset.bin: what the oracle prover, the verifier
restatement and the GPU all consume) is evaluated on the same rows: 669 constraints x 6 rows and 88 lookup entries x 3 rows, value
for value, kind for kind.  Together with tests/test_air_emits.py (order, kinds, columns named) this takes the twelve AIRs out of
the "builder's reading" column: a wrong coefficient, a swapped operand, a missing term in any constraint or lookup column changes
a value on a random row."""
import json
import os

import pytest

from olavm_amd.air import dsl, ola_tables as T

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIXTURE = os.path.join(HERE, "golden", "air_eval_vectors.json")
P = dsl.P
KIND = {dsl.KIND_ALL: "all", dsl.KIND_TRANSITION: "transition", dsl.KIND_FIRST: "first_row", dsl.KIND_LAST: "last_row"}


def rows_for(seed, table, ncols):
    """tools/rust_air_eval.py rows_for: splitmix64 mod p, or one value in every cell"""
    if isinstance(seed, dict):
        return [seed["fill"] % P] * ncols, [seed["fill"] % P] * ncols
    m = 2**64 - 1
    x = (seed * 0x9E3779B97F4A7C15 + table * 0xD1342543DE82EF95 + 0x1234567) & m
    out = []
    for _ in range(2 * ncols):
        x = (x + 0x9E3779B97F4A7C15) & m
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        z ^= z >> 31
        out.append(z % P)
    return out[:ncols], out[ncols:]


def eval_emits(tab, lv, nv, params):
    """the table's constraint program in Python integers -> [(kind, value)] in emission order"""
    val = [0] * len(tab.nodes)
    for j, (op, a, b) in enumerate(tab.nodes):
        if op == dsl.OP_LOCAL:
            v = lv[a]
        elif op == dsl.OP_NEXT:
            v = nv[a]
        elif op == dsl.OP_CONST:
            v = a
        elif op == dsl.OP_PARAM:
            v = params[a]
        elif op == dsl.OP_ADD:
            v = (val[a] + val[b]) % P
        elif op == dsl.OP_SUB:
            v = (val[a] - val[b]) % P
        elif op == dsl.OP_MUL:
            v = (val[a] * val[b]) % P
        else:
            assert op == dsl.OP_ISZERO
            v = 1 if val[a] == 0 else 0
        val[j] = v
    return [(KIND[k], val[i]) for k, i in tab.emits]


@pytest.fixture(scope="module")
def vectors():
    return json.load(open(FIXTURE))


def test_every_table_and_lookup_was_evaluated(vectors):
    assert vectors["problems"] == []
    assert all("points" in t and len(t["points"]) == 6 for t in vectors["tables"])
    assert sum(len(t["points"][0]["values"]) for t in vectors["tables"]) == 669
    assert sum(len(c["entries"]) for c in vectors["cross_table_lookups"]) == 19 + 69
    # the heap-pointer rows are in: the is_zero branch of the memory table was taken both ways
    assert any(isinstance(p["seed"], dict) and p["seed"]["fill"] == T.ADDR_HEAP_PTR for p in vectors["tables"][1]["points"])


def test_constraint_values_equal_the_interpreted_reference(vectors):
    stark = T.ola_stark()
    compared = 0
    for index, (tab, ref) in enumerate(zip(stark.tables, vectors["tables"])):
        for pt in ref["points"]:
            lv, nv = rows_for(pt["seed"], index, tab.ncols)
            ours = eval_emits(tab, lv, nv, [vectors["param"]] * max(1, tab.n_params))
            assert len(ours) == len(pt["values"]), ref["table"]
            for i, ((kind, value), rkind, rvalue) in enumerate(zip(ours, pt["kinds"], pt["values"])):
                assert kind == rkind and value == rvalue, (
                    f"{ref['table']}: constraint {i} ({ref['emit_sites'][i]}) on rows {pt['seed']}: the transcription gives {kind} {value}, "
                    f"the reference's source evaluates to {rkind} {rvalue}")
                compared += 1
    assert compared == 669 * 6
    # not vacuous: on random rows (almost) every constraint is non-zero
    nonzero = sum(1 for t in vectors["tables"] for v in t["points"][0]["values"] if v)
    assert nonzero >= 660


def test_lookup_column_values_equal_the_interpreted_reference(vectors):
    stark = T.ola_stark()

    def col_eval(col, row):
        return (sum(int(f) * row[int(c)] for c, f in col.terms) + int(col.constant)) % P

    compared = 0
    for ours, ref in zip(stark.ctls, vectors["cross_table_lookups"]):
        twcs = [ours.looked_table] + list(ours.looking_tables)
        assert len(twcs) == len(ref["entries"]), ref["name"]
        for twc, e in zip(twcs, ref["entries"]):
            for pt in e["points"]:
                row, _ = rows_for(pt["seed"], twc.table, stark.tables[twc.table].ncols)
                assert [col_eval(c, row) for c in twc.columns] == pt["data"], (ref["name"], e["data_fn"], pt["seed"])
                assert (None if twc.filter_column is None else col_eval(twc.filter_column, row)) == pt["filter"], (ref["name"], e["filter_fn"], pt["seed"])
                compared += 1
    assert compared == 88 * 3


def stream_for(seed, salt, count):
    """tools/rust_air_eval.py stream_for: the Z openings and challenges of the vanishing-polynomial vectors"""
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


def test_whole_vanishing_polynomial_equals_the_interpreted_reference(vectors):
    """`eval_vanishing_poly` (vanishing_poly.rs:20-45) as the reference computes it -- the table's AIR, then `eval_permutation_checks`
    (permutation.rs:302-360: Z(1) = 1 per batch, then Z(gx) prod(rhs) = Z(x) prod(lhs) with `get_permutation_batches`' chunking), then
    `eval_cross_table_lookup_checks` (cross_table_lookup.rs:380-421) over the `CtlCheckVars` that the reference's own
    `CtlCheckVars::from_proofs` (:330-378) builds from `all_cross_table_lookups()` -- interpreted from the source for all twelve tables
    (1 049 constraints) on two sets of rows, Z openings and challenges.  The same sequence from the AIR data in Python: what
    olavm_amd/air/codegen.py prints into the quotient kernels and oracle/stark.cpp evaluates -- kinds, order (in particular the order
    in which a table's lookup Z columns are consumed) and values."""
    stark = T.ola_stark()
    nch, zper = 2, 256

    def col_eval(col, row):
        return (sum(int(f) * row[int(c)] for c, f in col.terms) + int(col.constant)) % P

    assert len(vectors["vanishing_poly"]) == 2
    compared = 0
    for vs in vectors["vanishing_poly"]:
        seed = vs["seed"]
        cc = stream_for(seed, 99, 2 * nch)
        for t, (tab, ref) in enumerate(zip(stark.tables, vs["tables"])):
            lv, nv = rows_for(seed, t, tab.ncols)
            z = stream_for(seed, 100 + t, 2 * zper)
            zl, zn = z[:zper], z[zper:]
            emits = eval_emits(tab, lv, nv, [vectors["param"]] * max(1, tab.n_params))
            nperm, bs = tab.num_permutation_batches(nch), tab.quotient_degree_factor
            assert nperm == ref["num_permutation_zs"], tab.name
            if tab.permutation_pairs:
                pc = stream_for(seed, 200 + t, 2 * bs * nch)
                emits += [("first_row", (zl[b] - 1) % P) for b in range(nperm)]
                total, inst = len(tab.permutation_pairs) * nch, 0
                for b in range(nperm):
                    pl = pr = 1
                    for i in range(bs):
                        if inst >= total:
                            break
                        pair, c = tab.permutation_pairs[inst // nch], inst % nch
                        beta, gamma = pc[2 * (i * nch + c)], pc[2 * (i * nch + c) + 1]
                        pl = pl * (sum(lv[lc] * pow(beta, k, P) for k, (lc, _) in enumerate(pair)) + gamma) % P
                        pr = pr * (sum(lv[rc] * pow(beta, k, P) for k, (_, rc) in enumerate(pair)) + gamma) % P
                        inst += 1
                    emits.append(("all", (zn[b] * pr - zl[b] * pl) % P))
            k = 0
            for ctl in stark.ctls:                    # AirSet.ctl_jobs order: per lookup, per challenge, this table's looking entries, then the looked one
                for c in range(nch):
                    beta, gamma = cc[2 * c], cc[2 * c + 1]
                    for twc in list(ctl.looking_tables) + [ctl.looked_table]:
                        if twc.table != t:
                            continue
                        sel = []
                        for row in (lv, nv):
                            combo = (sum(col_eval(col, row) * pow(beta, j, P) for j, col in enumerate(twc.columns)) + gamma) % P
                            if twc.filter_column is not None:
                                f = col_eval(twc.filter_column, row)
                                combo = (f * combo + 1 - f) % P
                            sel.append(combo)
                        emits.append(("first_row", (zl[nperm + k] - sel[0]) % P))
                        emits.append(("transition", (zn[nperm + k] - zl[nperm + k] * sel[1]) % P))
                        k += 1
            assert k == ref["num_ctl_zs"] == len(stark.ctl_jobs(t, nch)), tab.name
            assert [e[0] for e in emits] == ref["kinds"], tab.name
            assert [e[1] for e in emits] == ref["values"], tab.name
            compared += len(emits)
    assert compared == 2 * 1049


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")
def test_vectors_are_what_the_interpreter_computes_today_and_it_notices_a_changed_formula(tmp_path):
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import extract_air_emits as X
    import rust_air_eval as R
    sys.setrecursionlimit(20000)
    X.Src.cache.clear()
    data = R.extract("/root/reference", 2)
    assert json.dumps(data, separators=(",", ":")) + "\n" == open(FIXTURE).read()
    # teeth: one coefficient of one constraint changed in a copy of the reference -> exactly that table's values move
    ref = tmp_path / "ref"
    for sub in ("circuits/src", "core/src", "plonky2/plonky2/src/hash", "plonky2/plonky2/src/plonk"):
        shutil.copytree(os.path.join("/root/reference", sub), ref / sub)
    p = ref / "circuits/src/cpu/call.rs"
    s = p.read_text()
    assert "P::Scalar::from_canonical_u64(2)" in s or "P::ONES" in s
    mutated = s.replace("P::ONES", "(P::ONES + P::ONES)", 1)
    assert mutated != s
    p.write_text(mutated)
    X.Src.cache.clear()
    other = R.extract(str(ref), 2)
    X.Src.cache.clear()
    same = [a["points"] == b["points"] for a, b in zip(data["tables"], other["tables"])]
    assert same == [False] + [True] * 11
