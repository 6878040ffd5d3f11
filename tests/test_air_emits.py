"""The AIR transcription against the reference's SOURCE, mechanically (PARITY.md "AIR emission order").

tests/golden/air_emit_kinds.json is written by tools/extract_air_emits.py, which walks `eval_packed_generic` of the twelve `*_stark.rs`
files of the reference (and the cpu/*.rs opcode files and lookup.rs they call) as a control-flow interpreter: every
`yield_constr.constraint / _transition / _first_row / _last_row` in emission order, loops expanded from the source's own constants;
plus COLUMNS, constraint_degree(), the number of permutation pairs, and the cross-table lookups of stark/ola_stark.rs:122-560.
The emission order and kinds decide which alpha power and which selector every constraint meets (constraint_consumer.rs:34-78), and
with them every quotient byte -- and they were the one part of the proof that only the builder's reading pinned: oracle, verifier
restatement and GPU all consume olavm_amd/air/ola_tables.py.  Here that restatement is compared with what the walker read."""
import json
import os

import pytest

from olavm_amd.air import dsl, ola_tables as T

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIXTURE = os.path.join(HERE, "golden", "air_emit_kinds.json")
KIND = {dsl.KIND_ALL: "all", dsl.KIND_TRANSITION: "transition", dsl.KIND_FIRST: "first_row", dsl.KIND_LAST: "last_row"}
TABLE_INDEX = {"Cpu": 0, "Memory": 1, "Bitwise": 2, "Cmp": 3, "RangeCheck": 4, "Poseidon": 5, "PoseidonChunk": 6, "StorageAccess": 7, "Tape": 8,
               "SCCall": 9, "Program": 10, "ProgChunk": 11}       # enum Table, stark/ola_stark.rs:103-118


@pytest.fixture(scope="module")
def golden():
    return json.load(open(FIXTURE))


def test_every_table_was_walked(golden):
    assert golden["problems"] == []
    assert [t["table"] for t in golden["tables"]] == list(TABLE_INDEX)
    assert all("emit_kinds" in t for t in golden["tables"])
    assert sum(len(t["emit_kinds"]) for t in golden["tables"]) == 669


def test_emission_order_and_kinds_equal_the_reference_source(golden):
    stark = T.ola_stark()
    for ours, ref in zip(stark.tables, golden["tables"]):
        kinds = [KIND[k] for k, _ in ours.emits]
        assert len(kinds) == len(ref["emit_kinds"]), (ref["table"], len(kinds), len(ref["emit_kinds"]))
        for i, (a, b) in enumerate(zip(kinds, ref["emit_kinds"])):
            assert a == b, f"{ref['table']}: emit {i} is `{a}` here, `{b}` at {ref['emit_sites'][i]} in the reference"
        assert ours.ncols == ref["columns"], ref["table"]
        assert ours.constraint_degree == ref["constraint_degree"], ref["table"]
        assert len(ours.permutation_pairs) == ref["permutation_pairs"], ref["table"]
        if ref["permutation_pairs"]:       # PermutationPair::singletons(lhs, rhs), both columns evaluated from the source
            assert [[list(map(int, p)) for p in pair] for pair in ours.permutation_pairs] == [[pr] for pr in ref["permutation_pair_columns"]], ref["table"]


def test_cells_the_reference_names_directly_are_read_by_the_same_constraint_here(golden):
    """For 495 of the 669 constraints the walker could read trace cells off the argument of the reference's `yield_constr` call, directly or
    through immutable locals of the same function (`lv[COL_X]`, `wrapper.nv[COL_Y.start + i]`, `let lv_is_padding = lv[COL_IS_PADDING]`:
    1 309 cells, indices evaluated from the source's constants and loop variables).  Each of
    them must be a cell the transcription's expression for THAT emit reads -- a swapped column or a local / next mix-up in a direct
    reference, or an emit attached to the wrong expression, fails here."""
    from olavm_amd.air import codegen
    stark = T.ola_stark()
    named = 0
    for ours, ref in zip(stark.tables, golden["tables"]):
        full = codegen._emit_cells(ours)
        assert len(full) == len(ref["emit_direct_cells"])
        for i, cells in enumerate(ref["emit_direct_cells"]):
            have = {r + str(c) for r, c in full[i]}
            for c in cells:
                assert c in have, f"{ref['table']}: emit {i} ({ref['emit_sites'][i]}) names {c} in the reference; the transcription's emit {i} reads {sorted(have)}"
                named += 1
    assert named == 1309


def test_the_blob_the_library_reads_carries_the_same_header(golden):
    """include/ola_airset.bin (what the Rust shim embeds, integration/rust/build.rs) = AirSet.blob(): table headers word for word"""
    import numpy as np
    blob = np.fromfile(os.path.join(ROOT, "include", "ola_airset.bin"), dtype=np.uint64)
    stark = T.ola_stark()
    assert np.array_equal(blob, stark.blob())
    assert int(blob[2]) == 12 and int(blob[3]) == len(golden["cross_table_lookups"]) == 19
    pos = 4
    for t, ref in zip(stark.tables, golden["tables"]):
        w = t.words()
        assert [int(x) for x in blob[pos:pos + 2]] == [ref["columns"], ref["constraint_degree"]] and int(blob[pos + 4]) == ref["permutation_pairs"]
        pos += len(w)


def test_cross_table_lookups_equal_the_reference_source(golden):
    stark = T.ola_stark()
    assert len(stark.ctls) == len(golden["cross_table_lookups"])
    for ours, ref in zip(stark.ctls, golden["cross_table_lookups"]):
        assert ours.looked_table.table == TABLE_INDEX[ref["looked"]["table"]], ref["name"]
        seq = [TABLE_INDEX[r["table"]] for r in ref["looking_in_source_order"]]
        assert [t.table for t in ours.looking_tables] == seq, ref["name"]
        assert sum(ref["looking"].values()) == len(ours.looking_tables)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")
def test_fixture_is_what_the_extractor_reads_today():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import extract_air_emits as X
    assert json.dumps(X.extract("/root/reference"), indent=1) + "\n" == open(FIXTURE).read()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")
def test_the_walker_notices_a_changed_loop_bound_and_a_swapped_kind(tmp_path):
    """the check has teeth: a copy of the reference with one loop bound changed / one emit kind swapped reads differently"""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import extract_air_emits as X
    ref = tmp_path / "ref"
    for sub in ("circuits/src", "core/src", "plonky2/plonky2/src/hash"):
        shutil.copytree(os.path.join("/root/reference", sub), ref / sub)
    base = {t["table"]: t["emit_kinds"] for t in X.extract(str(ref))["tables"]}
    p = ref / "circuits/src/cpu/call_sc.rs"
    s = p.read_text()
    assert "for i in 0..REGISTER_NUM {" in s
    p.write_text(s.replace("for i in 0..REGISTER_NUM {", "for i in 0..REGISTER_NUM - 1 {", 1))
    X.Src.cache.clear()
    changed = {t["table"]: t["emit_kinds"] for t in X.extract(str(ref))["tables"]}
    assert len(changed["Cpu"]) < len(base["Cpu"]) and all(changed[k] == base[k] for k in base if k != "Cpu")
    p.write_text(s)
    q = ref / "circuits/src/builtins/tape/tape_stark.rs"
    s = q.read_text()
    q.write_text(s.replace("yield_constr.constraint_transition(", "yield_constr.constraint(", 1))
    X.Src.cache.clear()
    changed = {t["table"]: t["emit_kinds"] for t in X.extract(str(ref))["tables"]}
    assert changed["Tape"] != base["Tape"] and len(changed["Tape"]) == len(base["Tape"])
    # a direct reference to another column is read as such
    q.write_text(s.replace("nv[COL_TAPE_IS_INIT_SEG]", "nv[COL_TAPE_OPCODE]", 1))
    X.Src.cache.clear()
    cells0 = X.extract("/root/reference")["tables"][8]["emit_direct_cells"]
    X.Src.cache.clear()
    cells1 = X.extract(str(ref))["tables"][8]["emit_direct_cells"]
    assert cells0 != cells1
    X.Src.cache.clear()


def test_lookup_columns_equal_the_columns_the_references_functions_name(golden):
    """Per TableWithColumns the reference constructs (19 looked + 69 looking entries, stark/ola_stark.rs:146-560): the column indices its
    `ctl_data_*` and `ctl_filter_*` functions name -- every constant expression of the function's body evaluated from the source, with
    the entry's argument bound (`ctl_data_with_mem_src(3)`) -- against the columns the transcription's entry reads.  Equal where the
    function names exactly what it returns (80 of 88 entries); a subset where the body selects among named columns (`match i`)."""
    stark = T.ola_stark()

    def cols_of(cols):
        return {int(c) for col in cols for c, _ in col.terms}

    checked = exact = 0
    for ours, ref in zip(stark.ctls, golden["cross_table_lookups"]):
        pairs = [(ours.looked_table, ref["looked"])] + list(zip(ours.looking_tables, ref["looking_in_source_order"]))
        assert len(pairs) == 1 + len(ours.looking_tables) == 1 + len(ref["looking_in_source_order"])
        for twc, r in pairs:
            assert twc.table == TABLE_INDEX[r["table"]], (ref["name"], r["data_fn"])
            have, want = cols_of(twc.columns), set(r["data_columns"])
            assert have <= want and (have == want or not r["data_exact"]), (ref["name"], r["data_fn"], sorted(have), sorted(want))
            assert (twc.filter_column is None) == ("filter_fn" not in r), (ref["name"], r["data_fn"])
            if twc.filter_column is not None:
                have, want = cols_of([twc.filter_column]), set(r["filter_columns"])
                assert have <= want and (have == want or not r["filter_exact"]), (ref["name"], r["filter_fn"], sorted(have), sorted(want))
            checked += 1
            exact += bool(r["data_exact"] and r.get("filter_exact", True))
    assert checked == 88 and exact >= 80
