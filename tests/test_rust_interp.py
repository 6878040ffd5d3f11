"""The interpreter that runs the reference's source (tools/rust_air_eval.py) against the Rust language itself, on small functions whose
values are fixed by the language (tests/rust_snippets/semantics.rs): the constructs the reference's prover and verifier lean on and that an
interpreter written in a language with reference semantics gets wrong most easily -- scalars behind `&mut`, `*x = array`, element references
out of `iter_mut()` / `&mut xs` (also through `skip` / `zip`), closures that assign captured variables, `chunks_mut`, assignments from nested
branches, `?`, `bool::then` + `Option::map`, integer widths, arrays copied out of a borrowed struct.  Each of them was at some point the
reason a run of the reference's code disagreed with this repository's provers -- and each time the interpreter, not the prover, was wrong."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
SNIPPETS = os.path.join(HERE, "rust_snippets", "semantics.rs")


@pytest.fixture(scope="module")
def run():
    import extract_air_emits as X
    import rust_air_eval as R
    sys.setrecursionlimit(20000)

    class NoTree:                      # the snippets need no reference tree
        root = os.path.dirname(SNIPPETS)
        files = []
        consts = {}
        const_files = set()

        def find_fn(self, name, here, module=None):
            return SNIPPETS if name in X.Src.get(SNIPPETS).functions() else None

        def near(self, cands, here):
            return cands[0] if cands else None

        def index_consts(self, f):
            pass
    it = R.Interp(NoTree())
    return lambda name, *args: it.call_free(SNIPPETS, name, list(args)), R


def test_mutation_through_references(run):
    call, R = run
    assert call("mut_scalar_through_calls") == 3
    assert call("deref_assign_array") == [4, 3, 2, 1]
    assert call("iter_mut_zip") == [11, 22, 33]
    assert call("iter_mut_skip_take") == [1, 20, 30, 4]
    assert call("chunks_mut_for_each") == [1, 1, 1, 1, 12, 12, 12, 12]
    assert call("swaps") == [0, 4, 2, 6, 1, 5, 3, 7]
    before, after = call("array_leaves_by_copy")
    assert before == [1, 2, 3] and after == [9, 2, 3]


def test_scopes_and_closures(run):
    call, R = run
    assert call("closure_assigns_captured") == [0, 0, 1, 3, 6]
    assert [call("nested_assignment", n) for n in (5, 50, 500)] == [7, 2, 1]
    assert call("loop_until") == 3
    assert call("shift_assign") == 5


def test_results_and_options(run):
    call, R = run
    assert call("question_mark", 0) == 2
    r = call("question_mark", 2)
    assert isinstance(r, R.Enum) and r.variant == "Err"
    r = call("question_mark", 7)
    assert isinstance(r, R.Enum) and r.variant == "Err"
    assert call("uses_option", True) == 3 and call("uses_option", False) == 0


def test_integer_widths(run):
    call, R = run
    assert [int(x) for x in call("le_bytes")] == [0x78, 0x56, 0x34, 0x12]
    assert tuple(int(x) for x in call("casts")) == (0xEF, 40)
