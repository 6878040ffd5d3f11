"""The reference's transforms -- the path BASELINE.json's metric is quoted on -- RUN from their source (PARITY.md "transforms by
interpretation").

tools/rust_air_eval.py --ntt interprets plonky2/field/src/cfft/mod.rs with the build's feature set ("parallel"): `evaluate_poly` (:22),
`evaluate_poly_with_offset` (:65, domain offset `F::coset_shift()`, blowup 8: the commitment's LDE), `interpolate_poly` (:128) and
`interpolate_poly_with_offset` (:180), which dispatch to serial.rs (`fft_in_place`, `permute`) below 2^10 elements and to
concurrent.rs (`split_radix_fft`: transpose, inner FFTs, transpose, twiddles, outer FFTs; the batched `permute`, `clone_and_shift`) from
2^10 on -- sizes 2^1 .. 2^16, i.e. both shapes of the four-step split (square at even, 2:1 at odd exponents) and, from 2^14 on, the sizes
the GPU transforms on its T-form pass kernels (the kernels of the headline figure).  tests/golden/ref_ntt_vectors.json holds
a digest of each output.  Here the oracle's transforms (CPU) and the GPU's (`ola_ntt_batch`, -m gpu) are held to them; the larger sizes
are compared GPU against oracle elsewhere (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIXTURE = os.path.join(HERE, "golden", "ref_ntt_vectors.json")
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def vectors():
    return json.load(open(FIXTURE))["vectors"]


def record(words):
    import rust_air_eval as R
    return R.ntt_record([int(x) for x in words])


def same(v, words):
    r = record(words)
    return all(r[k] == v[k] for k in ("sha256", "len", "head", "tail"))


def test_fixture_covers_both_code_paths_and_all_four_transforms(vectors):
    assert len(vectors) == 64
    assert {(v["op"], v["log_n"]) for v in vectors} == {(op, n) for op in ("evaluate_poly", "interpolate_poly", "evaluate_poly_with_offset",
                                                                             "interpolate_poly_with_offset") for n in range(1, 17)}
    assert {v["path"] for v in vectors if v["log_n"] >= 10} == {"concurrent.rs"} and {v["path"] for v in vectors if v["log_n"] < 10} == {"serial.rs"}
    assert all(v["len"] == (8 if v["op"] == "evaluate_poly_with_offset" else 1) << v["log_n"] for v in vectors)


def test_oracle_transforms_equal_the_interpreted_reference(vectors, oracle):
    import rust_air_eval as R
    fn = {"evaluate_poly": oracle.evaluate_poly, "interpolate_poly": oracle.interpolate_poly,
          "evaluate_poly_with_offset": oracle.evaluate_poly_with_offset, "interpolate_poly_with_offset": oracle.interpolate_poly_with_offset}
    for v in vectors:
        x = np.array(R.ntt_input(v["log_n"], v["op"]), dtype=np.uint64)
        assert same(v, fn[v["op"]](x)), (v["op"], v["log_n"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")
def test_vectors_are_what_the_interpreter_computes_today(vectors):
    """the serial path at three sizes and the concurrent path at 2^10 (all four transforms), re-derived; and the reference's own inverse
    pairs: interpolate(evaluate(x)) = x through its code"""
    import rust_air_eval as R
    sys.setrecursionlimit(20000)
    it = R.ntt_interp("/root/reference")
    by = {(v["op"], v["log_n"]): v for v in vectors}
    for log_n in (1, 4, 9, 10):
        for op in R.NTT_OPS:
            assert same(by[(op, log_n)], R.ntt_run(it, op, log_n)), (op, log_n)
    for log_n in (3, 10):
        n = 1 << log_n
        x = [R.Fe(w) for w in R.stream_for(77, log_n, n)]
        y = list(x)
        it.call_free(it.cfft, "evaluate_poly", [y, it.call_free(it.cfft, "get_twiddles", [n])])
        assert [a.v for a in y] != [a.v for a in x]
        it.call_free(it.cfft, "interpolate_poly", [y, it.call_free(it.cfft, "get_inv_twiddles", [n])])
        assert [a.v for a in y] == [a.v for a in x]


@pytest.mark.gpu
def test_gpu_transforms_equal_the_interpreted_reference(vectors):
    import rust_air_eval as R
    from olavm_amd import backend as B
    be = B.Backend()
    op = {"evaluate_poly": (B.OLA_NTT_EVALUATE, 0), "interpolate_poly": (B.OLA_NTT_INTERPOLATE, 0),
          "evaluate_poly_with_offset": (B.OLA_NTT_COSET_LDE, 3), "interpolate_poly_with_offset": (B.OLA_NTT_COSET_INTERPOLATE, 0)}
    for v in vectors:
        x = np.array(R.ntt_input(v["log_n"], v["op"]), dtype=np.uint64)
        code, blowup_log = op[v["op"]]
        out = be.ntt(code, x, shift=7, blowup_log=blowup_log)
        assert same(v, out.reshape(-1)), (v["op"], v["log_n"])
        # a batch of three columns: the middle one is the vector's
        batch = np.stack([x[::-1].copy(), x, (x + np.uint64(1))])
        assert same(v, be.ntt(code, batch, shift=7, blowup_log=blowup_log)[1]), (v["op"], v["log_n"], "batched")
    be.close()
