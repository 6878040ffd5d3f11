"""GPU parity of the multi-table STARK prover (ola_prove_with_traces): AllProof bytes identical to the oracle's and
accepted by the oracle's restatement of the reference verifier."""
import numpy as np
import pytest

from olavm_amd.air import AirSet, ola_tables as T
from tests import tracegen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from olavm_amd.backend import Backend
    b = Backend(device=0)
    yield b
    b.close()


def mini_set(range_bits):
    return AirSet([T.cmp_table(), T.rangecheck_table(range_bits)], [T.ctl_cmp_rangecheck(0, 1)])


@pytest.mark.parametrize("range_bits,n_cmp", [(4, 6), (5, 30), (7, 100)])
def test_cmp_rangecheck_proof_bytes_match_oracle(be, oracle, range_bits, n_cmp):
    rng = np.random.default_rng(range_bits * 7 + n_cmp)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, n_cmp, range_bits)
    blob = mini_set(range_bits).blob()
    got = be.prove_with_traces(blob, [cmp_t, rc_t])
    want = oracle.prove_with_traces(blob, [cmp_t, rc_t])
    assert len(got) == len(want)
    assert got == want
    rc, why = oracle.verify_all_proof(blob, got)
    assert rc == 0, why


def test_real_size_rangecheck_u16(be, oracle):
    """The reference's real RangeCheck table (2^16-row fixed table, BASE = 2^16) with a 2^10-row Cmp table: too slow for
    the oracle prover in a test, so the oracle VERIFIER checks the GPU proof (constraints at zeta, CTL, FRI)."""
    rng = np.random.default_rng(16)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, 1000, 16)
    assert rc_t.shape == (12, 1 << 16) and cmp_t.shape == (6, 1 << 10)
    blob = mini_set(16).blob()
    proof = be.prove_with_traces(blob, [cmp_t, rc_t])
    rc, why = oracle.verify_all_proof(blob, proof)
    assert rc == 0, why


def test_invalid_trace_yields_a_rejected_proof(be, oracle):
    """Cmp has quotient_degree_factor 2 = 2^qdb, so trim_to_len(n*q) (prover.rs:469-473) is vacuous exactly as in the
    reference (SURVEY F10 discusses the same effect): the prover still emits bytes, identical to the oracle's, and
    the verifier rejects them at the quotient identity."""
    rng = np.random.default_rng(3)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, 6, 4)
    cmp_t[T.COL_CMP_GTE, 2] ^= 1
    blob = mini_set(4).blob()
    proof = be.prove_with_traces(blob, [cmp_t, rc_t])
    assert proof == oracle.prove_with_traces(blob, [cmp_t, rc_t])
    rc, why = oracle.verify_all_proof(blob, proof)
    assert rc != 0 and "quotient" in why


def test_twelve_table_all_proof_bytes_match_oracle(be, oracle):
    """All 12 OlaStark tables and all 19 cross-table lookups (ola_stark.rs:29-64,122-560) on the empty-program instance
    with miniature fixed tables: AllProof bytes identical to the oracle's and accepted by its verifier."""
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    for log_n in (3, 8):        # 2^3 rows: interpreter kernel (tables below 256 rows); 2^8: the generated kernels
        traces, params, compress = tracegen.empty_program_instance(log_n=log_n, live=np.random.default_rng(log_n))
        got = be.prove_with_traces(blob, traces, params, compress)
        want = oracle.prove_with_traces(blob, traces, params, compress)
        assert got == want
        rc, why = oracle.verify_all_proof(blob, got, params)
        assert rc == 0, why


def test_twelve_table_larger_traces_verify(be, oracle):
    """Same AIR set with 2^10-row padding traces and the u8 bitwise table of the reference (2^18 rows): the oracle
    verifier accepts the GPU proof."""
    s = T.ola_stark(range_bits=8, limb_bits=8)
    blob = s.blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=8, limb_bits=8, live=np.random.default_rng(10))
    assert traces[2].shape == (59, 1 << 18)
    proof = be.prove_with_traces(blob, traces, params, compress)
    rc, why = oracle.verify_all_proof(blob, proof, params)
    assert rc == 0, why


@pytest.mark.parametrize("table", ["cmp", "rangecheck", "tape", "cpu", "bitwise", "program", "sccall", "storage_access", "poseidon",
                                   "poseidon_chunk", "prog_chunk", "memory"])
def test_constraint_interpreter_on_random_rows(be, oracle, table):
    """Random (invalid) traces drive every opcode of a table's constraint program with generic values.  When
    quotient_degree_factor is a power of two the prover still emits bytes (trim_to_len is vacuous) and they
    must equal the oracle's; otherwise both sides must refuse with the quotient-degree error (prover.rs:469-473)."""
    full = T.ola_stark(range_bits=4, limb_bits=2)
    i = [t.name for t in full.tables].index(table)
    tab = full.tables[i]
    # two copies of the table joined by one lookup, so that each has a CTL Z column (the prover refuses tables without
    # any Z polynomial, prover.rs "No CTL?")
    from olavm_amd.air.dsl import Col, CrossTableLookup, TableWithColumns
    s = AirSet([tab, tab], [CrossTableLookup([TableWithColumns(0, [Col.single(0), Col.single(tab.ncols - 1)], Col.single(1))],
                                             TableWithColumns(1, [Col.single(2), Col.single(3)]))])
    blob = s.blob()
    rng = np.random.default_rng(i)
    tr = [rng.integers(0, tracegen.P, size=(tab.ncols, 16), dtype=np.uint64) for _ in range(2)]
    tr[0][1] = rng.integers(0, 2, size=16, dtype=np.uint64)          # the lookup's filter column must be binary
    params = [int(x) for x in rng.integers(0, tracegen.P, size=2 * tab.n_params, dtype=np.uint64)] or None
    q = tab.quotient_degree_factor
    if q & (q - 1) == 0:
        assert be.prove_with_traces(blob, tr, params) == oracle.prove_with_traces(blob, tr, params)
    else:
        from olavm_amd.backend import OlaGpuError
        with pytest.raises(OlaGpuError, match="uotient"):
            be.prove_with_traces(blob, tr, params)
        with pytest.raises(RuntimeError, match="uotient"):
            oracle.prove_with_traces(blob, tr, params)


def test_non_binary_ctl_filter_is_refused(be, oracle):
    """cross_table_lookup.rs:303-305: a filter value outside {0,1} aborts the prover."""
    from olavm_amd.backend import OlaGpuError
    rng = np.random.default_rng(5)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, 6, 4)
    rc_t[T.RC_CMP_FILTER, 3] = 2
    blob = mini_set(4).blob()
    with pytest.raises(OlaGpuError, match="Non-binary filter"):
        be.prove_with_traces(blob, [cmp_t, rc_t])
    with pytest.raises(RuntimeError, match="Non-binary filter"):
        oracle.prove_with_traces(blob, [cmp_t, rc_t])


def _random_rows_with_binary_filters(rng, airset, t, n):
    """Uniform random trace for table t, except that the columns its CTL filters read are zero or one-hot per row, chosen
    among the assignments for which every filter (sums / differences / `1 - col` of selector columns) evaluates to 0 or
    1, as cross_table_lookup.rs:303 demands."""
    tab = airset.tables[t]
    tr = rng.integers(0, tracegen.P, size=(tab.ncols, n), dtype=np.uint64)
    filters = [j.filter_column for j in airset.ctl_jobs(t, 1) if j.filter_column is not None]
    cols = sorted({c for f in filters for c, _ in f.terms})
    if not cols:
        return tr

    def ok(hot):
        return all((sum(k for c, k in f.terms if c == hot) + f.constant) % tracegen.P in (0, 1) for f in filters)
    options = [h for h in [None] + cols if ok(h)]
    assert len(options) > 1, "no non-trivial binary filter assignment"
    tr[cols] = 0
    for r in range(n):
        h = options[int(rng.integers(0, len(options)))]
        if h is not None:
            tr[h, r] = 1
    return tr


@pytest.mark.parametrize("rows", [512, 64, 8, 2])
@pytest.mark.parametrize("t", range(12))
def test_specialised_quotient_kernels_match_interpreter_on_random_rows(be, t, rows, monkeypatch):
    """Every generated straight-line kernel against the interpreter on generic data (OLA_AIR_KERNELS=crosscheck compares
    all quotient values on the device).  Table t is the real one; the tables before it are replaced by constraint-free
    stand-ins of the same width (their vanishing polynomial -- CTL checks only -- is divisible for any trace), so the
    prover reaches table t.  The run may still end in the quotient-degree error for table t (random rows do not satisfy
    its constraints), which is raised after the comparison.  rows < 256 (round 6): the generated kernels with workgroups of `rows`
    threads, one per coset -- tables that small went to the interpreter before."""
    from olavm_amd.air.dsl import AirTable
    from olavm_amd.backend import OlaGpuError
    full = T.ola_stark(range_bits=4, limb_bits=2)
    tabs = [full.tables[i] if i >= t else AirTable("standin%d" % i, full.tables[i].ncols, 3) for i in range(12)]
    s = AirSet(tabs, full.ctls)
    assert s.signature(t) == full.signature(t)
    blob = s.blob()
    avail = be.air_kernels_available(blob, 12)
    assert avail[t] and not any(avail[:t])
    rng = np.random.default_rng(100 + t)
    traces = [_random_rows_with_binary_filters(rng, s, i, rows) for i in range(12)]
    params = [int(x) for tab in tabs for x in rng.integers(0, tracegen.P, size=tab.n_params, dtype=np.uint64)] or None
    monkeypatch.setenv("OLA_AIR_KERNELS", "crosscheck")
    try:
        be.prove_with_traces(blob, traces, params)
    except OlaGpuError as e:
        assert "Quotient has failed" in str(e), str(e)


def test_interpreter_and_specialised_kernels_give_the_same_proof(be, monkeypatch):
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    assert all(be.air_kernels_available(blob, 12))
    traces, params, compress = tracegen.empty_program_instance(log_n=9)
    fast = be.prove_with_traces(blob, traces, params, compress)
    monkeypatch.setenv("OLA_AIR_KERNELS", "interpreter")
    assert be.prove_with_traces(blob, traces, params, compress) == fast


def test_malformed_inputs_are_refused(be):
    """Error behaviour of ola_prove_with_traces: a damaged AIR-set blob, a table too large for the LDE domain and missing
    parameters are reported through error codes (never a crash, never proof bytes)."""
    from olavm_amd.backend import OlaGpuError
    rng = np.random.default_rng(3)
    cmp_t, rc_t = tracegen.cmp_rangecheck_instance(rng, 6, 4)
    blob = mini_set(4).blob()
    bad = blob.copy()
    bad[0] ^= 1                                                   # magic
    with pytest.raises(OlaGpuError, match="magic"):
        be.prove_with_traces(bad, [cmp_t, rc_t])
    with pytest.raises(OlaGpuError, match="truncated|trailing"):
        be.prove_with_traces(blob[:-3], [cmp_t, rc_t])
    with pytest.raises(OlaGpuError, match="trailing"):
        be.prove_with_traces(np.concatenate([blob, blob[:2]]), [cmp_t, rc_t])
    # a table with a CTL but whose partner references a table index that does not exist
    worse = blob.copy()
    worse[2] = 1                                                  # claim a single table: the CTL then points outside
    with pytest.raises(OlaGpuError):
        be.prove_with_traces(worse, [cmp_t])
    # the proof of a valid instance still works afterwards (no state is left behind by the failures)
    assert len(be.prove_with_traces(blob, [cmp_t, rc_t])) > 1000


def test_device_poseidon_trace_generation(be, oracle):
    """ola_generate_poseidon_trace against the reference's own rows (tests/golden/poseidon_air_rows.json: the ZERO-hash
    padding row and the row of hashing [1000, 1001, ...], both produced by the reference's executor), and -- for random
    inputs -- against the Poseidon AIR, under which every generated row must vanish."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_air_rows.json")))["rows"]
    rng = np.random.default_rng(8)
    n = 64
    inputs = rng.integers(0, tracegen.P, size=(12, n), dtype=np.uint64)
    inputs[:, 0] = 0
    inputs[:, 1] = np.array(g["1000"][4:16], dtype=np.uint64)
    inputs[9:12, 2:] = 0                     # capacity lanes stay free for the rows that carry a tree-key / storage filter
    filters = np.zeros((4, n), dtype=np.uint64)
    filters[0, 2::3] = 1                     # FILTER_LOOKED_NORMAL
    filters[1, 3::3] = 1                     # FILTER_LOOKED_TREEKEY (needs inputs 9..11 = 0)
    tr = be.generate_poseidon_trace(inputs, filters)
    assert [int(x) for x in tr[:, 0]] == g["ZERO"]
    assert [int(x) for x in tr[:, 1]] == g["1000"]
    assert np.array_equal(tr[:4], filters) and np.array_equal(tr[4:16], inputs)
    full = T.ola_stark(range_bits=4, limb_bits=2)
    assert oracle.check_constraints(full.blob(), 5, tr) == -1
    bad = tr.copy()
    bad[70, 5] = (int(bad[70, 5]) + 1) % tracegen.P
    assert oracle.check_constraints(full.blob(), 5, bad) == 5
    # and the outputs are the permutation
    for i in (0, 1, 7, 63):
        assert np.array_equal(tr[16:28, i], oracle.poseidon(inputs[:, i]))


@pytest.mark.parametrize("program", ["fibonacci", "mixed", "memory", "hash", "call", "tape", "storage", "heap", "storage_heavy"])
def test_real_execution_proof_bytes_match_oracle(be, oracle, program):
    """Traces of a real execution (olavm_amd/air/miniexec.py: live CPU opcodes, program fetches, hashed program chunks,
    bitwise / comparison / range-check lookups): AllProof bytes identical to the oracle's, verifier accepts."""
    from olavm_amd.air import miniexec as M
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    factory, kwargs = M.EXAMPLES[program]
    traces, params, compress = M.instance(factory(), **kwargs)
    got = be.prove_with_traces(blob, traces, params, compress)
    assert got == oracle.prove_with_traces(blob, traces, params, compress)
    rc, why = oracle.verify_all_proof(blob, got, params)
    assert rc == 0, why


def test_full_size_tables_proof_bytes_match_oracle(be, oracle):
    """ola_stark() as the reference instantiates it (2^16-entry range check, 2^18-row bitwise table with its sixteen
    permuted-column pairs) on the executor's 32-bit-operand program: AllProof bytes identical to the oracle's."""
    from olavm_amd.air import miniexec as M
    s = T.ola_stark()
    blob = s.blob()
    traces, params, compress = M.instance(M.wide_program(), range_bits=16, limb_bits=8)
    got = be.prove_with_traces(blob, traces, params, compress)
    assert got == oracle.prove_with_traces(blob, traces, params, compress)
    rc, why = oracle.verify_all_proof(blob, got, params)
    assert rc == 0, why


def test_too_small_proof_buffer_does_not_cost_a_second_proof(be):
    """A caller that under-sizes the output buffer gets OLA_E_INVALID_ARG with the needed size, and the finished proof from
    ola_take_pending_proof -- identical to what a large buffer returns."""
    import ctypes as C
    from olavm_amd.air import miniexec as M
    s = T.ola_stark(range_bits=4, limb_bits=2)
    traces, params, compress = M.instance(M.fibonacci(5))
    whole = be.prove_with_traces(s.blob(), traces, params, compress)
    assert be.prove_with_traces(s.blob(), traces, params, compress, cap=1000) == whole      # goes through ola_take_pending_proof
    need = C.c_size_t(0)
    assert be.lib.ola_take_pending_proof(be.ctx, C.create_string_buffer(16), 16, C.byref(need)) == -1    # nothing pending any more


def test_per_table_entry_point_reassembles_the_same_all_proof(be, oracle):
    """The reference's own structure kept on the host: commit every trace (PolynomialBatch::from_values), observe the caps,
    draw the CTL challenges, then one ola_prove_single_table per table on the shared transcript -- the concatenation is the
    AllProof of the one-call path, byte for byte."""
    import struct
    from olavm_amd.air import miniexec as M
    from olavm_amd.backend import Challenger
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    traces, params, compress = M.instance(M.mixed_program())
    whole = be.prove_with_traces(blob, traces, params, compress)
    batches = [be.commit(t) for t in traces]
    ch = Challenger()
    for b in batches:
        ch.observe(b.cap())
    ctl = [(ch.get(), ch.get()) for _ in range(2)]                 # (beta, gamma) per challenge, permutation.rs:190-216
    out, poff = struct.pack("<I", len(traces)), 0
    for t, tr in enumerate(traces):
        k = s.tables[t].n_params
        out += be.prove_single_table(blob, t, tr, batches[t], ctl, params[poff:poff + k], ch)
        poff += k
    out += struct.pack("<I", len(traces)) + b"".join(struct.pack("<Q", int(c)) for c in compress)
    assert out == whole
    rc, why = oracle.verify_all_proof(blob, out, params)
    assert rc == 0, why
    # a failing table leaves the caller's transcript where it was
    before = ch.state().copy()
    bad = traces[0].copy()
    bad[T.COL_DST, 3] = (int(bad[T.COL_DST, 3]) + 1) % tracegen.P          # a wrong sum in the CPU table
    bad_batch = be.commit(bad)
    with pytest.raises(Exception, match="not divisible"):
        be.prove_single_table(blob, 0, bad, bad_batch, ctl, [], ch)
    assert np.array_equal(before, ch.state())
    bad_batch.free()
    for b in batches:
        b.free()


def test_long_fibonacci_execution_verifies(be, oracle, monkeypatch):
    """1300 loop iterations = 7804 executed CPU rows (2^13-row CPU and program tables, specialised kernels, crosschecked
    against the interpreter kernel on this live data); the oracle verifier accepts the GPU proof."""
    from olavm_amd.air import miniexec as M
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    traces, params, compress = M.instance(M.fibonacci(1300))
    assert traces[0].shape == (94, 1 << 13)
    monkeypatch.setenv("OLA_AIR_KERNELS", "crosscheck")
    proof = be.prove_with_traces(blob, traces, params, compress)
    rc, why = oracle.verify_all_proof(blob, proof, params)
    assert rc == 0, why


@pytest.mark.parametrize("program", ["mixed", "fibonacci"])
def test_memory_lean_proof_is_the_same_proof(be, oracle, monkeypatch, program):
    """OLA_LEAN=1: no LDE is kept -- commitments hash one coset at a time, the quotient is evaluated coset by coset from
    re-derived values, opened rows are re-derived per queried coset.  Same transcript, so the AllProof bytes must equal the
    resident-mode proof's (and the oracle's), for tables on the specialised kernels and on the interpreter."""
    from olavm_amd.air import miniexec as M
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    traces, params, compress = M.instance(M.mixed_program() if program == "mixed" else M.fibonacci(300))
    monkeypatch.setenv("OLA_LEAN", "0")
    resident = be.prove_with_traces(blob, traces, params, compress)
    monkeypatch.setenv("OLA_LEAN", "1")
    be.memory_stats(reset=True)
    lean = be.prove_with_traces(blob, traces, params, compress)
    assert lean == resident
    if program == "mixed":
        assert lean == oracle.prove_with_traces(blob, traces, params, compress)
    monkeypatch.setenv("OLA_AIR_KERNELS", "crosscheck")
    assert be.prove_with_traces(blob, traces, params, compress) == resident


def test_phase_entry_points_reassemble_prove_single_table(be, oracle):
    """The reference's prove_single_table scope by scope (prover.rs:330-567), every scope one C-ABI call, the transcript on the
    host: compact -> permutation challenges -> ola_perm_z + ola_ctl_z ("compute permutation Z(x) polys") -> from_values of the Zs
    -> alphas -> ola_quotient ("compute quotient polys" + split) -> from_coeffs -> ola_open_and_prove.  The bytes equal
    ola_prove_single_table's for every table of an executed program (tables with and without permutation arguments)."""
    import struct
    from olavm_amd.air import miniexec as M
    from olavm_amd.backend import Challenger
    s = T.ola_stark(range_bits=4, limb_bits=2)
    blob = s.blob()
    traces, params, compress = M.instance(M.mixed_program())
    batches = [be.commit(t) for t in traces]
    ch = Challenger()
    for b in batches:
        ch.observe(b.cap())
    ctl = [(ch.get(), ch.get()) for _ in range(2)]
    ch_whole = ch.clone()

    def cap_bytes(cap):
        return struct.pack("<I", cap.shape[0]) + b"".join(struct.pack("<Q", int(x)) for x in cap.reshape(-1))

    poff, seen_perm = 0, 0
    for t, tr in enumerate(traces):
        shape = be.table_shape(blob, t)
        k = shape["n_params"]
        pr = params[poff:poff + k]
        poff += k
        want = be.prove_single_table(blob, t, tr, batches[t], ctl, pr, ch_whole)
        # --- the same, one scope at a time
        ch.compact()
        perm_ch = None
        if shape["perm_zs"]:
            seen_perm += 1
            perm_ch = [[(ch.get(), ch.get()) for _ in range(2)] for _ in range(shape["permutation_batch_size"])]
        n = tr.shape[1]
        zp = be.perm_z(blob, t, tr, perm_ch) if shape["perm_zs"] else np.zeros((0, n), dtype=np.uint64)
        zc = be.ctl_z(blob, t, tr, ctl)
        assert zp.shape[0] == shape["perm_zs"] and zc.shape[0] == shape["ctl_zs"]
        zs = be.commit(np.concatenate([zp, zc]))
        ch.observe(zs.cap())
        alphas = [ch.get(), ch.get()]
        chunks = be.quotient(blob, t, batches[t], zs, perm_ch, ctl, alphas, pr, n)
        qb = be.commit(chunks, from_coeffs=True)
        ch.observe(qb.cap())
        openings, fri = be.open_and_prove(batches[t], zs, qb, shape["perm_zs"], ch)
        got = cap_bytes(batches[t].cap()) + cap_bytes(zs.cap()) + cap_bytes(qb.cap()) + openings + fri
        assert got == want, s.tables[t].name
        assert np.array_equal(ch.state(), ch_whole.state())
        zs.free(); qb.free()
    assert seen_perm >= 3
    # a wrong trace surfaces where the reference's prover notices it: the quotient is not a polynomial of the right degree
    from olavm_amd.backend import OlaGpuError
    bad = traces[0].copy()
    bad[T.COL_DST, 3] = (int(bad[T.COL_DST, 3]) + 1) % tracegen.P
    bb = be.commit(bad)
    zc = be.ctl_z(blob, 0, bad, ctl)
    zb = be.commit(zc)
    with pytest.raises(OlaGpuError, match="not divisible"):
        be.quotient(blob, 0, bb, zb, None, ctl, [5, 7], [], bad.shape[1])
    for b in batches + [bb, zb]:
        b.free()


def test_device_resident_tables_give_the_same_proof(be, oracle):
    """ola_prove_with_traces looks at each table pointer: tables already in HBM (here: all of them, then a mix) are copied
    device to device instead of uploaded; the proof is the one the host tables give."""
    import torch
    from olavm_amd.air import miniexec as M
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = M.instance(M.mixed_program())
    want = be.prove_with_traces(blob, traces, params, compress)
    dev = [torch.from_numpy(np.ascontiguousarray(t).view(np.int64)).cuda() for t in traces]
    torch.cuda.synchronize()
    assert be.prove_with_traces(blob, dev, params, compress) == want
    mixed = [d if i % 2 == 0 else t for i, (d, t) in enumerate(zip(dev, traces))]
    assert be.prove_with_traces(blob, mixed, params, compress) == want
    for d, t in zip(dev, traces):           # the caller's tables are not modified
        assert np.array_equal(d.cpu().numpy().view(np.uint64), t)
