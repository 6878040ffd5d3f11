"""The whole-proof entry point from the reference's own trace type -- per table a list of columns, every column its own allocation
(circuits/src/stark/prover.rs:79-83, plonky2/field/src/polynomial/mod.rs:24-26) -- through ola_prove_with_traces_cols, the three
upload paths of olavm_amd/csrc/upload.h, and the `timed!` scopes handed back for the caller's TimingTree (ola_gpu_scope_times)."""
import numpy as np
import pytest

from olavm_amd.air import miniexec as M, ola_tables as T, tracegen
from olavm_amd.backend import Backend

pytestmark = pytest.mark.gpu

# every `timed!` name of the path: prover.rs:113,374,403,441,465,481,544; fri/oracle.rs:58,80,88,223; fri/prover.rs:43,56
# ("transpose LDEs", oracle.rs:84, has no counterpart: the LDE is produced in leaf order)
REFERENCE_SCOPES = {"compute trace commitments", "compute permutation Z(x) polys", "compute Zs commitment", "compute quotient polys",
                    "split quotient polys", "compute quotient commitment", "compute openings proof", "IFFT", "FFT + blinding",
                    "build Merkle tree", "fold codewords in the commitment phase", "find proof-of-work witness"}


@pytest.fixture(scope="module")
def be():
    b = Backend(device=0)
    yield b
    b.close()


def scattered(traces, rng):
    """every column a separate allocation, with gaps of odd sizes in between (an allocator's view of Vec<PolynomialValues<F>>)"""
    out, spacers = [], []
    for t in traces:
        cols = []
        for c in range(t.shape[0]):
            spacers.append(np.empty(int(rng.integers(1, 4096)), dtype=np.uint8))
            cols.append(np.array(t[c], dtype=np.uint64, copy=True))
        out.append(cols)
    return out


def test_columns_in_separate_allocations_prove_the_same_bytes(be, oracle):
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = M.instance(M.mixed_program())
    want = oracle.prove_with_traces(blob, traces, params, compress)
    assert be.prove_with_traces(blob, traces, params, compress) == want
    cols = scattered(traces, np.random.default_rng(5))
    assert be.prove_with_traces(blob, cols, params, compress) == want
    st = be.upload_stats()
    assert st["mode"] == "staged" and st["bytes"] == sum(t.size * 8 for t in traces)
    # tables may be mixed: some as blocks, some as column lists
    mixed = [cols[i] if i % 2 else traces[i] for i in range(len(traces))]
    assert be.prove_with_traces(blob, mixed, params, compress) == want


def test_upload_paths_and_ring_geometries_agree(be, oracle, monkeypatch):
    """Columns larger than a staging slot are cut into pieces, small ones travel together; a ring of 3 slots with 4 copier threads
    wraps around hundreds of times; one upload stream or two; the pageable path moves the same bytes."""
    blob = T.ola_stark().blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=18, log_n_mem=17)
    want = be.prove_with_traces(blob, traces, params, compress)
    rc, why = oracle.verify_all_proof(blob, want, params)
    assert rc == 0, why
    cols = scattered(traces, np.random.default_rng(6))
    for env in ({"OLA_UPLOAD_PIECE_MB": "1", "OLA_UPLOAD_SLOTS": "3", "OLA_UPLOAD_THREADS": "4"},
                {"OLA_UPLOAD_PIECE_MB": "1", "OLA_UPLOAD_SLOTS": "2", "OLA_UPLOAD_THREADS": "1"},
                {"OLA_UPLOAD_PIECE_MB": "1", "OLA_UPLOAD_SLOTS": "5", "OLA_UPLOAD_THREADS": "3", "OLA_UPLOAD_STREAMS": "1"},
                {"OLA_UPLOAD": "pageable"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert be.prove_with_traces(blob, cols, params, compress) == want, env
        assert be.prove_with_traces(blob, traces, params, compress) == want, env
        assert be.upload_stats()["mode"] == env.get("OLA_UPLOAD", "staged")
        for k in env:
            monkeypatch.delenv(k)


def test_narrow_columns_travel_as_32_bit_words(be, oracle, monkeypatch):
    """Columns whose words are all below 2^32 cross the link as 32-bit words (below 2^8: as bytes) and are widened on the device (upload.h); columns with
    field-sized values (hashes, products, inverses) travel whole, also when the first large word comes late in the column.  An executed
    program has all three kinds.  Same proof bytes with the packing on and off."""
    from olavm_amd.air import fastexec
    blob = T.ola_stark().blob()
    traces, params, compress = fastexec.instance(M.memory_program(6000), range_bits=16, limb_bits=8, max_steps=1 << 21)
    assert traces[0].shape[1] == 1 << 17
    monkeypatch.setenv("OLA_UPLOAD_PIECE_MB", "1")          # a 2^17-row column is 1 MB: every column of the large tables is a piece of its own
    want = be.prove_with_traces(blob, traces, params, compress)
    st = be.upload_stats()
    total = sum(t.size * 8 for t in traces)
    assert st["bytes"] == total and 0 < st["link_bytes"] < total, st
    big = [t for t in traces if t.shape[1] >= 1 << 17]
    wide_from_the_start = sum(int((t[:, :4096] >= (1 << 32)).any(axis=1).sum()) for t in big)
    wide_later_only = sum(int(((t >= (1 << 32)).any(axis=1) & ~(t[:, :4096] >= (1 << 32)).any(axis=1)).sum()) for t in big)
    narrow = sum(int((~(t >= (1 << 32)).any(axis=1)).sum()) for t in big)
    assert wide_from_the_start > 0 and wide_later_only > 0 and narrow > 0, (wide_from_the_start, wide_later_only, narrow)
    # at least the narrow columns of the large tables were halved (a piece is judged by itself: the narrow first half of a 2^18-row
    # column whose large words come later travels narrow, too)
    saved = total - st["link_bytes"]
    # round 6: columns whose words are all below 2^8 travel as bytes (7 of 8 bytes saved), the other narrow ones as 32-bit words
    byte_cols = sum(int((~(t >= (1 << 8)).any(axis=1)).sum()) * t.shape[1] for t in big)
    half_cols = sum(int((~(t >= (1 << 32)).any(axis=1)).sum()) * t.shape[1] for t in big) - byte_cols
    assert byte_cols > 0 and half_cols > 0
    assert byte_cols * 7 + half_cols * 4 <= saved < total * 7 // 8, st
    rc, why = oracle.verify_all_proof(blob, want, params)
    assert rc == 0, why
    monkeypatch.setenv("OLA_UPLOAD_PACK8", "0")             # 32-bit words only: round 5's path
    assert be.prove_with_traces(blob, traces, params, compress) == want
    st = be.upload_stats()
    saved = total - st["link_bytes"]
    assert (byte_cols + half_cols) * 4 <= saved < total // 2, st
    monkeypatch.delenv("OLA_UPLOAD_PACK8")
    # columns shorter than a slot get a piece of their own (OLA_UPLOAD_SOLO_KB): with 16 MB slots the 1 MB columns of this trace used to
    # share pieces and none of them was packed
    monkeypatch.setenv("OLA_UPLOAD_PIECE_MB", "16")
    assert be.prove_with_traces(blob, traces, params, compress) == want
    assert total - be.upload_stats()["link_bytes"] >= byte_cols * 7 + half_cols * 4
    monkeypatch.setenv("OLA_UPLOAD_SOLO_KB", "0")
    assert be.prove_with_traces(blob, traces, params, compress) == want
    assert be.upload_stats()["link_bytes"] == total
    monkeypatch.delenv("OLA_UPLOAD_SOLO_KB")
    monkeypatch.setenv("OLA_UPLOAD_PACK", "0")
    assert be.prove_with_traces(blob, traces, params, compress) == want
    st = be.upload_stats()
    assert st["link_bytes"] == st["bytes"] == total


def test_resident_and_host_tables_mixed_column_by_column(be, oracle):
    import torch
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = M.instance(M.fibonacci(30))
    want = oracle.prove_with_traces(blob, traces, params, compress)
    cols = scattered(traces, np.random.default_rng(7))
    dev = torch.from_numpy(np.ascontiguousarray(traces[0]).view(np.int64)).cuda()
    torch.cuda.synchronize()
    assert be.prove_with_traces(blob, [dev] + cols[1:], params, compress) == want


def test_timed_scopes_come_back_with_device_times(be, oracle):
    stark = T.ola_stark(range_bits=4, limb_bits=2)
    blob = stark.blob()
    traces, params, compress = M.instance(M.mixed_program())
    be.scope_times(enable=True)
    be.proof_stats(enable=True)
    try:
        proof = be.prove_with_traces(blob, traces, params, compress)
        sc = be.scope_times()
    finally:
        be.scope_times(enable=False)
        be.proof_stats(enable=False)
    assert proof == oracle.prove_with_traces(blob, traces, params, compress)        # recording does not disturb the proof
    names = {s["name"] for s in sc if s["reference"]}
    finals = {n for n in names if n.startswith("perform final FFT ")}
    assert names - finals == REFERENCE_SCOPES and finals
    assert all(s["name"] in REFERENCE_SCOPES or s["name"].startswith("perform final FFT ") for s in sc if s["reference"])
    # the reference's tree: "compute trace commitments" once at the top, the seven per-table scopes once per table, in its order
    top = [s for s in sc if s["reference"] and s["ref_depth"] == 0]
    per_table = ["compute permutation Z(x) polys", "compute Zs commitment", "compute quotient polys", "split quotient polys",
                 "compute quotient commitment", "compute openings proof"]
    assert top[0]["name"] == "compute trace commitments" and top[0]["table"] == -1
    rest = [s for s in top[1:]]
    by_table = {}
    for s in rest:
        by_table.setdefault(s["table"], []).append(s["name"])
    assert sorted(by_table) == list(range(12))
    for t, got in by_table.items():
        assert got == [n for n in per_table if n in got] and got[-3:] == per_table[-3:], (t, got)
        assert ("compute permutation Z(x) polys" in got) == bool(stark.tables[t].permutation_pairs), (t, got)   # prover.rs:371-377
    # times: non-negative, children inside their parents, scopes in stream order
    assert all(s["ms"] >= 0 and s["start_ms"] >= 0 and 0 <= s["sharded_ms"] <= s["ms"] + 1e-3 for s in sc)
    assert all(b["start_ms"] >= a["start_ms"] - 1e-3 for a, b in zip(sc, sc[1:]))
    total = [s for s in sc if s["name"] == "prove_with_traces total"]
    assert len(total) == 1 and total[0]["depth"] == 0
    assert all(s["start_ms"] + s["ms"] <= total[0]["start_ms"] + total[0]["ms"] + 1e-2 for s in sc)
    # a buffer that is too small is refused and the count comes back
    import ctypes as C
    from olavm_amd.backend import OlaScopeTime
    n = C.c_uint32()
    small = (OlaScopeTime * 3)()
    be.scope_times(enable=True)
    be.prove_with_traces(blob, traces, params, compress)
    assert be.lib.ola_gpu_scope_times(be.ctx, 0, small, 3, C.byref(n)) == -1 and n.value == len(sc)
    # switched off again: the next proof leaves no scopes behind
    be.prove_with_traces(blob, traces, params, compress)
    assert be.scope_times() == []
