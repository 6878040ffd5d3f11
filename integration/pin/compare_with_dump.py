#!/usr/bin/env python3
"""Compare the MI355X prover (and the oracle) with proofs dumped by the real Rust prover (integration/pin/pin_dump.rs).

    python integration/pin/compare_with_dump.py DIR [--no-gpu]

For every DIR/<name>.traces: prove on the GPU (ola_prove_with_traces) and with the oracle, and compare with DIR/<name>.proof.
The 12 pow_witness words are the only bytes that may differ (the reference returns *a* grinding nonce, this backend the minimal
one); both witnesses are checked against the proof-of-work condition by the oracle verifier.  When more differs, the tool parses
both proofs (parse_all_proof: the reference wire format span by span) and names the FIRST DIFFERING PHASE -- "table 3:
permutation_ctl_zs_cap" says the trace commitment of table 3 agreed and its Z columns did not -- and, when DIR/<name>.diag is
present (the reference's own CTL challenges), checks the transcript before anything else.
"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def read_traces(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"OLAPIN01", "not a pin dump"
    off = 8
    (nt,) = struct.unpack_from("<I", raw, off); off += 4
    traces = []
    for _ in range(nt):
        ncols, log_n = struct.unpack_from("<II", raw, off); off += 8
        n = 1 << log_n
        traces.append(np.frombuffer(raw, dtype="<u8", count=ncols * n, offset=off).reshape(ncols, n).copy())
        off += ncols * n * 8
    (nc,) = struct.unpack_from("<I", raw, off); off += 4
    compress = [int(x) for x in np.frombuffer(raw, dtype="<u8", count=nc, offset=off)]
    return traces, compress


def write_traces(path, traces, compress):
    """The writer half in Python (used by the tests to exercise the reader; the real dumps come from pin_dump.rs)."""
    with open(path, "wb") as f:
        f.write(b"OLAPIN01" + struct.pack("<I", len(traces)))
        for t in traces:
            t = np.ascontiguousarray(t, dtype="<u8")
            f.write(struct.pack("<II", t.shape[0], int(t.shape[1]).bit_length() - 1))
            f.write(t.tobytes())
        f.write(struct.pack("<I", len(compress)) + b"".join(struct.pack("<Q", int(c)) for c in compress))


def parse_all_proof(raw, blob=None):
    """The spans of an AllProof in the reference wire format (circuits/src/stark/serialization.rs:349-393 write_proof /
    write_all_proof), in the order the prover produces them: [(name, start, end)], tiling `raw` exactly.  Digests are 32 bytes
    under either hash configuration.  Used to say WHICH phase of WHICH table differs first."""
    import struct as st
    spans, off = [], 0

    def u32():
        nonlocal off
        (v,) = st.unpack_from("<I", raw, off)
        off += 4
        return v

    def span(name, nbytes):
        nonlocal off
        if off + nbytes > len(raw):
            raise ValueError(f"truncated proof at {name}")
        spans.append((name, off, off + nbytes))
        off += nbytes

    def cap(name):
        n = u32()
        span(name, 32 * n)

    def ext_vec(name):
        n = u32()
        span(name, 16 * n)

    def field_vec(name):
        n = u32()
        span(name, 8 * n)

    def merkle_proof(name):
        nonlocal off
        depth = raw[off]
        off += 1
        span(name, 32 * depth)

    nt = u32()
    for t in range(nt):
        T = f"table {t}: "
        cap(T + "trace_cap")
        cap(T + "permutation_ctl_zs_cap (Z columns: challenges, running products, their commitment)")
        cap(T + "quotient_polys_cap (constraint quotient and its commitment)")
        ext_vec(T + "openings.local_values")
        ext_vec(T + "openings.next_values")
        ext_vec(T + "openings.permutation_ctl_zs")
        ext_vec(T + "openings.permutation_ctl_zs_next")
        field_vec(T + "openings.ctl_zs_last")
        ext_vec(T + "openings.quotient_polys")
        nl = u32()
        for li in range(nl):
            cap(T + f"fri.commit_phase_merkle_caps[{li}]")
        nq = u32()
        for q in range(nq):
            no = u32()
            for o in range(no):
                field_vec(T + f"fri.query[{q}].initial_trees_proof[{o}].leaf")
                merkle_proof(T + f"fri.query[{q}].initial_trees_proof[{o}].path")
            ns = u32()
            for li in range(ns):
                ar = u32()
                span(T + f"fri.query[{q}].step[{li}].evals", 16 * ar)
                merkle_proof(T + f"fri.query[{q}].step[{li}].path")
        ext_vec(T + "fri.final_poly")
        span(T + "fri.pow_witness", 8)
    nc = u32()
    span("compress_challenges", 8 * nc)
    if off != len(raw):
        raise ValueError(f"{len(raw) - off} trailing bytes after the proof")
    return spans


def first_difference(got, want):
    """Name of the first span in which two proofs differ, ignoring the pow_witness words (None if there is none)."""
    try:
        spans = parse_all_proof(want)
    except (ValueError, IndexError, Exception) as e:          # noqa: BLE001
        return f"(reference proof does not parse: {e})"
    if len(got) != len(want):
        try:
            mine = parse_all_proof(got)
        except Exception as e:                                # noqa: BLE001
            return f"(lengths differ and this prover's proof does not parse: {e})"
        for (na, a0, a1), (nb, b0, b1) in zip(mine, spans):
            if na != nb or a1 - a0 != b1 - b0:
                return f"{nb}: {b1 - b0} bytes in the reference proof, {na}: {a1 - a0} bytes here (shape differs)"
        return "(span count differs)"
    for name, a, b in spans:
        if name.endswith("pow_witness"):
            continue
        if got[a:b] != want[a:b]:
            return name
    return None


def transcript_challenges(want, hasher="poseidon"):
    """The cross-table-lookup challenges the transcript yields after observing every table's trace cap (get_challenges.rs:23-33),
    recomputed with this backend's host challenger from the REFERENCE proof's caps: [(beta, gamma)] * num_challenges."""
    from olavm_amd.backend import Challenger
    ch = Challenger(hasher=hasher)
    for name, a, b in parse_all_proof(want):
        if name.endswith("trace_cap"):
            ch.observe_cap(np.frombuffer(want[a:b], dtype="<u8").reshape(-1, 4))
    out = []
    for _ in range(2):
        beta, gamma = ch.get(), ch.get()
        out.append((beta, gamma))
    return out


def read_diag(path):
    """<name>.diag of pin_dump.rs: the reference's own CTL challenges and per-table caps, as text."""
    d = {"ctl_challenges": [], "caps": {}}
    for line in open(path):
        w = line.split()
        if not w:
            continue
        if w[0] == "ctl_challenge":
            d["ctl_challenges"].append((int(w[1]), int(w[2])))
        elif w[0] == "table":
            # pin_dump.rs writes a cap with Buffer::write_merkle_cap (serialization.rs:125-134): a u32 entry count, then the digests
            raw = bytes.fromhex(w[3])
            (n,) = struct.unpack_from("<I", raw, 0)
            if len(raw) != 4 + 32 * n:
                raise ValueError(f"{path}: cap of table {w[1]} ({w[2]}): {len(raw)} bytes do not hold a count and {n} digests")
            d["caps"][(int(w[1]), w[2])] = raw[4:]
    return d


def compare(name, got, want, label):
    if len(got) != len(want):
        print(f"  {name}: {label}: LENGTH differs ({len(got)} vs {len(want)} bytes)")
        return False
    diff = [o for o in range(0, len(got), 1) if got[o] != want[o]]
    if not diff:
        print(f"  {name}: {label}: identical, pow_witness included ({len(got)} bytes)")
        return True
    # group into 8-byte aligned words relative to the 4-byte header: fields start at odd multiples of 4 as well, so report runs
    runs, start, prev = [], diff[0], diff[0]
    for o in diff[1:]:
        if o != prev + 1:
            runs.append((start, prev)); start = o
        prev = o
    runs.append((start, prev))
    ok = len(runs) <= 12 and all(e - s < 8 for s, e in runs)
    print(f"  {name}: {label}: {len(runs)} differing field(s) at byte offsets {[s for s, _ in runs][:16]}"
          + (" -- at most one 8-byte field per table: the pow_witness words" if ok else " -- MORE than the pow_witness words differ"))
    if not ok:
        print(f"  {name}: {label}: FIRST DIFFERING PHASE: {first_difference(got, want)}")
    return ok


def main():
    d = sys.argv[1]
    use_gpu = "--no-gpu" not in sys.argv
    from olavm_amd.air import ola_tables as T
    from tests import oracle_lib
    o = oracle_lib.load()
    blob = T.ola_stark().blob()
    be = be3 = None
    if use_gpu:
        from olavm_amd.backend import Backend
        be = Backend(device=0)
    all_ok = True
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".traces"):
            continue
        name = fn[:-7]
        traces, compress = read_traces(os.path.join(d, fn))
        want = open(os.path.join(d, name + ".proof"), "rb").read()
        params = [compress[2], compress[10]]                 # the bitwise and program compress challenges are the tables' parameters
        print(name, "heights 2^%s" % [int(t.shape[1]).bit_length() - 1 for t in traces])
        rc, why = o.verify_all_proof(blob, want, params)
        print(f"  {name}: oracle verifier on the reference's proof: {'accepts' if rc == 0 else 'REJECTS: ' + str(why)}")
        all_ok &= rc == 0
        diag = os.path.join(d, name + ".diag")
        if os.path.exists(diag):            # the reference's own challenges: a transcript mismatch shows here before anything else
            dg = read_diag(diag)
            mine = transcript_challenges(want)
            same = dg["ctl_challenges"] == mine
            verdict = "equal to the reference's own" if same else "DIFFER: " + str(mine) + " vs " + str(dg["ctl_challenges"])
            print(f"  {name}: CTL challenges from the reference's trace caps through this backend's challenger: {verdict}")
            all_ok &= same
        if max(t.shape[1] for t in traces) <= (1 << 14):
            all_ok &= compare(name, o.prove_with_traces(blob, traces, params, compress), want, "oracle prover vs reference")
        if be is not None:
            got = be.prove_with_traces(blob, traces, params, compress)
            all_ok &= compare(name, got, want, "GPU prover vs reference")
            rc, why = o.verify_all_proof(blob, got, params)
            all_ok &= rc == 0
        # the same traces under Blake3GoldilocksConfig (<name>.blake3.proof, written by pin_dump.rs next to the Poseidon proof)
        b3 = os.path.join(d, name + ".blake3.proof")
        if os.path.exists(b3):
            want3 = open(b3, "rb").read()
            with o.hasher("blake3"):
                rc, why = o.verify_all_proof(blob, want3, params)
                print(f"  {name}: oracle verifier (Blake3 configuration) on the reference's proof: {'accepts' if rc == 0 else 'REJECTS: ' + str(why)}")
                all_ok &= rc == 0
                if max(t.shape[1] for t in traces) <= (1 << 14):
                    all_ok &= compare(name, o.prove_with_traces(blob, traces, params, compress), want3, "oracle prover vs reference, Blake3")
            if use_gpu:
                if be3 is None:
                    be3 = Backend(device=0, hasher="blake3")
                all_ok &= compare(name, be3.prove_with_traces(blob, traces, params, compress), want3, "GPU prover vs reference, Blake3")
    print("ALL OK" if all_ok else "DIFFERENCES FOUND")
    return 0 if all_ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
