#!/usr/bin/env python3
"""Compare the MI355X prover (and the oracle) with proofs dumped by the real Rust prover (integration/pin/pin_dump.rs).

    python integration/pin/compare_with_dump.py DIR [--no-gpu]

For every DIR/<name>.traces: prove on the GPU (ola_prove_with_traces) and with the oracle, and compare with DIR/<name>.proof.
The 12 pow_witness words are the only bytes that may differ (the reference returns *a* grinding nonce, this backend the minimal
one); both witnesses are checked against the proof-of-work condition by the oracle verifier.
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
    print(f"  {name}: {label}: {len(runs)} differing field(s) at byte offsets {[s for s, _ in runs]}"
          + (" -- at most one 8-byte field per table: the pow_witness words" if ok else " -- MORE than the pow_witness words differ"))
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
