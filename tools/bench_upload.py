"""Same-box alternation of the trace-upload paths (olavm_amd/csrc/upload.h) inside the 2^22-row proof: wall-clock of
ola_prove_with_traces / _cols, the proving thread's wait for column groups and the upload's own rate, per path and thread count.
Usage: python tools/bench_upload.py [log_n=22] [hasher=blake3] [rounds=3]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd.backend import Backend

L = int(sys.argv[1]) if len(sys.argv) > 1 else 22
hasher = sys.argv[2] if len(sys.argv) > 2 else "blake3"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
blob = T.ola_stark().blob()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=L, log_n_mem=L)
cols = [[np.array(t[c], dtype=np.uint64, copy=True) for c in range(t.shape[0])] for t in traces]
be = Backend(device=0, hasher=hasher)
want = be.prove_with_traces(blob, traces, params, compress)
be.prove_with_traces(blob, traces, params, compress)
variants = [("pageable", {"OLA_UPLOAD": "pageable"}, traces), ("staged K=2", {"OLA_UPLOAD_THREADS": "2"}, traces), ("staged K=4 (default)", {}, traces),
            ("staged K=8", {"OLA_UPLOAD_THREADS": "8"}, traces),
            ("staged 1 stream, 4 MB x 32 (first version)", {"OLA_UPLOAD_THREADS": "6", "OLA_UPLOAD_PIECE_MB": "4", "OLA_UPLOAD_SLOTS": "32", "OLA_UPLOAD_STREAMS": "1"}, traces),
            ("staged 1 stream, 8 MB", {"OLA_UPLOAD_STREAMS": "1"}, traces),
            ("staged 2 streams, 4 MB x 32", {"OLA_UPLOAD_PIECE_MB": "4", "OLA_UPLOAD_SLOTS": "32"}, traces),
            ("staged 2 streams, 16 MB x 8", {"OLA_UPLOAD_PIECE_MB": "16", "OLA_UPLOAD_SLOTS": "8"}, traces),
            ("staged 3 streams, 8 MB", {"OLA_UPLOAD_STREAMS": "3"}, traces),
            ("pageable, scattered columns", {"OLA_UPLOAD": "pageable"}, cols),
            ("staged default, scattered columns", {}, cols), ("staged default", {}, traces)]
res = {name: [] for name, _, _ in variants}
for r in range(rounds):
    for name, env, tr in variants:
        for k, v in env.items():
            os.environ[k] = v
        t0 = time.perf_counter()
        ok = be.prove_with_traces(blob, tr, params, compress) == want
        dt = time.perf_counter() - t0
        up = be.upload_stats()
        res[name].append((dt, up["waited_ms"], up["bytes"] / 1e6 / max(up["total_ms"], 1e-9), ok))
        for k in env:
            del os.environ[k]
out = {}
for name, rows in res.items():
    rows.sort()
    m = rows[len(rows) // 2]
    out[name] = {"seconds": round(m[0], 4), "min_seconds": round(rows[0][0], 4), "upload_wait_ms": round(m[1], 1), "upload_GBps": round(m[2], 1), "identical": all(x[3] for x in rows)}
    print("%-36s %.4f s (min %.4f)  wait %6.1f ms  upload %5.1f GB/s  identical %s" % (name, m[0], rows[0][0], m[1], m[2], all(x[3] for x in rows)), flush=True)
print(json.dumps({"log_n": L, "hasher": hasher, "rounds": rounds, "variants": out}))
