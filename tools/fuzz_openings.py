"""Randomised check of round 6's opening kernels against the kernels they replace: 12-table instances with random heights of the large
tables (2^12 .. 2^20: both sides of every size threshold of eval_points_wide_kernel / fold16_kernel / leaf_b3_ext16_kernel), both hash
configurations.  The same instances are proven in two child processes -- one with the new kernels (default), one with OLA_EVAL_WIDE=0
OLA_FOLD16=0 OLA_LEAF_EXT_STAGED=0 (the switches are read once per process) -- and the proofs must be the same bytes; the oracle's verifier
accepts the first of each configuration.        usage: python tools/fuzz_openings.py [iterations] [seed]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(iters, seed, verify):
    import numpy as np
    from olavm_amd.air import ola_tables as T, tracegen
    from olavm_amd.backend import Backend
    rng = np.random.default_rng(seed)
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    bes = {h: Backend(device=0, hasher=h) for h in ("poseidon", "blake3")}
    o = None
    out = []
    for it in range(iters):
        lc, lm, lp = (int(x) for x in rng.integers(12, 21, size=3))
        ln = int(rng.integers(3, 12))
        traces, params, compress = tracegen.empty_program_instance(log_n=ln, log_n_cpu=lc, log_n_mem=lm, log_n_poseidon=lp, live=rng)
        for h, be in bes.items():
            proof = be.prove_with_traces(blob, traces, params, compress)
            if verify and it == 0:
                from tests import oracle_lib
                o = o or oracle_lib.load()
                with o.hasher(h):
                    rc, why = o.verify_all_proof(blob, proof, params)
                assert rc == 0, (h, why)
            out.append({"it": it, "hasher": h, "heights": [ln, lc, lm, lp], "bytes": len(proof), "sha256": hashlib.sha256(bytes(proof)).hexdigest()})
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1")
        sys.exit(0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 606
    res = {}
    for name, env in (("new", {}), ("old", {"OLA_EVAL_WIDE": "0", "OLA_FOLD16": "0", "OLA_LEAF_EXT_STAGED": "0"})):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(iters), str(seed), "1" if name == "new" else "0"], env=e, capture_output=True, text=True, timeout=3000)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for a, b in zip(res["new"], res["old"]):
        assert a == b, ("the kernels disagree", a, b)
        print("ok", a["it"], a["hasher"], "heights", a["heights"], "bytes", a["bytes"], a["sha256"][:16], flush=True)
    print("openings fuzz passed: %d proofs identical under both kernel sets" % len(res["new"]))
