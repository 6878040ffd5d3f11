#!/usr/bin/env python3
"""`ola prove`'s call order in fresh processes (bench.py cold_process_prove): OlaStark::default() [early hook or not] -> trace
generation -> context + first proof -> second proof; both hash configurations, alternated.  With --phases the children run under
OLA_TIMING=1 and their [ola-timing] lines are kept (first proof against second, phase by phase)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 22
if "--phases" in sys.argv:
    os.environ["OLA_TIMING"] = "1"
    for hasher in (os.environ.get("OLA_HASHER", "blake3"),):
        out = subprocess.run([sys.executable, "-c", bench.COLD_CHILD % {"root": ROOT, "log_n": log_n, "device": 0, "early": 1, "hasher": hasher}],
                             capture_output=True, text=True, timeout=900)
        print(out.stdout[-600:])
        print("\n".join(l for l in out.stderr.splitlines() if l.startswith("[ola-timing]") and not l.startswith("[ola-timing]      ")))
    sys.exit(0)
for rep in range(2):
    for hasher in ("poseidon", "blake3"):
        for early in (True, False):
            time.sleep(float(os.environ.get("OLA_COLD_GAP_S", "0")))
            d = bench.cold_process_prove(log_n, 0, early, hasher)
            print(json.dumps({"hasher": hasher, **d}), flush=True)
