#!/usr/bin/env python3
"""Where a proof's time goes with respect to the coset partition (ola_gpu_proof_stats), measured on ONE GPU, and the speed-up
that projects to for 2 / 4 / 8 GPUs.  Same instances as bench.py's `prove` (padding rows, 2^log_n-row CPU and memory tables) and
`prove_real_execution`; both hash configurations.

    python tools/proof_stats.py [--log-n 22] [--real] [--devices 0 0] [--out gpurun_out/proof_stats.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

# one xGMI link, one direction (MI355X: 7 links per GPU, 153.6 GB/s per link both directions together); a pull all-gather
# over G GPUs uses G - 1 of a rank's links at once.  0.7 of the spec rate is what large peer copies reach on MI300-class parts.
XGMI_LINK_BPS = 0.7 * 76.8e9
# two thread barriers + G - 1 copy launches + event waits: measured with G ranks sharing the one GPU of the test box
# (tests/gpu_peer_gather_check.cpp, profiles/r03_peer_all_gather.txt) -- an upper bound for ranks that have a GPU each
EXCHANGE_LATENCY_S = {2: 31e-6, 4: 59e-6, 8: 158e-6}


def project(st, world):
    """T(G) from a one-GPU run: the bracketed kernel time divides by min(G, 2^k), the rest is repeated by every rank, the
    exchanges are added (a rank receives (G-1)/G of the gathered bytes over G-1 links)."""
    t = st["wall_ms"] * 1e-3
    saved = 0.0
    for k, key in ((1, "sharded_ms_upto2"), (2, "sharded_ms_upto4"), (3, "sharded_ms_upto8")):
        saved += st[key] * 1e-3 * (1.0 - 1.0 / min(world, 1 << k))
    recv = st["exchange_bytes"] * (world - 1) / world
    xchg = recv / ((world - 1) * XGMI_LINK_BPS) + st["exchanges"] * EXCHANGE_LATENCY_S[world]
    return t - saved + xchg


def summarise(st):
    sharded = st["sharded_ms_upto2"] + st["sharded_ms_upto4"] + st["sharded_ms_upto8"]
    out = {"wall_ms": round(st["wall_ms"], 2), "sharded_kernel_ms": round(sharded, 2),
           "replicated_ms": round(st["wall_ms"] - sharded, 2), "replicated_share": round(1 - sharded / st["wall_ms"], 4),
           "sharded_kernel_ms_by_max_world": {"2": round(st["sharded_ms_upto2"], 2), "4": round(st["sharded_ms_upto4"], 2), "8": round(st["sharded_ms_upto8"], 2)},
           "exchange_bytes": st["exchange_bytes"], "exchanges": st["exchanges"]}
    for g in (2, 4, 8):
        out["projected_speedup_%d" % g] = round(st["wall_ms"] * 1e-3 / project(st, g), 2)
    return out


def measure(be, blob, traces, params, compress, reps=3):
    be.proof_stats(enable=True)
    be.prove_with_traces(blob, traces, params, compress)
    best = None
    for _ in range(reps):
        be.prove_with_traces(blob, traces, params, compress)
        st = be.proof_stats()
        if best is None or st["wall_ms"] < best["wall_ms"]:
            best = st
    be.proof_stats(enable=False)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--devices", type=int, nargs="*", default=None, help="run on a multi-device context as well (entries may repeat)")
    ap.add_argument("--hashers", nargs="+", default=["poseidon", "blake3"])
    ap.add_argument("--out", default="gpurun_out/proof_stats.json")
    args = ap.parse_args()
    from olavm_amd.air import ola_tables as T, tracegen
    from olavm_amd.backend import Backend
    blob = T.ola_stark().blob()
    if args.real:
        from olavm_amd.air import fastexec, miniexec
        count = ((1 << args.log_n) - 8) // 14
        traces, params, compress = fastexec.instance(miniexec.memory_program(count), range_bits=16, limb_bits=8, max_steps=1 << (args.log_n + 1))
    else:
        traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=args.log_n, log_n_mem=args.log_n)
    res = {"heights": [int(t.shape[1]).bit_length() - 1 for t in traces], "real_execution": args.real,
           "assumptions": {"xgmi_link_bytes_per_s_one_direction": XGMI_LINK_BPS, "exchange_latency_s": {str(k): v for k, v in EXCHANGE_LATENCY_S.items()}}}
    for h in args.hashers:
        be = Backend(device=0, hasher=h)
        st = measure(be, blob, traces, params, compress)
        res[h] = summarise(st)
        print(h, json.dumps(res[h]), flush=True)
        single = be.prove_with_traces(blob, traces, params, compress)
        be.close()
        if args.devices:
            mb = Backend(devices=args.devices, hasher=h)
            mb.memory_stats(reset=True)
            mb.prove_with_traces(blob, traces, params, compress)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                p = mb.prove_with_traces(blob, traces, params, compress)
                ts.append(time.perf_counter() - t0)
            st = mb.proof_stats()
            res[h]["multi_device_context"] = {"devices": args.devices, "seconds": round(min(ts), 4), "identical_to_single_gpu_proof": p == single,
                                              "peer_exchanges": st["peer_exchanges"], "peer_bytes_moved": st["peer_bytes_moved"],
                                              # each rank's own pool (ranks may share a GPU here): the largest high-water mark over the ranks
                                              "pool_high_water_gb_largest_rank": round(mb.memory_stats()["reserved_peak"] / 1e9, 2)}
            print(h, "multi", json.dumps(res[h]["multi_device_context"]), flush=True)
            mb.close()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
