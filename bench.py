#!/usr/bin/env python3
"""bench.py -- headline measurement for the MI355X Goldilocks proving backend.

Step = one pass of the hot path over one batch of synthetic input already resident in HBM:
a batched forward NTT (the reference's cfft::evaluate_poly, natural order in and out) over the columns of a
2^22-row trace (94 columns = the width of OlaVM's CPU table), BASELINE.json configs[1].

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU; columns are independent, so ranks shard them with no data-path collective (weak scaling: every
rank transforms its own `--cols` columns).  Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(log_n, seconds_budget=20.0):
    """The CPU oracle's evaluate_poly (a port of the reference's cfft, NOT the reference itself: no Rust toolchain here)
    timed on the host cores over a bounded sample of the same workload."""
    from tests import oracle_lib
    o = oracle_lib.load()
    lib = o.lib
    lib.oracle_num_threads.restype = C.c_int
    lib.oracle_evaluate_poly_batch.argtypes = [oracle_lib.U64P, C.c_size_t, C.c_size_t]
    cores = int(lib.oracle_num_threads())
    n = 1 << log_n
    rng = np.random.default_rng(1)
    # calibrate on one column per core, then size the sample to the budget
    d = oracle_lib.rand_field(rng, (cores, n))
    t0 = time.perf_counter()
    lib.oracle_evaluate_poly_batch(oracle_lib.ptr(d), n, cores)
    t1 = time.perf_counter() - t0
    rounds = max(1, min(8, int(seconds_budget / max(t1, 1e-3)) - 1))
    cols = cores * rounds
    d = oracle_lib.rand_field(rng, (cols, n))
    t0 = time.perf_counter()
    lib.oracle_evaluate_poly_batch(oracle_lib.ptr(d), n, cols)
    dt = time.perf_counter() - t0
    gbps = 16.0 * n * cols / dt / 1e9
    return {"value": round(gbps, 4), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{cols} columns x 2^{log_n} forward NTT (oracle evaluate_poly, OpenMP over columns) in {dt:.2f}s"}


def cpu_baseline_prove(be, log_n=14):
    """The CPU prove-time leg of the baseline: the oracle's prover (a port of circuits::stark::prover::prove_with_traces, NOT the
    reference itself -- no Rust toolchain here) on the host cores, bounded to the 12-table instance with every large table at
    2^log_n rows, next to the GPU proving the identical instance; the bytes are compared.  log_n = 17 (about 45 s of CPU work on
    the 16-core test box) is out of the GPU's launch-bound regime; 14 is the miniature of rounds 4 - 5, kept as `prove_small`.
    The oracle is timed here as the baseline and used as the checker, nothing it computes is shipped."""
    import numpy as np
    from olavm_amd.air import ola_tables as T
    from tests import oracle_lib, tracegen
    o = oracle_lib.load()
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=log_n, live=np.random.default_rng(12))
    t0 = time.perf_counter()
    want = o.prove_with_traces(blob, traces, params, compress)
    cpu_s = time.perf_counter() - t0
    be.prove_with_traces(blob, traces, params, compress)
    t0 = time.perf_counter()
    got = be.prove_with_traces(blob, traces, params, compress)
    gpu_s = time.perf_counter() - t0
    return {"log_n": log_n, "cpu_seconds": round(cpu_s, 2), "gpu_seconds": round(gpu_s, 4), "ratio": round(cpu_s / max(gpu_s, 1e-9), 1), "kind": "port",
            "identical_bytes": bool(got == want), "proof_bytes": len(got),
            "sample": f"prove_with_traces, 12 tables, heights 2^{[int(t.shape[1]).bit_length() - 1 for t in traces]} (miniature fixed tables), "
                      "oracle prover (OpenMP) on the host cores vs one MI355X, same traces"}


def pmc_record(log_n, cols):
    """HBM bytes per launch and VALU instructions per element of the NTT pass kernels from the rocprofv3 PMC record
    profiles/r*_ntt_pmc.json (tools/pmc_ntt.sh: FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE + SQ_INSTS_VALU, separate
    passes over the same 94 x 2^22 transform).  bench.py cannot run the profiler on itself; the record carries the hash of the
    kernel sources it was taken from and is used only while those sources are unchanged -- otherwise the fields are null."""
    import glob
    import hashlib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ntt_pmc.json")), reverse=True):      # the newest record whose sources still match
        try:
            d = json.load(open(path))
            h = hashlib.sha256()
            for f in d["sources"]:
                h.update(open(os.path.join(ROOT, f), "rb").read())
            if h.hexdigest()[:16] != d["source_sha16"] or (log_n, cols) != (d["shape"]["log_n"], d["shape"]["columns"]):
                continue
            r = d["ntt_94x2^22"]
            return {"traffic": r["traffic_bytes_per_launch"], "valu_insts_per_element": r["valu_insts_per_element"],
                    "source": f"profiles/{os.path.basename(path)} ({d['timestamp']}, kernel sources {d['source_sha16']} unchanged)"}
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def valu_roofline(rec, log_n, cols, ms_per_transform):
    """The ceiling that binds the NTT (DESIGN.md "NTT roofline accounting"): VALU issue slots.  Instructions per element from
    the PMC record above, against the guide's issue rate on 256 CUs x 4 SIMDs at 2.4 GHz (one wave64 instruction per
    SIMD every 2 cycles)."""
    if not rec or not rec.get("valu_insts_per_element"):
        return None
    ipe = rec["valu_insts_per_element"]
    wave_insts = ipe * cols * (1 << log_n) / 64.0
    achieved = wave_insts / (ms_per_transform * 1e-3)
    return {"insts_per_element": round(ipe, 1), "achieved": round(achieved / 1e9, 1), "unit": "G wave-instructions/s",
            "peak_2_cycle_issue": round(VALU_PEAK_GUIDE / 1e9, 1), "frac_of_2_cycle_issue_peak": round(achieved / VALU_PEAK_GUIDE, 3),
            "note": "the measured issue cost on this chip is 2.6 cycles for plain 32-bit moves / adds / logic and 4.3 - 4.8 for 64-bit adds, carries, "
                    "selects and multiply-adds (tools/ubench/valu_rates.hip); a kernel mixes both, so only the 2-cycle figure is a ceiling",
            "source": rec["source"]}


VALU_PEAK_GUIDE = 256 * 4 * 2.4e9 / 2.0        # wave64 instructions/s: one per SIMD every 2 cycles (MI355X_MICROARCH.md, SIMD-32)
POSEIDON_VALU_PER_PERMUTATION = 15.5e3           # SQ_INSTS_VALU per permutation, profiles/r05_ntt_pmc.json (poseidon.cuh unchanged since)


def partition_projection(st):
    """From ONE GPU's accounting (ola_gpu_proof_stats) to the coset partition over G GPUs: the bracketed kernel time divides by
    min(G, 2^k), everything else is repeated by every rank, the exchanges are added -- a rank receives (G-1)/G of the gathered
    bytes over G-1 xGMI links at once (0.7 x 76.8 GB/s per link and direction assumed); per exchange the latency the library's
    all-gather showed on the one-GPU box with G ranks sharing the device (tests/gpu_peer_gather_check.cpp,
    profiles/r03_peer_all_gather.txt: 31 / 59 / 158 us for 2 / 4 / 8 ranks -- an upper bound for ranks with a GPU each)."""
    link, lat_by_g = 0.7 * 76.8e9, {2: 31e-6, 4: 59e-6, 8: 158e-6}
    sharded = st["sharded_ms_upto2"] + st["sharded_ms_upto4"] + st["sharded_ms_upto8"]
    out = {"wall_ms": round(st["wall_ms"], 2), "sharded_kernel_ms": round(sharded, 2), "replicated_ms": round(st["wall_ms"] - sharded, 2),
           "replicated_share": round(1.0 - sharded / max(st["wall_ms"], 1e-9), 4), "exchange_bytes": st["exchange_bytes"],
           "exchanges": st["exchanges"], "assumed_xgmi_link_GBps_one_direction": round(link / 1e9, 1), "assumed_exchange_latency_us": {"2": 31, "4": 59, "8": 158}}
    for g in (2, 4, 8):
        t = st["wall_ms"] * 1e-3
        for k, key in ((1, "sharded_ms_upto2"), (2, "sharded_ms_upto4"), (3, "sharded_ms_upto8")):
            t -= st[key] * 1e-3 * (1.0 - 1.0 / min(g, 1 << k))
        t += st["exchange_bytes"] * (g - 1) / g / ((g - 1) * link) + st["exchanges"] * lat_by_g[g]
        out["projected_speedup_%d" % g] = round(st["wall_ms"] * 1e-3 / t, 2)
    return out


def phase_block(be, hasher="poseidon"):
    """SURVEY 8(d) config 4 figures of the proof that has just run (accounting on): Merkle leaves/s and permutations/s with both
    rooflines, FRI-fold GB/s, and the transforms, from HIP events around the kernel families inside the library."""
    ph = be.phase_stats()
    out = {}
    ms, calls, byts = ph["leaf_hash"]
    nms, nodes, nbytes = ph["merkle_levels"]
    if ms > 0:
        perms = calls + nodes
        t = (ms + nms) * 1e-3
        m = {"leaf_hash_ms": round(ms, 2), "merkle_levels_ms": round(nms, 2), "levels_over_leaves": round(nms / ms, 3),
             "hash_invocations": int(perms), "invocations_per_s": round(perms / t / 1e9, 3), "invocations_unit": "G/s",
             "invocation": "Poseidon permutation" if hasher == "poseidon" else "Blake3 compression",
             "hbm": {"bytes": int(byts + nbytes), "achieved_GBps": round((byts + nbytes) / t / 1e9, 1), "frac": round((byts + nbytes) / t / (HBM_PEAK_GBPS * 1e9), 4)}}
        if hasher == "poseidon":
            wi = perms * POSEIDON_VALU_PER_PERMUTATION / 64.0 / t
            m["valu"] = {"wave_insts_per_s_G": round(wi / 1e9, 1), "frac_of_2_cycle_issue_peak": round(wi / VALU_PEAK_GUIDE, 3),
                         "insts_per_permutation": POSEIDON_VALU_PER_PERMUTATION}
        out["merkle"] = m
    top = getattr(be, "phase_top", {})

    def dominant(name):
        """the phase's scope that moved the most bytes: what one large launch reaches, without the launch-bound small ones"""
        tms, tb = top.get(name, (0, 0))
        return {"ms": round(tms, 4), "bytes": int(tb), "GBps": round(tb / (tms * 1e-3) / 1e9, 1), "frac": round(tb / (tms * 1e-3) / (HBM_PEAK_GBPS * 1e9), 4)} if tms > 0 else None
    ms, byts, elems = ph["fri_fold"]
    if ms > 0:
        out["fri_fold"] = {"ms": round(ms, 3), "GBps": round(byts / (ms * 1e-3) / 1e9, 1), "extension_elements": int(elems),
                           "frac_of_hbm_peak": round(byts / (ms * 1e-3) / (HBM_PEAK_GBPS * 1e9), 4), "dominant_launch": dominant("fri_fold")}
    ms, byts, cc = ph["lde"]
    if ms > 0:
        out["lde"] = {"ms": round(ms, 2), "GBps": round(byts / (ms * 1e-3) / 1e9, 1), "column_cosets": int(cc)}
    ms, byts, cc = ph["intt"]
    if ms > 0:
        out["intt"] = {"ms": round(ms, 2), "GBps": round(byts / (ms * 1e-3) / 1e9, 1), "columns": int(cc)}
    ms, pts, byts = ph["quotient"]
    if ms > 0:      # quotient row streaming is HBM-class -- bytes = the LDE cells a point reads (local and next row of the trace and Z batches) + 16 written
        out["quotient"] = {"ms": round(ms, 2), "points_per_s_G": round(pts / (ms * 1e-3) / 1e9, 3), "bytes": int(byts),
                           "GBps": round(byts / (ms * 1e-3) / 1e9, 1), "frac": round(byts / (ms * 1e-3) / (HBM_PEAK_GBPS * 1e9), 4), "bound": "hbm (contract); measured: re-loads of trace cells on top of 36 k VALU instructions per point (48 k before "
                                    "round 6's limb-product sums: profiles/r06_quotient_limb_sums.txt) -- the CPU table's kernel issues 611 cell loads per point for 346 distinct cells "
                                    "(profiles/r05_quotient_belady_order.txt; per-kernel HBM bytes in profiles/r06_proof_pmc_blake3.txt)"}
    ms, prods, byts = ph["open_eval"]
    if ms > 0:      # every coefficient of every committed polynomial read once per point pair
        out["open_eval"] = {"ms": round(ms, 2), "coefficient_point_products_per_s_G": round(prods / (ms * 1e-3) / 1e9, 2), "bytes": int(byts),
                            "GBps": round(byts / (ms * 1e-3) / 1e9, 1), "frac": round(byts / (ms * 1e-3) / (HBM_PEAK_GBPS * 1e9), 4), "dominant_launch": dominant("open_eval"),
                            "bound": "hbm; round 6: 25 VALU per coefficient (22-bit limb products, multiplier in SGPRs; 124 before), profiles/r06_openings_ab.txt"}
    return out


def replicated_breakdown(st, scopes, upload, heights, min_log_n=12):
    """What every rank of the coset partition repeats (`replicated_ms` = wall - bracketed kernel time), split by where it sits:
    the reference's `timed!` scopes with device times (ola_gpu_scope_times), each minus the divided work that began inside it.
    Scopes of tables below the partition threshold are summed into one figure; the trace-commitment line of the large tables
    contains the GPU's idle time while it waits for column groups (reported next to it from the host's clock)."""
    large = {t for t, h in enumerate(heights) if h >= min_log_n}
    out, small = {}, 0.0
    names = {"compute permutation Z(x) polys": "z_columns", "compute CTL Z(x) polys": "z_columns", "compute Zs commitment": "zs_commitment_interpolation",
             "compute quotient polys": "quotient_interpolation_and_split", "split quotient polys": "quotient_interpolation_and_split",
             "compute quotient commitment": "quotient_commitment", "compute openings proof": "openings_composition_fri_tail_pow_queries"}
    total = 0.0
    for s in scopes:
        if s["name"] == "prove_with_traces total":
            total = s["ms"]
        rep = s["ms"] - s["sharded_ms"]
        if s["name"].endswith("trace commitment (upload overlapped)") or s["name"].endswith("prove_single_table"):
            if s["table"] not in large:
                small += rep
            elif s["name"].endswith("trace commitment (upload overlapped)"):
                out["trace_commitments_interpolation_and_upload_stalls"] = out.get("trace_commitments_interpolation_and_upload_stalls", 0.0) + rep
        elif s["table"] in large and s["depth"] == 2 and s["name"] in names:
            out[names[s["name"]]] = out.get(names[s["name"]], 0.0) + rep
    out["tables_below_the_partition_threshold"] = small
    out = {k: round(v, 2) for k, v in out.items()}
    out["host_side_and_launch_gaps"] = round(st["wall_ms"] - total, 2)
    out["upload_wait_of_the_proving_thread_host_clock"] = round(upload["waited_ms"], 2)
    out["sum_check"] = {"replicated_ms": round(st["wall_ms"] - (st["sharded_ms_upto2"] + st["sharded_ms_upto4"] + st["sharded_ms_upto8"]), 2),
                        "sum_of_lines": round(sum(v for k, v in out.items() if k != "upload_wait_of_the_proving_thread_host_clock" and not isinstance(v, dict)), 2)}
    return out


def accounted_proof(be, blob, traces, params, compress, hasher="poseidon"):
    """One more proof with the library's accounting on: the partition projection, the breakdown of the replicated share by
    `timed!` scope, and the per-kernel-family block."""
    be.proof_stats(enable=True)
    be.scope_times(enable=True)
    try:
        be.prove_with_traces(blob, traces, params, compress)
        scopes = be.scope_times()
    finally:
        be.scope_times(enable=False)
        st = be.proof_stats(enable=False)
    heights = [int(t.shape[1]).bit_length() - 1 for t in traces]
    part = partition_projection(st)
    part["replicated_breakdown_ms"] = replicated_breakdown(st, scopes, be.upload_stats(), heights)
    return {"partition": part, "kernels": phase_block(be, hasher)}


def scattered_columns(traces):
    """The traces as the reference's caller holds them (prover.rs:79-83: per table a Vec of PolynomialValues, every column its own
    Vec<F>): one allocation per column, every word written."""
    import numpy as np
    return [[np.array(t[c], dtype=np.uint64, copy=True) for c in range(t.shape[0])] for t in traces]


def verify_proofs(blob, proofs, params, hasher="poseidon"):
    """Every proof that was timed is checked AFTER the timed region: all runs must have produced the same bytes (the proof is
    a function of the traces; the PoW witness is the minimal one) and the oracle's restatement of the reference verifier
    (verifier.rs:35-206, incl. the cross-table products) must accept them.  The oracle is the checker here, nothing it
    computes is measured."""
    from tests import oracle_lib
    o = oracle_lib.load()
    same = all(p == proofs[0] for p in proofs[1:])
    t0 = time.perf_counter()
    with o.hasher(hasher):
        rc, why = o.verify_all_proof(blob, proofs[-1], params)
    return {"verified": bool(rc == 0 and same), "all_runs_identical": bool(same), "verifier": "oracle verify_all_proof",
            "verifier_seconds": round(time.perf_counter() - t0, 2), **({} if rc == 0 else {"verifier_error": str(why)[:200]})}


COLD_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy, torch                                       # not part of what a Rust host pays
from olavm_amd.air import ola_tables as T, tracegen
from olavm_amd import backend as B
blob = T.ola_stark().blob()
heights = [%(log_n)d, %(log_n)d, 18, 1, 16, 10, 10, 10, 10, 10, 10, 10]
t0 = time.perf_counter()                                  # `ola prove` has read its input: OlaStark::default() is next (client/src/main.rs:193)
B.load_library()
if %(early)d:
    B.warmup(%(device)d, airset=blob)                     # the patched ola_stark.rs:47 -> hip_prover::init_early(); returns at once
t_tr = time.perf_counter()
traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=%(log_n)d, log_n_mem=%(log_n)d)
assert [int(t.shape[1]).bit_length() - 1 for t in traces] == heights, [int(t.shape[1]).bit_length() - 1 for t in traces]
t1 = time.perf_counter()                                  # generate_traces is done: prove_with_traces is called (prover.rs:79)
be = B.Backend(device=%(device)d, hasher=%(hasher)r)      # hip_prover.rs with_ctx: waits for the warm-up thread, or pays the start-up here
t2 = time.perf_counter()
p1 = be.prove_with_traces(blob, traces, params, compress)
t3 = time.perf_counter()
up1 = be.upload_stats()
p2 = be.prove_with_traces(blob, traces, params, compress)
t4 = time.perf_counter()
up2 = be.upload_stats()
warm_ms = B.warmup_wait() if %(early)d else None
print(json.dumps({"early_hook": bool(%(early)d), "start_to_first_proof_seconds": round(t3 - t0, 4), "trace_generation_seconds": round(t1 - t_tr, 3),
                  "init_seconds_seen_by_the_prover": round(t2 - t1, 4), "first_proof_seconds": round(t3 - t2, 4), "second_proof_seconds": round(t4 - t3, 4),
                  "warmup_thread_ms": warm_ms, "upload_wait_ms": [round(up1["waited_ms"], 1), round(up2["waited_ms"], 1)],
                  "upload_ms": [round(up1["total_ms"], 1), round(up2["total_ms"], 1)], "identical": p1 == p2}))
"""


def cold_process_prove(log_n, device, early, hasher="poseidon"):
    """What `ola prove` sees (client/src/main.rs:174-214 proves once per process), in the reference's own order: OlaStark::default()
    -- where the reference calls init_gpu() (ola_stark.rs:47) and the patch calls hip_prover::init_early() -> ola_gpu_warmup -- then
    generate_traces on the host, then prove_with_traces: context creation and the FIRST proof, next to a second one.  early=False is
    the boundary without the hook: the HIP runtime, the device and the code objects come up inside the first prove_with_traces.
    `excess_over_warm` = (start -> first proof done, minus the host's trace generation) / a warm proof."""
    import subprocess
    # VRAM that a process has just freed is handed to the next one dirty and is scrubbed inside its hipMalloc (30 ms per GB) until
    # the driver has cleaned it in the background -- a few seconds (profiles/r06_cold_start.txt: children run back to back show
    # first proofs of 0.3 - 0.8 s at random, with a 4 s gap 0.237 - 0.243 s).  The gap keeps one child's dirt out of the next one's clock.
    time.sleep(float(os.environ.get("OLA_COLD_GAP_S", "4")))
    try:
        out = subprocess.run([sys.executable, "-c", COLD_CHILD % {"root": ROOT, "log_n": log_n, "device": device, "early": 1 if early else 0, "hasher": hasher}],
                             capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            return {"error": (out.stderr or out.stdout)[-300:]}
        d = json.loads(line[-1])
        d["cold_over_warm"] = round(d["first_proof_seconds"] / max(d["second_proof_seconds"], 1e-9), 3)
        d["excess_over_warm"] = round((d["start_to_first_proof_seconds"] - d["trace_generation_seconds"]) / max(d["second_proof_seconds"], 1e-9), 3)
        return d
    except Exception as e:          # noqa: BLE001 -- an extra: never at the price of the headline line
        return {"error": repr(e)[:200]}


def prove_time_2p24(be):
    """BASELINE config 5's single-GPU point: 2^24-row CPU and memory tables.  All LDEs resident would need about 350 GB, so the
    prover streams the large tables coset by coset (memory-lean mode, chosen automatically); one timed proof, verified."""
    from olavm_amd.air import ola_tables as T
    from olavm_amd.air import tracegen
    blob = T.ola_stark().blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=24, log_n_mem=24)
    be.trim()
    be.memory_stats(reset=True)
    be.prove_with_traces(blob, traces, params, compress)          # first call: allocations
    t0 = time.perf_counter()
    proof = be.prove_with_traces(blob, traces, params, compress)
    dt = time.perf_counter() - t0
    st = be.memory_stats()
    res = {"seconds": round(dt, 3), "proof_bytes": len(proof), "device_pool_high_water_gb": round(st["reserved_peak"] / 1e9, 1),
           "mode": "memory-lean (LDEs streamed coset by coset)", **verify_proofs(blob, [proof], params),
           "workload": "prove_with_traces, 12 tables, CPU and memory tables 2^24 rows, one MI355X"}
    be.trim()
    return res


def timed_proofs(be, blob, traces, params, compress, reps, hasher="poseidon"):
    times, proofs = [], []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        proofs.append(be.prove_with_traces(blob, traces, params, compress))
        times.append(time.perf_counter() - t0)
    first = times[0]
    times = sorted(times[1:])
    res = {"seconds": round(times[len(times) // 2], 4), "min_seconds": round(times[0], 4), "reps": reps, "proof_bytes": len(proofs[-1]),
           "first_call_seconds": round(first, 4), **verify_proofs(blob, proofs, params, hasher)}
    try:
        up = be.upload_stats()
        res["upload"] = {"path": up["mode"], "copier_threads": up["threads"], "trace_GB": round(up["bytes"] / 1e9, 2), "over_the_link_GB": round(up["link_bytes"] / 1e9, 2),
                         "ms": round(up["total_ms"], 1), "trace_GBps": round(up["bytes"] / 1e6 / max(up["total_ms"], 1e-9), 1),
                         "link_GBps": round(up["link_bytes"] / 1e6 / max(up["total_ms"], 1e-9), 1), "upload_wait_ms": round(up["waited_ms"], 1),
                         "first_group_ms": round(up["first_group_ms"], 2),
                         "note": "columns whose words are all below 2^32 cross the link as 32-bit words (olavm_amd/csrc/upload.h); every_column_64_bit = the same proof with that switched off"}
        # what a trace of field-sized values in every column would cost: the same proof with the narrow-column path switched off
        os.environ["OLA_UPLOAD_PACK"] = "0"
        try:
            ts, same = [], True
            for _ in range(reps):
                t0 = time.perf_counter()
                same &= be.prove_with_traces(blob, traces, params, compress) == proofs[-1]
                ts.append(time.perf_counter() - t0)
            up0 = be.upload_stats()
            ts.sort()
            res["upload"]["every_column_64_bit"] = {"seconds": round(ts[len(ts) // 2], 4), "identical": bool(same), "upload_wait_ms": round(up0["waited_ms"], 1),
                                                    "link_GBps": round(up0["link_bytes"] / 1e6 / max(up0["total_ms"], 1e-9), 1)}
        finally:
            del os.environ["OLA_UPLOAD_PACK"]
        # the same proof from the reference's own trace type: every column a separate allocation (ola_prove_with_traces_cols)
        cols = scattered_columns(traces)
        ts, same = [], True
        for _ in range(reps):
            t0 = time.perf_counter()
            same &= be.prove_with_traces(blob, cols, params, compress) == proofs[-1]
            ts.append(time.perf_counter() - t0)
        up = be.upload_stats()
        ts.sort()
        res["host_columns_scattered"] = {"seconds": round(ts[len(ts) // 2], 4), "min_seconds": round(ts[0], 4), "identical_to_contiguous_table_proof": bool(same),
                                         "over_contiguous": round(ts[len(ts) // 2] / max(res["seconds"], 1e-9), 3), "upload_wait_ms": round(up["waited_ms"], 1),
                                         "upload_GBps": round(up["bytes"] / 1e6 / max(up["total_ms"], 1e-9), 1), "columns": sum(len(c) for c in cols),
                                         "entry_point": "ola_prove_with_traces_cols"}
        del cols
    except Exception as e:              # noqa: BLE001 -- an extra
        res["host_columns_scattered"] = {"error": repr(e)[:200]}
    try:
        res.update(accounted_proof(be, blob, traces, params, compress, hasher))
    except Exception as e:              # noqa: BLE001 -- an extra
        res["partition"] = {"error": repr(e)[:200]}
    return res


def blake3_config(be_b3, blob, traces, params, compress, reps):
    """The same traces under the reference's Blake3GoldilocksConfig (plonk/config.rs:153-161: BLAKE3 Merkle trees and challenger,
    Poseidon proof of work) -- the configuration of the reference's README numbers and of its own full-prove tests -- on a
    second context created with hasher = OLA_HASH_BLAKE3; verified by the oracle verifier switched to the same configuration."""
    if be_b3 is None:
        return {}
    try:
        r = timed_proofs(be_b3, blob, traces, params, compress, reps, hasher="blake3")
        r["config"] = "Blake3GoldilocksConfig, field elements hashed as canonical words"
    except Exception as e:              # an extra: never at the price of the headline line
        r = {"error": repr(e)[:200]}
    be_b3.trim()
    return {"blake3_config": r}


def prove_time(be, log_n, reps=3, be_b3=None):
    """Second half of BASELINE.json's metric: wall-clock of the whole multi-table proof (ola_prove_with_traces, host
    traces in, AllProof bytes out -- so H2D of the traces is inside the timed region) for the 12-table OlaStark with a
    2^log_n-row CPU and memory trace.  The traces are an empty-program execution (padding rows, olavm_amd/air/tracegen.py: the
    image has no Rust executor to produce a program trace); prover work does not depend on cell values."""
    from olavm_amd.air import ola_tables as T
    from olavm_amd.air import tracegen
    blob = T.ola_stark().blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=log_n, log_n_mem=log_n)
    res = {**timed_proofs(be, blob, traces, params, compress, reps),
           "workload": f"prove_with_traces, 12 tables, heights 2^{[int(t.shape[1]).bit_length() - 1 for t in traces]}, "
                       "Poseidon config, rate_bits 3, 28 queries, 16 PoW bits; host traces in, proof bytes out"}
    try:        # the same call with the tables already resident in HBM (no PCIe upload inside the timed region)
        import numpy as np
        import torch
        want = be.prove_with_traces(blob, traces, params, compress)
        dev = [torch.from_numpy(np.ascontiguousarray(t).view(np.int64)).cuda() for t in traces]
        torch.cuda.synchronize()
        ts, same = [], True
        for _ in range(reps):
            t0 = time.perf_counter()
            same &= be.prove_with_traces(blob, dev, params, compress) == want
            ts.append(time.perf_counter() - t0)
        res["tables_resident_in_hbm"] = {"seconds": round(sorted(ts)[len(ts) // 2], 4), "identical_to_host_table_proof": bool(same)}
        del dev
        torch.cuda.empty_cache()
    except Exception as e:      # an extra: never at the price of the headline line
        res["tables_resident_in_hbm"] = {"error": repr(e)[:200]}
    res.update(blake3_config(be_b3, blob, traces, params, compress, reps))
    return res


def prove_config4(be, log_n):
    """BASELINE config 4: the Poseidon-builtin-heavy shape -- a 2^log_n-row Poseidon table (134 columns, every leaf 17
    permutations) next to 2^log_n-row CPU and memory tables -- one timed proof, verified, with the Merkle / FRI figures of that
    proof measured inside the library (HIP events around the kernel families)."""
    from olavm_amd.air import ola_tables as T
    from olavm_amd.air import tracegen
    blob = T.ola_stark().blob()
    t0 = time.perf_counter()
    traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=log_n, log_n_mem=log_n, log_n_poseidon=log_n)
    gen_s = time.perf_counter() - t0
    be.prove_with_traces(blob, traces, params, compress)
    t0 = time.perf_counter()
    proof = be.prove_with_traces(blob, traces, params, compress)
    dt = time.perf_counter() - t0
    res = {"seconds": round(dt, 4), "proof_bytes": len(proof), "trace_generation_seconds": round(gen_s, 2),
           "workload": f"prove_with_traces, 12 tables, heights 2^{[int(t.shape[1]).bit_length() - 1 for t in traces]} (Poseidon table 2^{log_n} rows), Poseidon config",
           **verify_proofs(blob, [proof], params)}
    res.update(accounted_proof(be, blob, traces, params, compress)["kernels"])
    be.trim()
    return res


def lde_roofline(be, torch, stream, log_n, cols, reps=3):
    """The transform the prover actually runs (fri/oracle.rs:66-99): coefficients -> x8 coset LDE in commitment-leaf order, `cols`
    columns of 2^log_n, operands resident; algorithmic bytes 72*n*B (read n, write 8n per column), HIP-event timed."""
    from olavm_amd.backend import OLA_NTT_COSET_LDE_LEAF_ORDER
    n = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(0x1DE)
    coef = torch.randint(0, 2**63 - 1, (cols, n), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty((cols, 8 * n), dtype=torch.int64, device="cuda")

    def run():
        be.ntt_dev(OLA_NTT_COSET_LDE_LEAF_ORDER, coef.data_ptr(), out.data_ptr(), log_n, cols, shift=7, blowup_log=3)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = 72.0 * n * cols
    passes = 1 if log_n <= 13 else (2 if log_n <= 16 else 3)
    moved = (8.0 * n * cols + 8.0 * 8 * n * cols) + (passes - 1) * 16.0 * 8 * n * cols     # what the 8 x `passes` launches stream
    del coef, out
    torch.cuda.empty_cache()
    return {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "ms": round(ms, 3), "algorithmic_bytes": alg,
            "bytes_streamed_by_the_passes": moved, "streamed_GBps": round(moved / (ms * 1e-3) / 1e9, 1),
            "kernel": "ntt2t_pass_kernel (two strided passes + bit-reversed closing pass per coset; the pre-scale rides on the load multipliers)",
            "workload": f"coset LDE x8 in leaf order (PolynomialBatch::from_coeffs' transform), {cols} columns x 2^{log_n} coefficients"}


def prove_time_real(be, log_n, reps=3, be_b3=None):
    """The same call on the traces of a REAL execution: the native trace generator (include/ola_tracegen.h, the f-1 row of
    SURVEY 8) runs the executor's memory program -- a store loop and a load / add / store / load loop -- long enough to fill a
    2^log_n-row CPU table against the full-size fixed tables; memory, range-check and program tables grow with it (the program
    table holds every fetched word: 2^(log_n+1) rows)."""
    from olavm_amd.air import fastexec, miniexec, ola_tables as T
    blob = T.ola_stark().blob()
    count = ((1 << log_n) - 8) // 14
    t0 = time.perf_counter()
    traces, params, compress = fastexec.instance(miniexec.memory_program(count), range_bits=16, limb_bits=8, max_steps=1 << (log_n + 1))
    gen_s = time.perf_counter() - t0
    return {**timed_proofs(be, blob, traces, params, compress, reps),
            "trace_generation_seconds": round(gen_s, 2),
            "workload": f"prove_with_traces on an executed program ({14 * count + 5} CPU rows, {4 * count} memory accesses), 12 tables, "
                        f"heights 2^{[int(t.shape[1]).bit_length() - 1 for t in traces]}; host traces in, proof bytes out",
            **blake3_config(be_b3, blob, traces, params, compress, reps)}


def readme_fibo_loop_blake3(be_b3, reps=3):
    """The workload of the reference's README table (README.md:69, circuits/benches/fibo_loop.rs:26,46: the Fibonacci loop with
    calldata [47, 1000], Blake3GoldilocksConfig, 866 115 executed instructions, a 2^20-row CPU table; 39.767 s on a 64-cpu Linux
    box).  Here: the same loop from the native trace generator (include/ola_tracegen.h; the image has no Rust executor), repeated
    until the CPU trace has the README run's row count -- this executor spends 6 instructions per inner step where the reference's
    compiled program spends about 18, hence 3000 repetitions for 866 k rows -- proven under the same hash configuration, verified.
    `vs_published` divides the README's seconds by ours: a different box, a different executor and the same prover shape; the line's
    top-level vs_baseline stays null because BASELINE.md publishes nothing for the headline metric itself."""
    from olavm_amd.air import fastexec, miniexec, ola_tables as T
    blob = T.ola_stark().blob()
    t0 = time.perf_counter()
    traces, params, compress = fastexec.instance(miniexec.fibonacci_loop(47, 3000), range_bits=16, limb_bits=8, max_steps=1 << 21)
    gen_s = time.perf_counter() - t0
    cpu_rows = 1 + 3000 * (3 + 47 * 6 + 3) + 1
    r = timed_proofs(be_b3, blob, traces, params, compress, reps, hasher="blake3")
    pub = 39.767
    r.update({"trace_generation_seconds": round(gen_s, 2), "executed_cpu_rows": cpu_rows,
              "workload": f"Fibonacci(47) loop x 3000 ({cpu_rows} executed CPU rows), 12 tables, heights 2^{[int(t.shape[1]).bit_length() - 1 for t in traces]}, "
                          "Blake3GoldilocksConfig; host traces in, proof bytes out",
              "published_seconds": pub, "published_on": "README.md:69, Linux 64-cpu 128 GB, Rust executor (866 115 instructions), circuits/benches/fibo_loop.rs",
              "vs_published": round(pub / r["seconds"], 1),
              "vs_published_note": "published, 64-cpu Linux, different box and executor; same table heights for the CPU table (2^20), same hash configuration"})
    return r


def sharded_commit_time(be, rank, world, log_n, cols, coll_dev, reps=2):
    """N > 1 only: PolynomialBatch::from_values of the (replicated) 94 x 2^log_n table under the coset partition -- every
    rank interpolates all columns, extends/hashes its 8/N cosets, the cap slices are all-gathered over RCCL.  Reported
    next to the headline metric; a failure here must not lose the headline line."""
    import torch
    import torch.distributed as dist
    from olavm_amd import sharding
    g = torch.Generator(device="cuda").manual_seed(0x0C05E7)          # same values on every rank
    vals = torch.randint(0, 2**63 - 1, (cols, 1 << log_n), dtype=torch.int64, device="cuda", generator=g)
    times, cap = [], None
    for _ in range(reps + 1):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        b, cap = sharding.commit_sharded(be, rank, world, dev_ptr=vals.data_ptr(), ncols=cols, log_n=log_n, device=coll_dev)
        torch.cuda.synchronize(); dist.barrier()
        times.append(time.perf_counter() - t0)
        b.free()
    t = sharding.max_over_ranks([min(times[1:])], device=coll_dev)[0]
    return {"ms": round(t * 1e3, 3), "workload": f"from_values({cols} x 2^{log_n}), rate_bits 3, coset-sharded over {world} GPUs, "
            "replicated iNTT + caps all-gather (RCCL)", "cap_word0": int(cap[0, 0])}


def sharded_prove_time(be, rank, world, log_n, coll_dev, reps=2, real=False):
    """N > 1 only: the end-to-end proof on the coset partition (ola_set_shard): every rank proves the same traces with its
    8/N cosets of every large table's commitments and quotient; each rank uploads 1/N of the trace columns and the values are
    all-gathered over xGMI; exchanges go through RCCL on the context's stream.  STRONG scaling: the instance is `prove`'s at
    N = 1 (or `prove_real_execution`'s with real=True); rank 0 also proves it unsharded in the same process, so the line carries
    the speed-up measured on this very box."""
    import torch
    import torch.distributed as dist
    from olavm_amd import sharding
    from olavm_amd.air import ola_tables as T
    from olavm_amd.air import tracegen
    blob = T.ola_stark().blob()
    if real:
        from olavm_amd.air import fastexec, miniexec
        count = ((1 << log_n) - 8) // 14
        traces, params, compress = fastexec.instance(miniexec.memory_program(count), range_bits=16, limb_bits=8, max_steps=1 << (log_n + 1))
    else:
        traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=log_n, log_n_mem=log_n)
    be.set_shard(rank, world)
    times, proof = [], b""
    for _ in range(reps + 1):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        proof = be.prove_with_traces(blob, traces, params, compress)
        torch.cuda.synchronize(); dist.barrier()
        times.append(time.perf_counter() - t0)
    exchanges = be.shard_calls // (reps + 1)
    be.set_shard(0, 1)
    t = sharding.max_over_ranks([min(times[1:])], device=coll_dev)[0]
    res = {"seconds": round(t, 4), "proof_bytes": len(proof), "exchanges_per_proof": exchanges, "scaling": "strong",
           "workload": f"prove_with_traces on the coset partition over {world} GPUs, 12 tables, heights 2^{[int(x.shape[1]).bit_length() - 1 for x in traces]}"}
    single = 0.0
    if rank == 0:           # the same proof on one GPU of the same box, for the speed-up and the byte comparison
        be.prove_with_traces(blob, traces, params, compress)
        t0 = time.perf_counter()
        one = be.prove_with_traces(blob, traces, params, compress)
        single = time.perf_counter() - t0
        res.update(verify_proofs(blob, [proof, one], params))
    dist.barrier()
    single = sharding.max_over_ranks([single], device=coll_dev)[0]
    res["single_gpu_seconds_same_box"] = round(single, 4)
    aliased = world > torch.cuda.device_count()          # a dry run with ranks sharing GPUs measures no speed-up
    res["speedup_over_one_gpu"] = round(single / t, 3) if t > 0 and not aliased else None
    if aliased:
        res["devices_aliased"] = True
    return res


def single_process_multi(args):
    """`python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset): this process drives all N GPUs itself, the way the
    reference's single-process caller would (client/src/main.rs:174-214).  Headline: every GPU transforms its own `cols` columns,
    one thread + one context per GPU, no collective (weak scaling).  Then the end-to-end proof on ONE context that spans the N
    GPUs (ola_gpu_init_multi: coset partition and xGMI all-gather inside the library), with the same proof on one GPU of the
    same box next to it.  When the box has fewer than N GPUs the ranks alias the devices it has, and the line says so."""
    import threading
    import torch
    from olavm_amd.backend import Backend, OLA_NTT_EVALUATE
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU fallback")
    N, ndev = args.gpus, torch.cuda.device_count()
    devices = [i % ndev for i in range(N)]
    aliased = ndev < N
    n, cols = 1 << args.log_n, args.cols
    bar = threading.Barrier(N)
    t_end = [0.0] * N
    dev_ms = [0.0] * N
    t_start = [0.0]
    errors = []

    def worker(r):
        try:
            d = devices[r]
            torch.cuda.set_device(d)
            stream = torch.cuda.Stream(device=d)
            with torch.cuda.stream(stream):
                be = Backend(device=d, stream=stream.cuda_stream)
                g = torch.Generator(device=f"cuda:{d}").manual_seed(0x01A5EED + r)
                data = torch.randint(-2**63, 2**63 - 1, (cols, n), dtype=torch.int64, device=f"cuda:{d}", generator=g)
                out, scratch = torch.empty_like(data), torch.empty_like(data)
                torch.cuda.synchronize(d)
                for _ in range(args.warmup):
                    be.ntt_dev(OLA_NTT_EVALUATE, data.data_ptr(), out.data_ptr(), args.log_n, cols, scratch_ptr=scratch.data_ptr())
                torch.cuda.synchronize(d)
                bar.wait()
                if r == 0:
                    t_start[0] = time.perf_counter()
                bar.wait()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(args.steps):
                    be.ntt_dev(OLA_NTT_EVALUATE, data.data_ptr(), out.data_ptr(), args.log_n, cols, scratch_ptr=scratch.data_ptr())
                e1.record(stream)
                torch.cuda.synchronize(d)
                t_end[r] = time.perf_counter()
                dev_ms[r] = e0.elapsed_time(e1)
                bar.wait()
                be.close()
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))
            bar.abort()

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(N)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        raise SystemExit("bench.py --gpus %d: %s" % (N, errors[0]))
    elapsed = max(t_end) - t_start[0]
    from olavm_amd import sharding
    value = sharding.aggregate_throughput(cols, 16.0 * n, N, args.steps, elapsed)
    passes = 1 if args.log_n <= 13 else (2 if args.log_n <= 16 else 3)
    launch_ms = max(dev_ms) / args.steps / passes
    achieved = 16.0 * n * cols / passes / (launch_ms * 1e-3) / 1e9
    res = {"metric": "goldilocks_ntt_throughput", "value": round(value, 2), "unit": "GB/s", "n_gpus": N, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"standalone batched Goldilocks NTT (cfft::evaluate_poly, natural in/out), {cols} columns x 2^{args.log_n} rows per GPU, "
                                  "inputs resident in HBM", "log_n": args.log_n, "columns_per_gpu": cols,
                      "parallelism": f"columns sharded over {N} GPU(s), no collective",
                      "launch": "single process: one thread and one context per GPU", "devices": devices, "devices_aliased": aliased},
           "roofline": {"bound": "hbm by contract; VALU issue binds (64-bit modmul on a 32-bit pipe)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                        "traffic": None, "algorithmic_bytes_per_launch": 16.0 * n * cols / passes, "kernel": "ntt2t_pass_kernel", "launches_per_step": passes,
                        "avg_launch_ms": round(launch_ms, 4), "note": "per GPU, slowest rank" + ("; ranks share physical GPUs" if aliased else "")}}
    rec = pmc_record(args.log_n, cols)          # the same per-launch counters as the N = 1 line (every rank runs the same kernels on its own columns)
    if rec:
        res["roofline"]["traffic"] = rec["traffic"]
        ipe = rec["valu_insts_per_element"]
        floor_ms = valu_floor_ms(ipe, cols * n)
        res["roofline"].update({"valu_insts_per_element": round(ipe, 1), "valu_floor_ms": round(floor_ms, 3),
                                "frac_of_floor": round(floor_ms / (launch_ms * passes), 3)})
    try:
        res["collective"] = collective_report(Backend, devices, aliased, args.log_n, cols)
    except Exception as e:              # noqa: BLE001 -- an extra: never at the price of the headline line
        res["collective"] = {"error": repr(e)[:300]}
    if not args.no_prove:
        carrier = os.environ.get("OLA_COLLECTIVE", "peer")
        try:
            res["prove_sharded"] = multi_context_prove(Backend, devices, args.log_n, real=False, aliased=aliased, collective=carrier)
            res["prove_real_execution_sharded"] = multi_context_prove(Backend, devices, args.log_n, real=True, aliased=aliased, collective=carrier)
        except Exception as e:          # noqa: BLE001 -- an extra: never at the price of the headline line
            res.setdefault("prove_sharded", {"error": repr(e)[:300]})
    emit(res)


def collective_report(Backend, devices, aliased, log_n, cols):
    """Which carrier a multi-device context over `devices` gets for each setting of OLA_COLLECTIVE (ola_gpu_collective: did RCCL
    see N ranks, or why not), and -- on distinct physical GPUs -- both carriers timed by themselves for the two exchanges that
    matter: the 512-byte Merkle-cap gather (latency) and the trace gather of a 2^log_n-row, `cols`-column table (bandwidth)."""
    N = len(devices)
    out = {"requested_by_default": os.environ.get("OLA_COLLECTIVE", "peer")}
    trace_block = (cols * (8 << log_n) // N + 7) // 8 * 8
    for carrier in ("peer", "rccl"):
        be = Backend(devices=devices, collective=carrier)
        try:
            c = be.collective()
            entry = {"got": c["carrier"], "ranks": c["ranks"], "note": c["note"]}
            if c["carrier"] == carrier and not aliased:
                for name, size, reps in (("cap_gather_512B", 512, 50), ("trace_gather", trace_block, 5)):
                    ms, bad = be.all_gather_check(carrier, size, reps=reps)
                    entry[name] = {"bytes_per_rank": size, "ms": round(ms, 4), "wrong_bytes": int(bad),
                                   "GBps_received_per_rank": round(size * (N - 1) / (ms * 1e-3) / 1e9, 2) if ms > 0 else None}
            out[carrier] = entry
        finally:
            be.close()
    if aliased:
        out["note"] = "ranks share physical GPUs on this box: RCCL cannot form a communicator over them and nothing is timed"
    return out


def multi_context_prove(Backend, devices, log_n, real, reps=2, aliased=False, collective="peer"):
    """STRONG scaling of the whole proof: ONE ola_prove_with_traces call on a context that spans `devices`, against the same
    call on a single-device context of the same box; bytes compared, oracle-verified."""
    from olavm_amd.air import ola_tables as T
    from olavm_amd.air import tracegen
    blob = T.ola_stark().blob()
    if real:
        from olavm_amd.air import fastexec, miniexec
        count = ((1 << log_n) - 8) // 14
        traces, params, compress = fastexec.instance(miniexec.memory_program(count), range_bits=16, limb_bits=8, max_steps=1 << (log_n + 1))
    else:
        traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=log_n, log_n_mem=log_n)
    mb = Backend(devices=devices, collective=collective)
    carrier = mb.collective()
    mb.proof_stats(enable=True)
    ts, proof = [], b""
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        proof = mb.prove_with_traces(blob, traces, params, compress)
        ts.append(time.perf_counter() - t0)
    st = mb.proof_stats(enable=False)
    mb.close()
    one = Backend(device=devices[0])
    one.prove_with_traces(blob, traces, params, compress)
    t0 = time.perf_counter()
    single = one.prove_with_traces(blob, traces, params, compress)
    t1 = time.perf_counter() - t0
    one.close()
    t = min(ts[1:])
    # a speed-up is only a speed-up on distinct physical GPUs; with aliased ranks one device does all ranks' work
    speed = {"speedup_over_one_gpu": round(t1 / t, 3)} if not aliased else {"speedup_over_one_gpu": None, "devices_aliased": True}
    return {"seconds": round(t, 4), "first_call_seconds": round(ts[0], 4), "proof_bytes": len(proof), "scaling": "strong",
            "exchanges_per_proof": st["peer_exchanges"] // (reps + 1), "bytes_moved_between_gpus_per_proof": st["peer_bytes_moved"] // (reps + 1),
            "single_gpu_seconds_same_box": round(t1, 4), **speed,
            "collective": ("RCCL: ncclAllGather on one communicator per device (olavm_amd/csrc/rccl_carrier.h)" if carrier["carrier"] == "rccl" else
                           "library all-gather: peer-to-peer pulls over xGMI ordered by stream events (olavm_amd/csrc/peer_group.h)"),
            "collective_ranks": carrier["ranks"], "collective_note": carrier["note"],
            "workload": f"ONE ola_prove_with_traces call on a context spanning {len(devices)} GPUs (devices {devices}), 12 tables, heights 2^{[int(x.shape[1]).bit_length() - 1 for x in traces]}"
                        + (", executed program" if real else ""),
            **verify_proofs(blob, [proof, single], params)}


DETAILS_FILE = os.environ.get("OLA_BENCH_DETAILS", os.path.join(ROOT, "bench_details.json"))
LINE_LIMIT = 4096           # the driver's parser lost round 5's 22 KB line; tests/test_bench_line.py holds the line to this
VALU_CYCLES_PER_ISSUE = 4.0  # measured mix of the pass kernels at their 3 - 4 waves per SIMD (3.1 - 3.4 cycles for plain 32-bit
                             # instructions, 5.2 - 5.4 for 64-bit / carry ones: profiles/r04_sq_counters_and_issue_rates.txt)
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9


def valu_floor_ms(insts_per_element, elements):
    """The time the transform's instructions need on the integer pipe at the measured issue cost: what bounds the NTT."""
    return insts_per_element * elements / 64.0 * VALU_CYCLES_PER_ISSUE / SIMD_CYCLES_PER_S * 1e3


def _num(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def _short(s, n=80):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "~"


def compact_line(res):
    """The ONE line the driver parses: the contract's keys, `roofline`, `roofline_lde`, `cpu_baseline` and a flat map of proof
    seconds -- numbers and strings of at most 80 characters.  Everything else bench.py measures goes to bench_details.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: res[k] for k in keep if k in res}
    cfg = res.get("config", {})
    out["config"] = {"workload": _short(cfg.get("workload_short") or cfg.get("workload", ""), 100), "log_n": cfg.get("log_n"), "columns_per_gpu": cfg.get("columns_per_gpu"),
                     "parallelism": _short(cfg.get("parallelism", ""), 60)}
    for k in ("launch", "devices_aliased"):
        if k in cfg:
            out["config"][k] = _short(cfg[k], 48) if isinstance(cfg[k], str) else cfg[k]
    r = res.get("roofline")
    if isinstance(r, dict):
        rr = {"bound": _short(r.get("bound_short", r.get("bound", "hbm"))), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
              "frac": r.get("frac"), "traffic": r.get("traffic"), "kernel": _short(r.get("kernel_short", r.get("kernel", ""))),
              "avg_launch_ms": r.get("avg_launch_ms"), "launches_per_step": r.get("launches_per_step"),
              "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch")}
        for k in ("valu_insts_per_element", "valu_floor_ms", "frac_of_floor", "dominant"):
            if r.get(k) is not None:
                rr[k] = r[k]
        out["roofline"] = rr
    r = res.get("roofline_lde")
    if isinstance(r, dict) and "error" not in r:
        out["roofline_lde"] = {"bound": _short(r.get("bound_short", r.get("bound", "hbm"))), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
                               "frac": r.get("frac"), "ms": r.get("ms"), "streamed_GBps": r.get("streamed_GBps")}
    c = res.get("cpu_baseline")
    if isinstance(c, dict):
        cc = {k: c[k] for k in ("value", "unit", "cores", "kind") if k in c}
        cc["sample"] = _short(c.get("sample", ""))
        for leg in ("prove", "prove_small"):
            p = c.get(leg)
            if isinstance(p, dict):
                cc[leg] = {k: p[k] for k in ("log_n", "cpu_seconds", "gpu_seconds", "ratio", "identical_bytes", "kind", "error") if k in p}
                if "error" in cc[leg]:
                    cc[leg]["error"] = _short(cc[leg]["error"])
        out["cpu_baseline"] = cc
    proofs = {}
    for name, path in (("poseidon_2p22", ("prove", "seconds")), ("blake3_2p22", ("prove", "blake3_config", "seconds")),
                       ("real_poseidon_2p22", ("prove_real_execution", "seconds")), ("real_blake3_2p22", ("prove_real_execution", "blake3_config", "seconds")),
                       ("readme_fibo_blake3", ("readme_fibo_loop_blake3", "seconds")), ("config4", ("config4_poseidon_heavy", "seconds")),
                       ("2p24", ("prove_2p24_rows", "seconds")), ("sharded", ("prove_sharded", "seconds")),
                       ("sharded_one_gpu_same_box", ("prove_sharded", "single_gpu_seconds_same_box")),
                       ("real_sharded", ("prove_real_execution_sharded", "seconds")), ("commit_sharded_ms", ("commit_sharded", "ms"))):
        v = _num(res, *path)
        if v is not None:
            proofs[name] = v
    if proofs:
        out["proofs"] = proofs
        flags = []
        for k in ("prove", "prove_real_execution", "readme_fibo_loop_blake3", "config4_poseidon_heavy", "prove_2p24_rows", "prove_sharded", "prove_real_execution_sharded"):
            for d in (res.get(k), _num(res, k, "blake3_config")):
                if isinstance(d, dict) and "verified" in d:
                    flags.append(bool(d["verified"]))
        out["proofs_verified"] = bool(flags) and all(flags)
        out["proofs_unit"] = "s"
    st = res.get("start")
    if isinstance(st, dict):
        out["start"] = {k: v for k, v in st.items() if isinstance(v, (int, float, bool))}
    # the HBM-class kernels of the openings (Blake3-configuration 2^22-row proof: bytes every coefficient once / device time of the
    # phase) and the launch-bound tables of the README-shape proof
    for name, path in (("open_eval_frac", ("prove", "blake3_config", "kernels", "open_eval", "frac")),
                       ("fri_fold_frac", ("prove", "blake3_config", "kernels", "fri_fold", "frac_of_hbm_peak")),
                       ("fri_fold_ms", ("prove", "blake3_config", "kernels", "fri_fold", "ms")),
                       ("fri_fold_dominant_frac", ("prove", "blake3_config", "kernels", "fri_fold", "dominant_launch", "frac")),
                       ("open_eval_dominant_frac", ("prove", "blake3_config", "kernels", "open_eval", "dominant_launch", "frac")),
                       ("quotient_ms", ("prove", "blake3_config", "kernels", "quotient", "ms")),
                       ("small_tables_ms", ("readme_fibo_loop_blake3", "partition", "replicated_breakdown_ms", "tables_below_the_partition_threshold"))):
        v = res.get(name) if res.get(name) is not None else _num(res, *path)
        if v is not None:
            out[name] = v
    out["details"] = os.path.basename(DETAILS_FILE)
    return out


def emit(res):
    """Details to the side file, then the compact line: the LAST thing on stdout, one line, under LINE_LIMIT bytes."""
    try:
        with open(DETAILS_FILE, "w") as f:
            json.dump(res, f, indent=1)
    except OSError as e:
        print(f"bench.py: could not write {DETAILS_FILE}: {e}", file=sys.stderr)
    line = json.dumps(compact_line(res), separators=(",", ":"))
    if len(line) >= LINE_LIMIT:          # never lose the headline to an extra: drop the optional blocks, largest first
        c = compact_line(res)
        for k in ("start", "proofs", "roofline_lde"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    sys.stderr.flush()
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--cols", type=int, default=94)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-prove-log-n", type=int, default=17, help="rows (log2) of the large tables in cpu_baseline.prove")
    ap.add_argument("--no-prove", action="store_true", help="skip the end-to-end prove_with_traces timing")
    ap.add_argument("--no-config4", action="store_true", help="skip the Poseidon-heavy proof (BASELINE config 4) and its Merkle / FRI block")
    ap.add_argument("--no-2p24", action="store_true", help="skip the 2^24-row single-GPU proof (BASELINE config 5's N = 1 point)")
    args = ap.parse_args()

    # N > 1 without a launcher: this process drives the N GPUs itself.  Under torch.distributed.run (WORLD_SIZE set) the ranks
    # are the launcher's processes; n_gpus in the line is always the number of ranks that really ran.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.gpus not in (2, 4, 8):
            raise SystemExit("--gpus must be 1, 2, 4 or 8")
        return single_process_multi(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU fallback")
    # OLA_BENCH_BACKEND=gloo lets several ranks share one GPU (dry run of the N > 1 path on a single-GPU box); the real
    # multi-GPU run uses RCCL
    backend = os.environ.get("OLA_BENCH_BACKEND", "nccl")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from olavm_amd.backend import Backend, OLA_NTT_EVALUATE
    # one explicit stream for torch and the library (torch's default stream is the null handle, which would make the library
    # create a stream of its own, unordered with torch's work)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    be = Backend(device=local_rank, stream=stream.cuda_stream)

    # The first proof of a fresh process (what `ola prove` sees), measured FIRST: before this process has allocated anything large.
    # VRAM that a still-living process has freed is handed out dirty and is scrubbed inside the next owner's hipMalloc (about 30 ms
    # per GB), which would be charged to whichever child runs first (docs/EXPERIMENTS.md, cold start).
    cold = cold_b3 = None
    if world == 1 and rank == 0 and not args.no_prove:
        # The very first GPU-heavy process on a box also pays for what the PREVIOUS tenant left behind (dirty VRAM is scrubbed inside
        # the new owner's hipMalloc): reported under its own name, so that the figures after it describe this library and not the
        # box's history.  Measured: 1.9x / 3.7x for whichever child ran first, 1.16 - 1.18x for the ones after it, either order.
        first = cold_process_prove(args.log_n, local_rank, early=True)
        cold = {"first_process_on_this_box": first,
                "cold_process": cold_process_prove(args.log_n, local_rank, early=True),
                "cold_process_without_early_hook": cold_process_prove(args.log_n, local_rank, early=False)}
        cold_b3 = {"cold_process": cold_process_prove(args.log_n, local_rank, early=True, hasher="blake3"),
                   "cold_process_without_early_hook": cold_process_prove(args.log_n, local_rank, early=False, hasher="blake3")}

    n = 1 << args.log_n
    cols = args.cols
    # synthetic trace columns: uniform 64-bit words reduced mod p on the fly by the kernels (inputs may be non-canonical)
    g = torch.Generator(device="cuda").manual_seed(0x01A5EED + rank)
    data = torch.randint(-2**63, 2**63 - 1, (cols, n), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty_like(data)
    scratch = torch.empty_like(data)

    def step():
        be.ntt_dev(OLA_NTT_EVALUATE, data.data_ptr(), out.data_ptr(), args.log_n, cols, scratch_ptr=scratch.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()          # torch's default stream hands over a null handle: the context then runs on a stream of its own
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    from olavm_amd import sharding
    elapsed, dev_ms = sharding.max_over_ranks([elapsed, dev_ms], device=coll_dev)
    # `steps` more transforms with the library's events around every pass launch (ola_gpu_ntt_pass_times): which pass is the
    # dominant kernel and how long one launch of it takes -- outside the timed region, which runs without the extra records
    pass_times, pass_times_error = None, None
    try:
        be.ntt_pass_times(enable=True)
        for _ in range(args.steps):
            step()
        pass_times = be.ntt_pass_times(enable=False)
    except Exception as e:              # noqa: BLE001 -- keep the headline
        pass_times_error = {"error": repr(e)[:200]}

    # N > 1 extras (coset-partitioned commitment and proof).  They run in a helper thread under a wall-clock guard: whatever
    # happens there -- an exception or a collective that never returns -- the headline line below is still printed.
    sharded = sharded_prove = sharded_prove_real = None
    extras_hung = False
    if world > 1 and world in (2, 4, 8) and os.environ.get("OLA_BENCH_SHARDED", "1") != "0":
        import threading
        del data, out, scratch
        torch.cuda.empty_cache()
        box = {}

        def extras():
            torch.cuda.set_device(local_rank)
            try:
                box["commit"] = sharded_commit_time(be, rank, world, args.log_n, cols, coll_dev)
            except Exception as e:                   # noqa: BLE001 -- keep the headline line
                box["commit"] = {"error": repr(e)[:200]}
            if not args.no_prove:
                try:
                    box["prove"] = sharded_prove_time(be, rank, world, args.log_n, coll_dev)
                except Exception as e:               # noqa: BLE001
                    box["prove"] = {"error": repr(e)[:200]}
                try:
                    box["prove_real"] = sharded_prove_time(be, rank, world, args.log_n, coll_dev, real=True)
                except Exception as e:               # noqa: BLE001
                    box["prove_real"] = {"error": repr(e)[:200]}

        th = threading.Thread(target=extras, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("OLA_BENCH_SHARDED_TIMEOUT", "420")))
        extras_hung = th.is_alive()
        sharded = box.get("commit", {"error": "timed out"} if extras_hung else None)
        sharded_prove = box.get("prove", {"error": "timed out"} if extras_hung and not args.no_prove else None)
        sharded_prove_real = box.get("prove_real")

    if rank == 0:
        bytes_per_step = 16.0 * n * cols            # algorithmic: one read + one write of every element
        value = sharding.aggregate_throughput(cols, 16.0 * n, world, args.steps, elapsed)
        # dominant kernel = ntt_pass_kernel; a 2^22 natural-order transform launches it `passes` times over the batch
        passes = 1 if args.log_n <= 13 else (2 if args.log_n <= 16 else 3)      # ntt2_run: (log_n + 7) / 8 passes from 2^14 on
        launch_ms = dev_ms / args.steps / passes
        achieved = bytes_per_step / passes / (launch_ms * 1e-3) / 1e9
        rec = pmc_record(args.log_n, cols)
        traffic, traffic_src = (rec["traffic"], rec["source"]) if rec else (None, None)
        # the DOMINANT kernel: the slowest of the transform's pass launches, from the library's own events around every launch
        # (ola_gpu_ntt_pass_times) over `steps` more transforms right after the timed region -- the timed region itself runs
        # without them, and the mean over the passes (dev_ms / steps / passes) stays on the record as `mean_launch_ms`
        dom_name, dom_ms, per_pass = "ntt2t_pass_kernel", launch_ms, pass_times_error
        if pass_times:
            per_pass = {k: round(v["avg_ms"], 4) for k, v in pass_times.items()}
            dom_name = max(pass_times, key=lambda k: pass_times[k]["avg_ms"])
            dom_ms = pass_times[dom_name]["avg_ms"]
        achieved = bytes_per_step / passes / (dom_ms * 1e-3) / 1e9
        ipe = rec["valu_insts_per_element"] if rec else None
        floor_ms = valu_floor_ms(ipe, cols * n) if ipe else None
        step_ms = dev_ms / args.steps
        res = {
            "metric": "goldilocks_ntt_throughput", "value": round(value, 2), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"standalone batched Goldilocks NTT (cfft::evaluate_poly, natural in/out), "
                                   f"{cols} columns x 2^{args.log_n} rows per GPU, inputs resident in HBM",
                       "workload_short": f"batched Goldilocks NTT (cfft::evaluate_poly), {cols} cols x 2^{args.log_n} rows per GPU, resident",
                       "log_n": args.log_n, "columns_per_gpu": cols, "parallelism": f"columns sharded over {world} GPU(s), no collective",
                       "launch": "torch.distributed.run, one process per GPU" if world > 1 else "single process", "gpus_requested": args.gpus},
            "roofline": {"bound": "hbm (contract); measured: VALU issue of 64-bit modular arithmetic on the 32-bit integer pipe",
                         "bound_short": "hbm by contract; VALU issue binds (64-bit modmul on a 32-bit pipe)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_step / passes,
                         "kernel": dom_name + " (the slowest of the transform's passes)", "kernel_short": dom_name, "launches_per_step": passes,
                         "avg_launch_ms": round(dom_ms, 4), "mean_launch_ms_over_the_passes": round(launch_ms, 4), "per_pass_avg_ms": per_pass,
                         "dominant": {"whole_transform_ms": round(step_ms, 4), "whole_transform_frac": round(bytes_per_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                         "valu_insts_per_element": round(ipe, 1) if ipe else None,
                         "valu_floor_ms": round(floor_ms, 3) if floor_ms else None,
                         "frac_of_floor": round(floor_ms / step_ms, 3) if floor_ms else None,
                         "valu_floor_note": f"instructions per element (PMC) x elements / 64 lanes x {VALU_CYCLES_PER_ISSUE} cycles per issue (measured mix at 3 - 4 waves) "
                                            "/ (256 CUs x 4 SIMDs x 2.4 GHz): the whole transform's floor on the integer pipe; frac_of_floor = floor / measured transform",
                         "valu": valu_roofline(rec, args.log_n, cols, launch_ms * passes),
                         "note": "every pass streams the whole batch once (traffic = 3 x algorithmic per launch); the passes sit between "
                                 "the VALU-issue ceiling of 64-bit modular arithmetic on a 32-bit integer pipe and the 128-byte-segment "
                                 "HBM rate, see DESIGN.md"},
        }
        if sharded is not None:
            res["commit_sharded"] = sharded
        if sharded_prove is not None:
            res["prove_sharded"] = sharded_prove
        if sharded_prove_real is not None:
            res["prove_real_execution_sharded"] = sharded_prove_real
        if world == 1:
            del data, out, scratch
            torch.cuda.empty_cache()
            try:        # the transform the prover runs: leaf-order coset LDE x8, on the record next to the natural-order NTT
                res["roofline_lde"] = lde_roofline(be, torch, stream, args.log_n, cols)
            except Exception as e:      # noqa: BLE001 -- an extra: never at the price of the headline line
                res["roofline_lde"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_prove:
            try:
                be_b3 = Backend(device=local_rank, stream=stream.cuda_stream, hasher="blake3")
            except Exception:
                be_b3 = None
            res["prove"] = prove_time(be, args.log_n, be_b3=be_b3)
            be.trim()
            try:
                res["prove_real_execution"] = prove_time_real(be, args.log_n, be_b3=be_b3)
            except Exception as e:          # an extra: never at the price of the headline line
                res["prove_real_execution"] = {"error": repr(e)[:200]}
            if be_b3 is not None:
                try:
                    res["readme_fibo_loop_blake3"] = readme_fibo_loop_blake3(be_b3)
                except Exception as e:      # an extra: never at the price of the headline line
                    res["readme_fibo_loop_blake3"] = {"error": repr(e)[:200]}
                be_b3.close()
            be.trim()
            torch.cuda.empty_cache()
            res["prove"].update(cold)
            try:        # the start-up figures of `ola prove`'s call order, for the compact line (Blake3 configuration: the shorter proof, the larger share)
                e, l = cold_b3["cold_process"], cold_b3["cold_process_without_early_hook"]
                res["start"] = {"init_seconds_without_early_hook": l["init_seconds_seen_by_the_prover"], "init_seconds_with_early_hook": e["init_seconds_seen_by_the_prover"],
                                "warmup_thread_ms": e["warmup_thread_ms"], "early_hook_excess_over_warm": e["excess_over_warm"],
                                "no_hook_excess_over_warm": l["excess_over_warm"], "warm_blake3_2p22_seconds": e["second_proof_seconds"]}
            except (KeyError, TypeError):
                pass
            if isinstance(res["prove"].get("blake3_config"), dict):
                res["prove"]["blake3_config"].update(cold_b3)
            if args.log_n == 22 and not args.no_config4:
                try:
                    res["config4_poseidon_heavy"] = prove_config4(be, args.log_n)
                except Exception as e:      # an extra: never at the price of the headline line
                    res["config4_poseidon_heavy"] = {"error": repr(e)[:200]}
            if args.log_n == 22 and not args.no_2p24:
                try:
                    res["prove_2p24_rows"] = prove_time_2p24(be)
                except Exception as e:      # an extra: never at the price of the headline line
                    res["prove_2p24_rows"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.log_n)
            if not args.no_prove:
                for leg, ln in (("prove", args.cpu_prove_log_n), ("prove_small", 14)):
                    try:
                        res["cpu_baseline"][leg] = cpu_baseline_prove(be, ln)
                    except Exception as e:      # noqa: BLE001 -- an extra: never at the price of the headline line
                        res["cpu_baseline"][leg] = {"error": repr(e)[:200]}
        emit(res)
    if extras_hung:          # a collective of the extras never returned: the line is out, leave without touching NCCL again
        sys.stdout.flush()
        os._exit(0)
    be.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
