#!/usr/bin/env python3
"""Aggregate the rocprofv3 counter CSVs of tools/pmc_ntt.sh into profiles/<tag>_ntt_pmc.json (see that script)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NTT_SOURCES = ["olavm_amd/csrc/gl.cuh", "olavm_amd/csrc/ntt.hip", "olavm_amd/csrc/ntt2.hip", "olavm_amd/csrc/ntt2t.cuh", "olavm_amd/csrc/tform.cuh"]


def source_hash():
    h = hashlib.sha256()
    for f in NTT_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "TCC_HIT_sum", "TCC_MISS_sum")


def collect(src, prefix=""):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for ctr in COUNTERS:
        files = glob.glob(os.path.join(src, prefix + ctr, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != ctr:
                continue
            k = r["Kernel_Name"].split("(")[0][:70]
            per[k][ctr][0] += 1
            per[k][ctr][1] += float(r["Counter_Value"])
    kernels = {}
    for k, c in per.items():
        kernels[k] = {ctr: {"dispatches": n, "per_dispatch": v / n} for ctr, (n, v) in c.items()}
    return kernels


def pass_totals(kernels, names):
    """FETCH / WRITE (KiB) and L2 hits / misses summed over the NTT pass kernels, per 94 x 2^22 transform"""
    passes = {k: v for k, v in kernels.items() if any(nm in k for nm in names) and "WRITE_SIZE" in v}
    tot = {c: sum(v[c]["per_dispatch"] * v[c]["dispatches"] for v in passes.values() if c in v) for c in COUNTERS}
    return passes, tot


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kernels = collect(src)
    n, cols = 1 << 22, 94
    elems = n * cols
    # the 2^22 natural-order transform is two strided 7-bit passes and the closing natural-order 8-bit pass
    names = ("ntt2_pass_kernel<7, 0, false, 4, false>", "ntt2_pass_kernel_w4<7, 0, false, 4>", "ntt2_pass_kernel<7, 0, false, 8, false>",
             "ntt2_pass_kernel_w4<7, 0, false, 8>", "ntt2_pass_kernel<8, 2, false, 1, false>", "ntt2_pass_kernel<8, 2, false, 4, false>",
             # round 4: the T-form passes (ntt2t.cuh): strided without / with load multipliers, closing natural-order pass
             "ntt2t_pass_kernel<7, 0, false, 8, 0>", "ntt2t_pass_kernel<7, 0, false, 8, 1>", "ntt2t_pass_kernel<8, 2, false, 8, 2>")
    passes, tot = pass_totals(kernels, names)
    g_kernels = collect(src, "G96_")
    g_passes, g_tot = pass_totals(g_kernels, names)
    g_launches = sum(v["WRITE_SIZE"]["dispatches"] for v in g_passes.values())

    def per_transform(t, transforms):
        if not transforms:
            return None
        hit, miss = t.get("TCC_HIT_sum", 0), t.get("TCC_MISS_sum", 0)
        return {"fetch_x2_plus_write_bytes": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / transforms,
                "TCC_HIT_sum": hit / transforms, "TCC_MISS_sum": miss / transforms,
                "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
    launches = sum(v["WRITE_SIZE"]["dispatches"] for v in passes.values())
    fetch = sum(v["FETCH_SIZE"]["per_dispatch"] * v["FETCH_SIZE"]["dispatches"] for v in passes.values() if "FETCH_SIZE" in v)
    write = sum(v["WRITE_SIZE"]["per_dispatch"] * v["WRITE_SIZE"]["dispatches"] for v in passes.values())
    valu = sum(v["SQ_INSTS_VALU"]["per_dispatch"] * v["SQ_INSTS_VALU"]["dispatches"] for v in passes.values() if "SQ_INSTS_VALU" in v)
    transforms = launches / 3.0 if launches else 0
    out = {
        "command": "tools/pmc_ntt.sh (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU in separate passes, --kernel-trace) over tools/pmc_workload.py",
        "timestamp": time.strftime("%Y-%m-%d %H:%M:%S"),
        "source_sha16": source_hash(), "sources": NTT_SOURCES,
        "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; FETCH_SIZE counts half of a wide streaming read on gfx950 (MI355X_MICROARCH.md) -> x2",
        "shape": {"log_n": 22, "columns": cols},
        "ntt_94x2^22": {
            "pass_launches_profiled": launches, "transforms": transforms,
            "traffic_bytes_per_launch": (2 * fetch + write) * 1024 / launches if launches else None,
            "algorithmic_bytes_per_launch": 16.0 * elems / 3,
            "valu_wave_insts_per_transform": valu / transforms if transforms else None,
            "valu_insts_per_element": valu * 64 / (transforms * elems) if transforms else None,
        },
        "infinity_cache_blocking_experiment": {
            "what": "the same 3 transforms launched over the whole batch (group 0) and in column groups of 96 MB (OLA_NTT2_GROUP_MB=96: all three "
                    "passes of a group back to back, so that a pass could read its predecessor's output from the 256 MiB Infinity Cache)",
            "whole_batch_per_transform": per_transform(tot, transforms),
            "groups_of_96MB_per_transform": per_transform(g_tot, 3.0 if g_launches else 0),
            "groups_of_96MB_launches": g_launches,
        },
        "calibration": {
            "leaf_hash_colmajor_kernel": {"algorithmic_read_kib": 94 * (1 << 22) * 8 / 1024,
                                          "FETCH_SIZE_kib": kernels.get("ola::leaf_hash_colmajor_kernel", {}).get("FETCH_SIZE", {}).get("per_dispatch")},
        },
        "kernels": kernels,
    }
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["ntt_94x2^22"], indent=1))


if __name__ == "__main__":
    main()
