#!/usr/bin/env python3
"""Per-kernel HBM traffic and VALU instruction counts of one proof, from the rocprofv3 counter CSVs of tools/pmc_proof.sh."""
import collections
import csv
import glob
import os
import sys

COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")


def short(name):
    k = name.split("(")[0]
    for p in ("void ", "ola::"):
        k = k.replace(p, "")
    return k[:58]


def main():
    src, hasher = sys.argv[1], sys.argv[2]
    per = collections.defaultdict(lambda: {c: [0, 0.0] for c in COUNTERS})
    for ctr in COUNTERS:
        files = glob.glob(os.path.join(src, ctr, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("# no counter file for", ctr)
            continue
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != ctr:
                continue
            e = per[short(r["Kernel_Name"])][ctr]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    rows = []
    for k, c in per.items():
        traffic = (2 * c["FETCH_SIZE"][1] + c["WRITE_SIZE"][1]) * 1024          # KiB -> bytes; FETCH_SIZE counts half of a wide read on gfx950
        rows.append((traffic, k, max(c[x][0] for x in COUNTERS), c["FETCH_SIZE"][1] * 2048, c["WRITE_SIZE"][1] * 1024, c["SQ_INSTS_VALU"][1]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"# tools/pmc_proof.sh: one prove_with_traces (12 tables, CPU and memory tables 2^22 rows, {hasher} configuration), rocprofv3 --pmc in three passes")
    print("# (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; --kernel-trace).  traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; gfx950 correction of")
    print("# MI355X_MICROARCH.md).  GB per proof, summed over the kernel's dispatches; VALU = wave-instructions per proof.")
    print(f"# all kernels: {tot / 1e9:.1f} GB per proof")
    print(f"{'kernel':58s} {'launches':>8s} {'read GB':>9s} {'write GB':>9s} {'traffic GB':>10s} {'share':>6s} {'VALU G':>8s}")
    for traffic, k, n, rd, wr, valu in rows[:28]:
        print(f"{k:58s} {n:8d} {rd / 1e9:9.2f} {wr / 1e9:9.2f} {traffic / 1e9:10.2f} {traffic / tot:6.1%} {valu / 1e9:8.2f}")


if __name__ == "__main__":
    main()
