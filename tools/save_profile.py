#!/usr/bin/env python3
"""Copy the rocprofv3 per-kernel stats of a gpurun_out/<name>/ run into profiles/<tag>_kernel_stats.csv (kernel names
shortened) together with the bench JSON line.  usage: tools/save_profile.py <gpurun_out subdir> <tag>"""
import csv, os, re, sys
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", src)
stats = [f for f in os.listdir(d) if f.endswith("kernel_stats.csv")][0]
rows = list(csv.reader(open(os.path.join(d, stats))))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = re.sub(r"\(anonymous namespace\)::", "", r[0])[:120]
        w.writerow(r)
for name in ("bench.log", "cmd.log"):
    if os.path.exists(os.path.join(d, name)):
        log = [l for l in open(os.path.join(d, name)).read().splitlines() if not re.match(r"^[EWI]\d{8} ", l)]
        js = [l for l in log if l.startswith("{")]
        if js:
            open(os.path.join(root, "profiles", f"{tag}_bench.json"), "w").write(js[-1] + "\n")
        else:
            open(os.path.join(root, "profiles", f"{tag}_run.log"), "w").write("\n".join(log[-20:]) + "\n")
print("saved", tag, len(rows) - 1, "kernels")
