#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run, as a short table (and optionally a CSV kept under profiles/).

    python3 tools/rocprof_top.py gpurun_out/X [--per N] [--csv profiles/rNN_x.csv] [--top 25]
`--per N`: also print time per N (epochs / steps) so that a line reads "us per epoch"."""
import argparse
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--per", type=float, default=0)
ap.add_argument("--csv", default="")
ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()
files = glob.glob(os.path.join(a.dir, "**", "*kernel_stats.csv"), recursive=True)
assert files, f"no *kernel_stats.csv under {a.dir}"
rows = []
for f in files:
    with open(f, newline="") as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{len(rows)} kernels, {tot / 1e6:.3f} ms of kernel time" + (f", {tot / 1e3 / a.per:.1f} us per unit" if a.per else ""))
for r in rows[:a.top]:
    name = r["Name"].replace("void ", "").replace("pygat::", "")
    if len(name) > 70:
        name = name[:67] + "..."
    per = f" {float(r['TotalDurationNs']) / 1e3 / a.per:9.1f} us/unit" if a.per else ""
    print(f"{name:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.1f} %{per}")
if a.csv:
    with open(a.csv, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
