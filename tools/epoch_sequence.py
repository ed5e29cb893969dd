#!/usr/bin/env python3
"""Ordered kernel sequence of ONE epoch from a rocprofv3 --kernel-trace run of tools/epoch_profile.py (development tool).

    python3 tools/epoch_sequence.py gpurun_out/X/cora_prof --epochs 35     # 5 warm-up + 30 timed epochs were traced
Prints the launches of the last epoch in start order (short names, duration in us)."""
import argparse
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--epochs", type=int, default=35)
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f, newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the graph build (K0 kernels, sorts) precedes the epochs: keep what follows the last slot_meta / symmetric_perm launch
last_build = max((i for i, r in enumerate(rows) if "slot_meta" in r["Kernel_Name"] or "csr_symmetric_perm" in r["Kernel_Name"]), default=-1)
rows = rows[last_build + 1:]
per = len(rows) // a.epochs
last = rows[-per:]
print(f"{len(rows)} launches after the graph build, {per} per epoch")
for i, r in enumerate(last):
    n = r["Kernel_Name"].replace("void ", "").replace("pygat::", "")
    n = re.sub(r"at::native::\(anonymous namespace\)::|at::native::|\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    if len(n) > 90:
        n = n[:87] + "..."
    print(f"{i:3d} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {n}")
