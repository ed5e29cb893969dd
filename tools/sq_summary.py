#!/usr/bin/env python3
"""Fold rocprofv3 SQ counter passes into per-kernel means (development / evidence tool).

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \\
              -d gpurun_out/X/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-epoch
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv \\
              -d gpurun_out/X/sq2 -- python3 bench.py ...
    python3 tools/sq_summary.py gpurun_out/X/sq1 gpurun_out/X/sq2 --out profiles/rNN_sq_counters.json

Counters are summed over the 8 XCDs per dispatch, then averaged over the launches of a kernel.  Derived: MFMA-busy
fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f, newline="")):
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
            for (disp, ctr), v in per.items():
                k = names[disp].split("(")[0].replace("void ", "").strip()
                if k.startswith("pygat::"):
                    acc[k][ctr].append(v)
    out = {}
    for k, ctrs in acc.items():
        e = {c: sum(v) / len(v) for c, v in ctrs.items()}
        e["launches_seen"] = max(len(v) for v in ctrs.values())
        if e.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_busy_fraction_of_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["GRBM_GUI_ACTIVE"] / 8)
        if e.get("SQ_WAVE_CYCLES"):
            for c, n in (("SQ_WAIT_ANY", "wait_any_per_wave_cycle"), ("SQ_WAIT_INST_ANY", "wait_inst_per_wave_cycle"),
                         ("SQ_ACTIVE_INST_ANY", "active_per_wave_cycle")):
                if c in e:
                    e[n] = e[c] / e["SQ_WAVE_CYCLES"]
        out[k] = e
    json.dump({"note": a.note, "kernels": out}, open(a.out, "w"), indent=1)
    for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        print(f"{k[:70]:70s} waves {e.get('SQ_WAVES', float('nan')):9.0f}  mfma busy {e.get('mfma_busy_fraction_of_simd_cycles', float('nan')):.2f}  "
              f"wait_any {e.get('wait_any_per_wave_cycle', float('nan')):.2f}  wait_inst {e.get('wait_inst_per_wave_cycle', float('nan')):.2f}")


if __name__ == "__main__":
    main()
