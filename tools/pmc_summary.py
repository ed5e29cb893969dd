#!/usr/bin/env python3
"""Fold rocprofv3 counter-collection CSVs into the per-kernel HBM traffic summary kept under profiles/.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/X/fetch -- python3 bench.py --no-cpu ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/X/write -- python3 bench.py ...
    python3 tools/pmc_summary.py gpurun_out/X/fetch gpurun_out/X/write --out profiles/rNN_pmc_bench.json [--latest]

Separate passes, as MI355X_MICROARCH.md prescribes.  FETCH_SIZE / WRITE_SIZE are KiB summed over the 8
XCDs; FETCH_SIZE is doubled (gfx950 reports 16-B/lane coalesced reads at half their size, same guide).
Values are per-launch means.  `--latest` also rewrites profiles/pmc_latest.json, the file bench.py reads the
`traffic` of every kernel span from (main kernel + its fix-up / reduce launches).
"""
import argparse
import collections
import csv
import glob
import json
import os


def fold(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
    return acc


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").split("<")[0].strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="*")
    ap.add_argument("--out")
    ap.add_argument("--promote", help="an existing summary (profiles/rNN_pmc_bench.json): rewrite profiles/pmc_latest.json from it")
    ap.add_argument("--note", default="")
    ap.add_argument("--latest", action="store_true")
    ap.add_argument("--edges", type=int, default=10760610,
                    help="E of the profiled workload (the headline graph since round 5: numpy stream, seed 1; rounds 1-4: 10758702)")
    ap.add_argument("--heads", type=int, default=8)
    a = ap.parse_args()
    if a.promote:   # the GPU box cannot write profiles/: summaries come back under gpurun_out/, are copied, then promoted here
        doc = json.load(open(a.promote))
        here = os.path.dirname(os.path.abspath(__file__))
        doc["source"] = os.path.relpath(os.path.abspath(a.promote), os.path.join(here, ".."))
        with open(os.path.join(here, "..", "profiles", "pmc_latest.json"), "w") as fh:
            json.dump({k: doc[k] for k in ("note", "source", "workload_edges", "heads_per_gpu", "traffic_bytes")}, fh, indent=1)
        print("pmc_latest.json <-", doc["source"], {k: round(v / 1e9, 3) for k, v in doc["traffic_bytes"].items()})
        return
    assert a.dirs and a.out, "counter directories and --out, or --promote"
    merged = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.dirs:
        for k, ctrs in fold(d).items():
            for c, vals in ctrs.items():
                merged[short(k)][c] += vals
    out = {}
    for k, ctrs in merged.items():
        if not k.startswith("pygat::"):
            continue
        e = {c: sum(v) / len(v) for c, v in ctrs.items()}
        e["launches_seen"] = max(len(v) for v in ctrs.values())
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_traffic_bytes"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
        if e.get("TCC_HIT_sum", 0) + e.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        out[k] = e
    spans = {   # bench.py span -> kernel name prefixes (pygat_amd/ops.py)
        # (bench.py also runs its `alt` pass -- the fp32-MFMA kernels -- under the profiler: the default mode's kernels only)
        "k1_project": ("pygat::gemm_smallk_x3_kernel", "pygat::gemm_rowtile_x3_kernel"),
        "k2_forward": ("pygat::gat_fwd_", "pygat::fwd_tail_kernel"),
        "k3a_prepare": ("pygat::gat_bwd_prepare",),
        "k3b_row": ("pygat::gat_bwd_row_kernel",),
        "k4_backward_col": ("pygat::gat_bwd_col_", "pygat::bwd_tail_kernel", "pygat::col_tail_kernel"),
        "k3c_rowsum": ("pygat::gat_bwd_rowsum_kernel",),
        "k5_agrad": ("pygat::a_grad_partial", "pygat::a_grad_final"),
        "k5_afold": ("pygat::a_grad_fold", "pygat::a_grad_final"),
        "k5_wgrad": ("pygat::gemm_tn_x3_kernel", "pygat::gemm_tn_x3w_kernel", "pygat::gemm_splitk_reduce_kernel", "pygat::unpack_wgrad"),
    }
    traffic = {}
    for span, prefixes in spans.items():
        t = sum(v.get("hbm_traffic_bytes", 0.0) for k, v in out.items() if k.startswith(prefixes))
        if t > 0:
            traffic[span] = t
    here = os.path.dirname(os.path.abspath(__file__))
    # ONE schema for every summary (round 3 lost `roofline.traffic` to a pmc_latest.json written without these keys):
    # bench.py reads workload_edges / heads_per_gpu / traffic_bytes / source from profiles/pmc_latest.json
    doc = {"note": a.note or "rocprofv3 --pmc, separate passes (FETCH_SIZE; WRITE_SIZE+TCC); per-launch means; "
           "FETCH_SIZE/WRITE_SIZE are KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "source": os.path.relpath(os.path.abspath(a.out), os.path.join(here, "..")),
           "workload_edges": a.edges, "heads_per_gpu": a.heads, "traffic_bytes": traffic, "kernels": out}
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    if a.latest:
        with open(os.path.join(here, "..", "profiles", "pmc_latest.json"), "w") as fh:
            json.dump({k: doc[k] for k in ("note", "source", "workload_edges", "heads_per_gpu", "traffic_bytes")}, fh, indent=1)
    print("traffic GB:", {k: round(v / 1e9, 3) for k, v in traffic.items()})
    for k, v in sorted(out.items()):
        print(f"{k:48s} {v.get('hbm_traffic_bytes', float('nan')) / 1e6:10.1f} MB  L2 hit {v.get('l2_hit_rate', float('nan')):.2f}")


if __name__ == "__main__":
    main()
