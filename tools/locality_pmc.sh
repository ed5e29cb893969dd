#!/bin/bash
# PMC counters (FETCH_SIZE; WRITE_SIZE + L2 hits / misses: separate passes) of the level's kernels under two node numberings
# (tools/locality_probe.py): does renumbering by degree buy cache hits, or something else?   bash tools/locality_pmc.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R; export TMPDIR=/tmp
prof() { (cd /tmp && timeout -k 10 400 rocprofv3 "$@"); }
for ord in generator degree; do
  prof --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$ord -- python3 $R/tools/locality_probe.py --orders $ord --steps 4 > $O/f_$ord.log 2>&1
  prof --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/w_$ord -- python3 $R/tools/locality_probe.py --orders $ord --steps 4 > $O/w_$ord.log 2>&1
  python3 tools/pmc_summary.py $O/f_$ord $O/w_$ord --out $O/pmc_$ord.json > $O/sum_$ord.log 2>&1
  rm -rf $O/f_$ord $O/w_$ord
done
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for ord_ in ("generator", "degree"):
    d = json.load(open(f"{O}/pmc_{ord_}.json"))
    for k, v in d["kernels"].items():
        if "gat_fwd_kernel" in k or "gat_bwd_col_kernel" in k or "prepare" in k:
            print(ord_, k[:40], "traffic GB", round(v.get("hbm_traffic_bytes", 0) / 1e9, 3), "read GB", round(v.get("hbm_read_bytes_corrected", 0) / 1e9, 3),
                  "L2 hit", round(v.get("l2_hit_rate", 0), 3), "hits", int(v.get("TCC_HIT_sum", 0)), "misses", int(v.get("TCC_MISS_sum", 0)))
PY
