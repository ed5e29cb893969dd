#!/bin/bash
# K4: what the row-local Wh_j re-reads (bit 2) and the per-head scalar records of the gathered row (bit 3) cost
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
for v in "" k4d4 k4d8 k4d12; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
done
python3 - <<PY
import json
for v in ("head","k4d4","k4d8","k4d12"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
for n in 8 4 2; do python3 bench.py --as-rank-of $n --no-cpu --no-epoch --no-v2 --steps 20 2>> $O/rank.err; done > $O/as_rank_of.jsonl
python3 - <<PY
import json
for l in open("$O/as_rank_of.jsonl"):
    x=json.loads(l); print(x['config']['heads_per_gpu'], round(x['ms_per_step'],4), {k['kernel']: round(k['avg_ms'],3) for k in x['kernels']})
PY
