#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
for v in "" k4load k4wrow3; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
done
PYGAT_DA_IN_K4=0 python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_noda.jsonl 2>> $O/bench.err
python3 - <<PY
import json
for v in ("head","k4load","k4wrow3","noda"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
