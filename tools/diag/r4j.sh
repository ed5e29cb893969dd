#!/bin/bash
# K3a with non-temporal streams; plain (eval) forward with the edge-record prefetch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
for v in "" k3nt1 k3nt3; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
done
python3 - <<PY
import json
for v in ("head","k3nt1","k3nt3"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
for v in "" k2pfall; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  echo "== kbench ${v:-head}"; PYGAT_AMD_LIB=$lib python3 tools/kbench.py --only k2 2>> $O/kbench.err | tee -a $O/kbench_${v:-head}.log
done
