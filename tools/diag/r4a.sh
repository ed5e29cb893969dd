#!/bin/bash
# round 4, first measurement call: K2 fix A/B (HEAD vs prefetch variants) + decomposition of the two streamed GEMMs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
for v in "" k2pf4 k2pf5; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
done
for v in "" k1d1 k1d2 k1d3 k1d4 k1d8 k1d12 k1d15; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  PYGAT_AMD_LIB=$lib python3 tools/gemm_headline_bench.py >> $O/gemm.log 2>> $O/gemm.err || exit 1
done
PYGAT_AMD_LIB= python3 tools/gemm_headline_bench.py --mode fp32-mfma --tag fp32-mfma >> $O/gemm.log 2>> $O/gemm.err
python3 - <<PY
import json
for v in ("head","k2pf4","k2pf5"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
cat $O/gemm.log
