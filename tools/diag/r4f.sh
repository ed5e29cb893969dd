#!/bin/bash
# round 4, sixth call: da-in-K4 without the end barrier; where its remaining cost sits (diag builds); RCCL world-1 test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
for v in "" k4d1 k4d3; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
done
PYGAT_DA_IN_K4=0 python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_noda.jsonl 2>> $O/bench.err
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_adam.py tests/test_gpu_parity.py -x -q -m gpu -k "rccl or adam or da_of" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python3 - <<PY
import json
for v in ("head","k4d1","k4d3","noda"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
