#!/bin/bash
# round 4, second call: parity of the da-in-K4 path and the new ABI, bench of HEAD (K2 prefetch, da in K4), K4 prefetch variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dropout.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in "" k4pf k4pf4; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
done
PYGAT_DA_IN_K4=0 python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_noda.jsonl 2>> $O/bench.err
python3 - <<PY
import json
for v in ("head","k4pf","k4pf4","noda"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
