#!/bin/bash
# round 4, fourth call: da-in-K4 with read-modify-write LDS sums; K1 old epilogue vs transposed one on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
for v in "" k1old; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
  PYGAT_AMD_LIB=$lib python3 tools/gemm_headline_bench.py >> $O/gemm.log 2>> $O/gemm.err
done
PYGAT_DA_IN_K4=0 python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_noda.jsonl 2>> $O/bench.err
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "da or fwd_bwd_small" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python3 - <<PY
import json
for v in ("head","k1old","noda"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
cat $O/gemm.log
