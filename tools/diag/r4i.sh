#!/bin/bash
# GR as two tables (Gp rows | per-head records) instead of interleaved 640-byte rows: K3a + K4 (experiment); dist changes under gloo
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
for v in "" grsplit; do
  lib=""; [ -n "$v" ] && lib=$R/pygat_amd/libpygat_amd_$v.so
  for i in 1 2; do PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-epoch --no-v2 --steps 40 --warmup 10 >> $O/bench_${v:-head}.jsonl 2>> $O/bench.err || exit 1; done
done
python3 - <<PY
import json
for v in ("head","grsplit"):
    for l in open("$O/bench_%s.jsonl" % v):
        d=json.loads(l); print(v, round(d["ms_per_step"],4), {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_ppi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 2 --steps 5 --warmup 2 --verify > $O/gloo2.json 2> $O/gloo2.log; echo "gloo2 rc=$?"; grep -E "verify|Error|error" $O/gloo2.log | tail -3
BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29624 bench.py --gpus 4 --heads 4 --steps 3 --warmup 1 --verify > $O/gloo4.json 2> $O/gloo4.log; echo "gloo4 rc=$?"; grep -E "verify|Error|error" $O/gloo4.log | tail -3
