#!/bin/bash
# round 4, fifth call: the whole -m gpu suite at the new ABI, then the default bench line (cpu child included)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu -s > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log
grep -E "fullsize\[|passed|failed|error" $O/pytest.log | tail -15
[ $rc -eq 0 ] || exit $rc
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read())
print(d["ms_per_step"], d["value"], {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
print(d.get("cpu_baseline")); print(d["roofline"]); print(d.get("epoch_ms")); print(d.get("gatv2",{}).get("ms_per_step"))
PY
