#!/bin/bash
# What the driver runs at round end, in one lease: the whole -m gpu suite, smoke(), the default bench line.
#   gpurun --timeout 1200 -- 'bash tools/diag/full_suite.sh <tag>'
TAG=${1:-full}; R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log
tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read())
print(d["ms_per_step"], d["value"], {k["kernel"]: round(k["avg_ms"],4) for k in d["kernels"]})
print("roofline", {k: d["roofline"][k] for k in ("kernel","frac","traffic")}, "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
