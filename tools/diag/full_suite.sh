#!/bin/bash
# The round-end checks as the driver runs them: full -m gpu suite, smoke, default bench.  bash tools/diag/full_suite.sh <tag>
TAG=${1:-full}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
grep -q "failed\|error" $O/pytest.log && { grep -n "FAILED\|ERROR" $O/pytest.log | head -20; }
timeout -k 10 200 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], {k: round(v['ms'],4) for k,v in d['epoch_ms'].items()}, d.get('gatv2',{}).get('ms_per_step'))"
echo done
