#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_dropout.py tests/test_gpu_graphed.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
grep -q "failed" $O/pytest_sub.log && { tail -60 $O/pytest_sub.log; }
for c in pubmed; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_prof -- python3 $R/tools/epoch_profile.py $c --epochs 30 > $O/${c}_prof.log 2>&1)
  python3 tools/rocprof_top.py $O/${c}_prof --per 35 --top 30 > $O/${c}_top.txt 2>&1
  python3 tools/epoch_sequence.py $O/${c}_prof --epochs 35 > $O/${c}_seq.txt 2>&1; rm -rf $O/${c}_prof
done
timeout -k 10 300 python3 bench.py --no-v2 --no-cpu --steps 10 > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], {k: round(v['ms'],4) for k,v in d['epoch_ms'].items()})"
echo done
