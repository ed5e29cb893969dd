#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_loss.py tests/test_gpu_graphed.py tests/test_gpu_training.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
grep -q "failed" $O/pytest_sub.log && { tail -60 $O/pytest_sub.log; }
timeout -k 10 300 python3 bench.py --no-v2 --no-cpu --steps 10 > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], {k: round(v['ms'],4) for k,v in d['epoch_ms'].items()})"
echo done
