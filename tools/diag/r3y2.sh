#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_graphed.py tests/test_gpu_training.py -q -m gpu -x > $O/pytest_fork.log 2>&1; tail -2 $O/pytest_fork.log
for f in 0 1; do echo "fork $f"; PYGAT_PROLOGUE_FORK=$f timeout -k 10 300 python3 bench.py --no-v2 --no-cpu --steps 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v['ms'],4) for k,v in d['epoch_ms'].items()})"; done
