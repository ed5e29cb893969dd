#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
for f in 0 1; do echo "snap back $f"; PYGAT_SNAP_BACK=$f timeout -k 10 300 python3 bench.py --no-v2 --no-cpu --steps 10 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [(k['kernel'], round(k['avg_ms'],3)) for k in d['kernels'][:4]], {k: round(v['ms'],4) for k,v in d['epoch_ms'].items()})"; done
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_ppi.py tests/test_gpu_gatv2.py -q -m gpu -x > $O/pytest_snap.log 2>&1; tail -3 $O/pytest_snap.log
