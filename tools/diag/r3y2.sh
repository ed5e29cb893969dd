#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
for w in 0 256; do for f in 64 128; do echo "fwd large window $w, F' $f"; PYGAT_FWD_WINDOW_LARGE=$w timeout -k 10 300 python3 bench.py --fout $f --no-cpu --no-epoch --no-v2 --steps 8 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [(k['kernel'], round(k['avg_ms'],3)) for k in d['kernels'][:2]])"; done; done
