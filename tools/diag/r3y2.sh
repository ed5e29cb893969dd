#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
for w in 256 512; do echo "large bwd window $w"; PYGAT_BWD_LARGE_WINDOW=$w timeout -k 10 300 python3 bench.py --fout 128 --no-cpu --no-epoch --no-v2 --steps 8 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('kernels_ms') or d.get('spans_ms'))"; done
