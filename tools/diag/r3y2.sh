#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
for w in 512 256; do echo "bwd windows for R > $w"; PYGAT_BWD_WINDOW_MIN_R=$w timeout -k 10 300 python3 bench.py --fout 64 --no-cpu --no-epoch --no-v2 --steps 8 2>/dev/null > $O/f64_$w.json; python3 -c "
import json; d=json.load(open('$O/f64_$w.json')); print(d['ms_per_step'], [(k['kernel'], round(k['avg_ms'],3)) for k in d['kernels']])"; done
