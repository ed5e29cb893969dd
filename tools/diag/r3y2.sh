#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R; export TMPDIR=/tmp
for w in 2 4 8; do
  (cd /tmp && PYGAT_DX_ROWS=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$w -- python3 $R/tools/epoch_profile.py pubmed --epochs 30 > $O/p_$w.log 2>&1)
  python3 tools/rocprof_top.py $O/p_$w --per 35 --top 40 2>&1 | grep narrow_dx | sed "s/^/rows $w: /"; rm -rf $O/p_$w
done
