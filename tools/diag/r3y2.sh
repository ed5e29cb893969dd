#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
for w in 0 512; do echo "fwd window $w"; PYGAT_FWD_WINDOW=$w timeout -k 10 200 python3 tools/epoch_profile.py ppi --epochs 300 2>&1 | grep -v amdgpu.ids | tail -1; done
