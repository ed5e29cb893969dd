#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_loss.py -q -m gpu -x > $O/pytest_bce.log 2>&1; tail -2 $O/pytest_bce.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ppi_prof -- python3 $R/tools/epoch_profile.py ppi --epochs 30 > $O/ppi_prof.log 2>&1)
python3 tools/epoch_sequence.py $O/ppi_prof --epochs 35 2>&1 | grep bce; rm -rf $O/ppi_prof
