#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_ppi.py -q -m gpu -x > $O/pytest_m64.log 2>&1; tail -2 $O/pytest_m64.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ppi_prof -- python3 $R/tools/epoch_profile.py ppi --epochs 30 > $O/ppi_prof.log 2>&1)
python3 tools/epoch_sequence.py $O/ppi_prof --epochs 35 > $O/ppi_seq.txt 2>&1; rm -rf $O/ppi_prof
timeout -k 10 300 python3 bench.py --no-v2 --no-cpu --steps 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v['ms'],4) for k,v in d['epoch_ms'].items()})"
