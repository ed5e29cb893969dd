R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4aj; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_project_narrow_shapes.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for v in default svold default svold; do
  if [ $v = default ]; then unset PYGAT_AMD_LIB; else export PYGAT_AMD_LIB=$R/pygat_amd/libpygat_amd_$v.so; fi
  for n in 8 4; do
  python3 bench.py --as-rank-of $n --no-cpu --no-epoch --no-v2 --steps 20 --warmup 3 2>> $O/err.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v rank-of-$n', round(j['ms_per_step'],4), {k['kernel']:round(k['avg_ms'],4) for k in j['kernels']})"
  done
done; unset PYGAT_AMD_LIB
