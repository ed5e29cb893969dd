R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ad; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for v in default tnwpad tnw3; do
  if [ $v = default ]; then unset PYGAT_AMD_LIB; else export PYGAT_AMD_LIB=$R/pygat_amd/libpygat_amd_$v.so; fi
  for sk in 512 768; do
  timeout -k 10 120 python3 tools/gemm_headline_bench.py --gap-ms 20 --split-k $sk --tag $v-$sk >> $O/gemm.log 2>> $O/gemm.err
  done
done; cat $O/gemm.log
