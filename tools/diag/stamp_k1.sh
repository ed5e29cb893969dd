R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ac; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for v in default k1head; do
  if [ $v = default ]; then unset PYGAT_AMD_LIB; else export PYGAT_AMD_LIB=$R/pygat_amd/libpygat_amd_$v.so; fi
  timeout -k 10 120 python3 tools/gemm_headline_bench.py --gap-ms 20 --tag $v >> $O/gemm.log 2>> $O/gemm.err
  timeout -k 10 120 python3 tools/gemm_headline_bench.py --tag $v-b2b >> $O/gemm.log 2>> $O/gemm.err
done; cat $O/gemm.log; unset PYGAT_AMD_LIB
bash tools/ab_variants.sh r4ac default k1head
