R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ab; mkdir -p $O; cd $R
for v in default k1stg20 k1stg44; do
  if [ $v = default ]; then unset PYGAT_AMD_LIB; else export PYGAT_AMD_LIB=$R/pygat_amd/libpygat_amd_$v.so; fi
  timeout -k 10 120 python3 tools/gemm_headline_bench.py --gap-ms 20 --tag $v >> $O/gemm.log 2>> $O/gemm.err
done; cat $O/gemm.log; unset PYGAT_AMD_LIB
bash tools/ab_variants.sh r4ab default k1stg20 k1stg44
