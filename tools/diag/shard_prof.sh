R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ah; mkdir -p $O; cd $R
A="--as-rank-of 8 --no-cpu --no-epoch --no-v2 --steps 20 --warmup 3"
for v in default k1nw4 default k1nw4; do
  if [ $v = default ]; then unset PYGAT_AMD_LIB; else export PYGAT_AMD_LIB=$R/pygat_amd/libpygat_amd_$v.so; fi
  python3 bench.py $A 2>> $O/err.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v', round(j['ms_per_step'],4), {k['kernel']:round(k['avg_ms'],4) for k in j['kernels']})"
done; unset PYGAT_AMD_LIB
