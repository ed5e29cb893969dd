O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O; cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/pygat_amd/libpygat_amd_x3gw.so
PYGAT_AMD_LIB=$V python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_blocked.py tests/test_gpu_ppi.py -x -q > $O/tests_x3gw.log 2>&1; tail -3 $O/tests_x3gw.log
for v in default x3gw default x3gw; do lib=""; [ $v != default ] && lib=$V; echo "== $v"; PYGAT_AMD_LIB=$lib python3 tools/gemm_bench.py --only ppi 2>&1 | grep -v amdgpu.ids; PYGAT_AMD_LIB=$lib python3 bench.py --no-cpu --no-v2 --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('epoch_ms', {k: round(v['ms'],4) for k,v in d['epoch_ms'].items()})"; done > $O/ab_x3gw.txt 2>&1
cat $O/ab_x3gw.txt
